"""Golden vectors of the LiDAR preprocessing and down-sampling (SURVEY 8a rows b1 + b3: Preprocess::process for velodyne input and
pcl::VoxelGrid::filter): tests/golden/lidar_pre_a.npz = a contiguous part of a raw synthetic scan (ring-ordered velodyne points), the
oracle's preprocessed points (point_filter_num 2, blind 2 m) and the 0.5 m voxel-filtered cloud.  The reference ships no vectors for this
path; these are made here from the CPU oracle and committed with this script.
Run from the repository root:  python tools/make_golden_lidar_pre.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from tc2li_slam_amd import synthetic  # noqa: E402
from oracle import pyoracle  # noqa: E402

pyoracle.build()
scan = synthetic.lidar_scan(synthetic.Scene(7), 3)
raw = scan[::9].copy()  # every 9th return: all rings and azimuths stay represented
pre = pyoracle.lidar_preprocess(raw, 2, 2.0, 1e-3)
down = pyoracle.voxel_grid(pre, 0.5)
print("raw", len(raw), "preprocessed", len(pre), "down-sampled", len(down))
path = os.path.join(ROOT, "tests", "golden", "lidar_pre_a.npz")
np.savez_compressed(path, raw=raw, point_filter_num=np.int32(2), blind=np.float64(2.0), time_unit_scale=np.float32(1e-3), leaf=np.float32(0.5),
                    out_pre=pre, out_down=down)
print("lidar_pre_a", os.path.getsize(path) // 1024, "KiB")
