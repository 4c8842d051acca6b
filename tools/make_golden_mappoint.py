"""Golden vectors of the map-point refresh (MapPoint::ComputeDistinctiveDescriptors + UpdateNormalAndDepth):
tests/golden/mappoint_a.npz = 200 points with 0 .. 30 observations (descriptors, camera centres), and the oracle's chosen observation,
mean viewing direction and scale-invariance distances.  The reference ships no vectors for this path; these are made here from the CPU
oracle and committed with this script.
Run from the repository root:  python tools/make_golden_mappoint.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from oracle import pyoracle  # noqa: E402
from test_mappoint import problem  # noqa: E402  (the seeded generator of the parity test)

pyoracle.build()
off, desc, centres, pos, ref, scales, last = problem(17, 200, 30)
best, normals, mn, mx = pyoracle.map_points_refresh(off, desc, centres, pos, ref, scales, last)
path = os.path.join(ROOT, "tests", "golden", "mappoint_a.npz")
np.savez_compressed(path, obs_off=off, descriptors=desc, centres=centres, positions=pos, ref_centres=ref, level_scale=scales, last_scale=np.float32(last),
                    out_best=best, out_normals=normals, out_min=mn, out_max=mx)
print("points", len(best), "observations", int(off[-1]), "mappoint_a", os.path.getsize(path) // 1024, "KiB")
