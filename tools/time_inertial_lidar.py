"""Per-kernel device time of tc2li_lidar_inertial_frontend_batch + map_incremental for S sequences with the GPU to itself.
python tools/time_inertial_lidar.py [S]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
import torch
from scipy.spatial.transform import Rotation
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
U = 4
scans, maps0, li = [], [], []
for u in range(U):
    sc = synthetic.Scene(u)
    scans.append(synthetic.lidar_scan(sc, u + 1))
    maps0.append(synthetic.lidar_map(sc, x_from=-700.0, x_to=300.0))
    R, p = synthetic.sensor_pose(u + 1)
    rng = np.random.default_rng(300 + u)
    ts = np.arange(998, 1012) / 100.0
    imu = np.zeros((len(ts), 7)); imu[:, 0] = ts
    imu[:, 1:4] = np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.02, (len(ts), 3)); imu[:, 4:7] = rng.normal(0, 0.002, (len(ts), 3))
    x = np.concatenate([p + rng.normal(0, 0.02, 3), (R @ Rotation.from_rotvec(rng.normal(0, 0.002, 3)).as_matrix()).ravel(), [10.0, 0.0, 0.0], np.zeros(3), np.zeros(3), [0, 0, -9.81], np.eye(3).ravel(), np.zeros(3)])
    A = rng.normal(0, 1, (23, 23))
    P = A @ A.T * 1e-6 + np.diag([1e-3] * 3 + [1e-4] * 3 + [1e-5] * 6 + [1e-2] * 3 + [1e-5] * 6 + [1e-6] * 2)
    li.append(dict(imu=imu, x=x, P=P, times=[10.0, 10.1, 9.999, 1.0]))
tile = [s % U for s in range(S)]
fe = pkg.LidarFrontEnd(max_points_per_scan=int(max(len(x) for x in scans)), max_scans=S)
maps = []
for t in tile:
    m = pkg.LidarMap(); m.Build(maps0[t]); maps.append(m)
raw = np.concatenate([scans[t] for t in tile])
offs = np.concatenate([[0], np.cumsum([len(scans[t]) for t in tile])]).astype(np.int32)
dev = torch.from_numpy(raw.view(np.uint8)).cuda()
stream = torch.cuda.current_stream().cuda_stream
b = pkg.capi.LidarInertialBatch(fe, offs, maps, np.stack([li[t]["x"] for t in tile]), np.stack([li[t]["P"] for t in tile]), [li[t]["imu"] for t in tile],
                                np.array([li[t]["times"] for t in tile]), np.array([0.1] * 6 + [1e-4] * 6), max_iter=3)
def step():
    b.run(dev.data_ptr(), stream)
    x = b.states36()
    st24 = np.concatenate([x[:, 3:12], x[:, 0:3], x[:, 24:33], x[:, 33:36]], 1)
    pkg.capi.map_incremental_batch(fe, np.arange(S, dtype=np.int32), maps, st24, stream=stream)
for _ in range(2): step()
t0 = time.perf_counter()
for _ in range(3): step()
wall = (time.perf_counter() - t0) / 3
pkg.capi.profile_enable(True)
for _ in range(3): step()
pkg.capi.profile_enable(False)
rep = pkg.capi.profile_report()
tot = sum(ms for _, ms in rep.values())
print("%d sequences: %.2f ms wall per step, %.2f ms of kernel time; scan stats %s" % (S, 1e3 * wall, tot / 3, b.stats()[0]))
for name, (calls, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-40s %5d launches  %9.1f us avg  %6.2f %%" % (name.split("(")[0][:40], calls, 1e3 * ms / calls, 100 * ms / tot))
