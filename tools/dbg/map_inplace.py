"""Debug driver: the bench's LiDAR maps (a street's accumulated map, 8 scenes) through lasermap_fov_segment, feature extraction and
map_incremental in place, with the grids checked on the host after every step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tc2li_loader
pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic

n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
U = int(os.environ.get("SCENES", "8"))
L = float(os.environ.get("MAPLEN", "1000"))
scans, states, mps = [], [], []
for u in range(U):
    sc = synthetic.Scene(u)
    scans.append(synthetic.lidar_scan(sc, u + 1))
    states.append(pkg.pack_lidar_state(*synthetic.lidar_state(u + 1)[:2]))
    mps.append(synthetic.lidar_map(sc, x_from=-0.7 * L, x_to=0.3 * L))
    print("scene", u, "map points", len(mps[-1]), "scan", len(scans[-1]), flush=True)
tile = [s % U for s in range(n_maps)]
fe = pkg.LidarFrontEnd(max_points_per_scan=max(len(s) for s in scans), max_scans=n_maps)
maps = []
for t in tile:
    m = pkg.LidarMap(); m.Build(mps[t]); maps.append(m)
import torch
raw = torch.from_numpy(np.concatenate([scans[t] for t in tile]).view(np.uint8)).cuda()
offs = np.concatenate([[0], np.cumsum([len(scans[t]) for t in tile])]).astype(np.int32)
st = np.stack([states[t] for t in tile])
boxes = [pkg.capi.LocalMapBox() for _ in range(n_maps)]
stream = torch.cuda.Stream(priority=-1).cuda_stream if os.environ.get("OWN_STREAM", "1") == "1" else 0
for step in range(int(os.environ.get("STEPS", "6"))):
    todo_maps, todo_boxes = [], []
    for s in range(n_maps):
        b = pkg.capi.lidar_fov_segment(boxes[s], st[s][9:12], cube_len=1000.0, det_range=100.0)
        if len(b):
            todo_maps.append(maps[s]); todo_boxes.append(b)
    if todo_maps:
        print("step", step, "box deletions for", len(todo_maps), "maps", flush=True)
        pkg.capi.delete_point_boxes_batch(todo_maps, todo_boxes, stream=stream)
    counts = fe.frontend_batch(raw.data_ptr(), offs, maps, st, stream=stream, want_points=False)[0]
    na, nn, sizes = pkg.capi.map_incremental_batch(fe, np.arange(n_maps, dtype=np.int32), maps, st, stream=stream)
    print("step", step, "to_add", na[:8], "no_need", nn[:8], "sizes", sizes[:8], flush=True)
    for m in maps[:U] + maps[-2:]:
        print("   ", m.stats(), flush=True)
        m.grid()
    print("  grids sound", flush=True)
print("done")
