import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import importlib
pkg = importlib.import_module("tc2li-slam_amd")
syn = importlib.import_module("tc2li-slam_amd.synthetic")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
imgs = []
for s in range(8):
    l, r = syn.stereo_pair(s, 1241, 376)
    imgs += [l, r]
batch = np.stack([imgs[i % 16] for i in range(n)])
dev = torch.from_numpy(batch).cuda()
e = pkg.OrbExtractor(nfeatures=2000, max_width=1241, max_height=376, max_images=n)
out = None
for prof in (0, 1):
    e.set_profiling(prof)
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        out = e.extract_batch_dev(dev.data_ptr(), n, 1241, 376, 1241, 1241 * 376, out=out)
        ts.append((time.perf_counter() - t0) * 1e3)
        t = e.last_timings()
    print("profiling", prof, "wall ms", np.round(ts, 2), "stages", np.round(t, 3), flush=True)
print("kp counts", out[2][:4])
