"""The ORB extraction of N images (default 1024 = the stereo pairs of 512 sequences) alone on the GPU: ms per call and the kernels' own
durations per call (tc2li_profile_*).  python tools/time_orb_alone.py [N]   (VERDICT r5 item 4's operating point: `--stages orb` alone)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, H = synthetic.WIDTH, synthetic.HEIGHT
imgs = []
for u in range(4):
    sc = synthetic.Scene(u)
    imgs.append(sc.render(0.0, W, H, noise_seed=1)[0]); imgs.append(sc.render(synthetic.BASELINE, W, H, noise_seed=2)[0])
dev = torch.from_numpy(np.stack([imgs[k % len(imgs)] for k in range(N)])).cuda()
ext = pkg.OrbExtractor(max_width=W, max_height=H, max_images=N)
st = torch.cuda.Stream()
out = None
for _ in range(3): out = ext.extract_batch_dev(dev.data_ptr(), N, W, H, W, W * H, stream=st.cuda_stream, out=out)
torch.cuda.synchronize()
reps = 8
t = time.perf_counter()
for _ in range(reps): out = ext.extract_batch_dev(dev.data_ptr(), N, W, H, W, W * H, stream=st.cuda_stream, out=out)
torch.cuda.synchronize()
print("%d images: %.3f ms per call, %.1f keypoints per image" % (N, (time.perf_counter() - t) * 1e3 / reps, float(np.mean(out[2]))))
if os.environ.get("TC2LI_ORB_SERIAL"):   # every kernel on one stream, one after the other: its duration is its own
    ext.set_profiling(True)
pkg.capi.profile_enable(True)
for _ in range(3): out = ext.extract_batch_dev(dev.data_ptr(), N, W, H, W, W * H, stream=st.cuda_stream, out=out)
torch.cuda.synchronize()
pkg.capi.profile_enable(False)
rep = pkg.capi.profile_report()
px = sum(a * b for a, b in [ext.level_size(l) for l in range(8)])
print("   kernel time per call %.3f ms in %d launches; pyramid %.3f MPx per image" % (sum(v[1] for v in rep.values()) / 3, sum(v[0] for v in rep.values()) // 3, px / 1e6))
for name, (n, t_ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:16]:
    print("   %-30s %3d launches per call, %8.1f us each, %7.3f ms per call" % (name, n // 3, 1e3 * t_ms / n, t_ms / 3))
