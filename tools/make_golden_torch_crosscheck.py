"""An independent statement of the three OpenCV float-convention pieces the ORB oracle restates in fixed point -- INTER_LINEAR resize, the
7 x 7 sigma 2 Gaussian with BORDER_REFLECT_101, fastAtan2 -- made with PyTorch operators (torch is in the build container; OpenCV is not):
tests/golden/torch_crosscheck.npz holds the inputs and torch's float results; tests/test_oracle_orb.py::test_torch_crosscheck holds the
oracle to them within the fixed-point error bounds.  This does not pin the oracle to OpenCV's exact integers (nothing here can), it pins
its sampling convention (half-pixel centres), border handling and angle convention to a second, unrelated implementation.
Run from the repository root:  python tools/make_golden_torch_crosscheck.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from tc2li_slam_amd import synthetic  # noqa: E402

left, _ = synthetic.stereo_pair(31)
img = np.ascontiguousarray(left[40:160, 200:411])  # 120 x 211: odd width
t = torch.from_numpy(img.astype(np.float64))[None, None]
out = dict(image=img)
# cv::resize(INTER_LINEAR): sample at (dst + 0.5) * scale - 0.5, clamp -- torch: bilinear, align_corners=False, antialias off
for k, (w, h) in enumerate([(176, 100), (147, 83), (50, 29)]):
    out["resize_%d" % k] = F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
    out["resize_size_%d" % k] = np.array([w, h], np.int32)
# cv::GaussianBlur(7 x 7, sigma 2, BORDER_REFLECT_101): reflect padding without repeating the border pixel
g = torch.exp(-torch.arange(-3, 4, dtype=torch.float64) ** 2 / 8.0)
g = g / g.sum()
pad = F.pad(t, (3, 3, 3, 3), mode="reflect")
out["blur"] = F.conv2d(F.conv2d(pad, g.view(1, 1, 1, 7)), g.view(1, 1, 7, 1))[0, 0].numpy()
# cv::fastAtan2(y, x) in degrees, [0, 360)
rng = np.random.default_rng(5)
y, x = rng.normal(0, 50, 4000).astype(np.float32), rng.normal(0, 50, 4000).astype(np.float32)
y[:8] = [0, 0, 1, -1, 1, -1, 5, -5]; x[:8] = [1, -1, 0, 0, 1, -1, -5, 5]
out["atan_y"], out["atan_x"] = y, x
out["atan_deg"] = (torch.rad2deg(torch.atan2(torch.from_numpy(y.astype(np.float64)), torch.from_numpy(x.astype(np.float64)))) % 360.0).numpy()
path = os.path.join(ROOT, "tests", "golden", "torch_crosscheck.npz")
np.savez_compressed(path, **{k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in out.items()})
print("torch_crosscheck", os.path.getsize(path) // 1024, "KiB, torch", torch.__version__)
