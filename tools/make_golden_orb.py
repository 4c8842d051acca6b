"""Generates tests/golden/orb_*.npz from the CPU oracle on seeded synthetic inputs (the
reference ships no golden vectors, so they are created here and committed together with this script).
Run from the repository root:  python tools/make_golden_orb.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from tc2li_slam_amd import synthetic  # noqa: E402
from oracle import pyoracle  # noqa: E402

out_dir = os.path.join(ROOT, "tests", "golden")
os.makedirs(out_dir, exist_ok=True)
cases = [
    # name, seed, crop (x0, y0, w, h), nfeatures, iniTh, minTh
    ("orb_a", 11, (100, 60, 400, 240), 500, 20, 7),
    ("orb_b", 12, (500, 100, 331, 201), 300, 12, 7),
]
cases.append(("orb_gauss_rounded", 11, (100, 60, 400, 240), 500, 20, 7))  # orb_a under the alternative Gaussian taps (oracle/cv_restate.hpp)
for name, seed, (x0, y0, w, h), nf, ini, mn in cases:
    pyoracle.set_gauss_variant("rounded" if name == "orb_gauss_rounded" else "error-diffused")
    left, _ = synthetic.stereo_pair(seed)
    img = np.ascontiguousarray(left[y0:y0 + h, x0:x0 + w])
    o = pyoracle.OrbOracle(nfeatures=nf, ini_th_fast=ini, min_th_fast=mn)
    mono, kps, desc = o.extract(img)
    kp_arr = np.stack([kps[f].astype(np.float32) for f in ("x", "y", "size", "angle", "response")] +
                      [kps["octave"].astype(np.float32)], 1)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), image=img, params=np.array([nf, ini, mn], np.int32),
                        mono=np.int32(mono), keypoints=kp_arr, descriptors=desc,
                        level7=o.level(7), blurred3=o.blurred(3))
    print(name, img.shape, mono, len(kps))
pyoracle.set_gauss_variant("error-diffused")
