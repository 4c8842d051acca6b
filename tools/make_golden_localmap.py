"""Golden vectors of the local-map bookkeeping (Tracking::UpdateLocalKeyFrames / UpdateLocalPoints on a flat graph mirror):
tests/golden/localmap_a.npz = a seeded 90-keyframe graph, three frames' map-point lists (one with the temporal block) and the
oracle's local keyframes / reference keyframe / local points / cleared flags for each.  The reference ships no vectors for this
path; these are made here from the CPU oracle and committed with this script.
Run from the repository root:  python tools/make_golden_localmap.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from oracle import pyoracle  # noqa: E402
from test_localmap import random_graph, frame_points_near  # noqa: E402  (the seeded generators of the parity test)

pyoracle.build()
g = random_graph(31, 90, 2500, slots=200)
rng = np.random.default_rng(310)
out = {k: np.asarray(v) for k, v in g.items()}
cases = [(20, -1), (45, 44), (89, -1)]
for i, (kf, temporal) in enumerate(cases):
    fp = frame_points_near(rng, g, kf, 300)
    kfs, ref, pts, cleared = pyoracle.update_local_map(g, fp, temporal)
    out.update({"frame_points_%d" % i: fp, "temporal_%d" % i: np.int32(temporal), "local_kfs_%d" % i: kfs, "reference_%d" % i: np.int32(ref),
                "local_points_%d" % i: pts, "cleared_%d" % i: cleared})
    print("case", i, "keyframes", len(kfs), "reference", ref, "points", len(pts), "cleared", int(cleared.sum()))
path = os.path.join(ROOT, "tests", "golden", "localmap_a.npz")
np.savez_compressed(path, n_cases=np.int32(len(cases)), **out)
print("localmap_a", os.path.getsize(path) // 1024, "KiB")
