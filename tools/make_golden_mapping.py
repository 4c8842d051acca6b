"""Golden vectors of LocalMapping's geometric steps (ORBmatcher::SearchForTriangulation, the pair loop of
LocalMapping::CreateNewMapPoints, the search of ORBmatcher::Fuse): tests/golden/mapping_a.npz = three keyframes of a synthetic drive as the
oracle's ORB + stereo stages produce them (keypoints, descriptors, uRight / depth, vocabulary nodes, poses), and the oracle's matches,
created points and fused keypoints.  The reference ships no vectors for this path; these are made here from the CPU oracle and committed
with this script.
Run from the repository root:  python tools/make_golden_mapping.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from tc2li_slam_amd import synthetic  # noqa: E402
from oracle import pyoracle  # noqa: E402
import test_mapping as T  # noqa: E402  (the seeded keyframe generator of the parity test)

pyoracle.build()
T.NFEAT = 700
kfs = T.make_keyframes(synthetic, T.oracle_features(pyoracle, synthetic), None, 3, seed=8)
cam4, mbf, mb = T.cam_of(synthetic)
sf, sg = T.tables()
out = dict(n_kf=np.int32(len(kfs)), cam4=cam4, mbf=np.float32(mbf), mb=np.float32(mb), sf=sf, sg=sg, width=np.int32(T.W), height=np.int32(T.H))
kpf = lambda k: np.stack([k[f].astype(np.float32) for f in ("x", "y", "size", "angle", "response")] + [k["octave"].astype(np.float32)], 1)
for i, kf in enumerate(kfs):
    out.update({"keys_%d" % i: kpf(kf["keys"]), "desc_%d" % i: kf["descriptors"], "u_right_%d" % i: kf["u_right"], "depth_%d" % i: kf["depth"],
                "has_point_%d" % i: kf["has_point"], "fv_node_%d" % i: kf["fv_node"], "fv_offset_%d" % i: kf["fv_offset"], "fv_index_%d" % i: kf["fv_index"],
                "pose7_%d" % i: kf["pose7"], "centre_%d" % i: kf["centre"]})
n, m = pyoracle.search_for_triangulation(kfs[0], kfs[2], cam4, sf, sg)
idx, x3 = pyoracle.create_new_map_points(kfs[0], kfs[1:], cam4, mb, mbf, sf, sg)
B, pts, valid, _, _, _, isg, logsf = T.fuse_problem(pyoracle, kfs, synthetic, 2, 0, seed=2)
nf, bi, bd = pyoracle.fuse_search(B["keys"], B["descriptors"], B["u_right"], T.W, T.H, B["pose7"], cam4, mbf, sf, isg, logsf, pts, valid, th=3.0)
print("keypoints", [len(k["keys"]) for k in kfs], "triangulation matches", n, "new points", len(idx), "fused", nf)
out.update(out_tri_n=np.int32(n), out_tri_matches=m, out_new_idx=idx, out_new_x3=x3, fuse_points=pts, fuse_valid=valid, fuse_isg=isg,
           fuse_logsf=np.float64(logsf), out_fuse_n=np.int32(nf), out_fuse_idx=bi, out_fuse_dist=bd)
path = os.path.join(ROOT, "tests", "golden", "mapping_a.npz")
np.savez_compressed(path, **out)
print("mapping_a", os.path.getsize(path) // 1024, "KiB")
