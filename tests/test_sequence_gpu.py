"""End-to-end camera path over a synthetic drive (the "ATE vs ref" half of BASELINE.json's metric): stereo ORB -> stereo matching
-> TrackWithMotionModel, frame after frame, the map of every frame being the stereo points of the frame before
(Tracking::UpdateLastFrame in odometry mode).  The product (C ABI, device-resident features) and the oracle run the same
harness; their trajectories are written in the reference's KITTI / TUM formats (tools/trajectory.py) and compared:
ATE(product, oracle) <= 1e-4 m (BASELINE.json: 1e-4 relative on SE3 poses), ATE(product, ground truth) at the stereo noise level."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

W, H = 1242, 375
N_FRAMES = 10


def camera_position(k):
    return np.array([0.06 * np.sin(0.7 * k), 0.0, 0.4 * k])  # 4 m/s at 10 Hz with a little lateral sway (the scene has a box 5 m ahead)


def compose(a, b):
    """SE3 product a * b of (q, t) 7-vectors in double."""
    import trajectory as T
    Ra, Rb = T.quat_to_R(a[:4]), T.quat_to_R(b[:4])
    return np.concatenate([T.R_to_quat(Ra @ Rb), Ra @ b[4:] + a[4:]])


def inverse(a):
    import trajectory as T
    R = T.quat_to_R(a[:4])
    return np.concatenate([T.R_to_quat(R.T), -R.T @ a[4:]])


def last_frame_from(keys, desc, depth, pose7, synthetic):
    """The previous frame as SearchByProjection sees it: its stereo points un-projected with its (estimated) pose."""
    import trajectory as T
    fx, fy, cx, cy = [np.float32(v) for v in (synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY)]
    has_point = (depth > 0).astype(np.uint8)
    z = np.where(depth > 0, depth, 1).astype(np.float32)
    Xc = np.stack([(keys["x"] - cx) * z / fx, (keys["y"] - cy) * z / fy, z], 1).astype(np.float64)
    R = T.quat_to_R(pose7[:4])
    Xw = ((Xc - pose7[4:].astype(np.float64)) @ R).astype(np.float32)  # Rwc (Xc - tcw)
    return dict(has_point=has_point, outlier=np.zeros(len(keys), np.uint8), Xw=Xw, keys=keys.copy(), descriptors=desc.copy(),
                pose7=pose7.astype(np.float32))


def drive(frontend, track, synthetic):
    """frontend(k) -> (keys, desc, u_right, depth) of frame k; track(k, features, last, pred) -> pose7 (double), matches, inliers."""
    poses = [np.concatenate([[0, 0, 0, 1], -camera_position(0)])]
    feats = frontend(0)
    stats = []
    for k in range(1, N_FRAMES):
        last = last_frame_from(feats[0], feats[1], feats[3], poses[-1].astype(np.float32), synthetic)
        if k == 1:
            velocity = np.concatenate([[0, 0, 0, 1], camera_position(0) - camera_position(1)])  # initial motion prior
        else:
            velocity = compose(poses[-1], inverse(poses[-2]))  # mVelocity = Tcw(k-1) * Twc(k-2)
        pred = compose(velocity, poses[-1]).astype(np.float32)
        feats = frontend(k)
        pose, nm, inl = track(k, feats, last, pred)
        assert inl >= 30, (k, nm, inl)
        poses.append(np.asarray(pose, np.float64))
        stats.append((nm, inl))
    return np.stack(poses), stats


def test_sequence_ate(pkg, oracle, synthetic, tmp_path):
    import torch
    import trajectory as T
    scene = synthetic.Scene(77)
    frames = []
    for k in range(N_FRAMES):
        c = camera_position(k)
        left, _ = scene.render(c[0], W, H, noise_seed=2 * k + 1, cam_z=c[2])
        right, _ = scene.render(c[0] + synthetic.BASELINE, W, H, noise_seed=2 * k + 2, cam_z=c[2])
        frames.append((left, right))
    bf = float(np.float32(synthetic.BF))
    b = float(np.float32(synthetic.BF) / np.float32(synthetic.FX))
    cam5 = np.float32([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, bf]).astype(np.float64)

    # ---- product: everything through the C ABI, features stay on the device ----
    ext = pkg.OrbExtractor(max_width=W, max_height=H, max_images=2)
    keep = {}

    def gpu_frontend(k):
        dev = torch.from_numpy(np.stack(frames[k])).cuda()
        kps, desc, counts, _ = ext.extract_batch_dev(dev.data_ptr(), 2, W, H, W, W * H)
        u_right, depth, _ = pkg.stereo_match_batch(ext, 1, bf, b)
        n = int(counts[0])
        keep[k] = (kps, u_right)
        return kps[0, :n].copy(), desc[0, :n].copy(), u_right[0, :n].copy(), depth[0, :n].copy()

    def gpu_track(k, feats, last, pred):
        kps, u_right = keep[k]
        poses, mp, nm, inl = pkg.capi.track_motion_model_batch(ext, 1, kps, u_right, pkg.capi.pack_last_frames([last]), pred[None], cam5, b, 7.0)
        return poses[0], int(nm[0]), int(inl[0])

    got, got_stats = drive(gpu_frontend, gpu_track, synthetic)

    # ---- oracle ----
    ol, orr = oracle.OrbOracle(), oracle.OrbOracle()
    scales, inv_sigma2 = ext.GetScaleFactors(), ext.GetInverseScaleSigmaSquares()

    def cpu_frontend(k):
        _, kl, dl = ol.extract(frames[k][0])
        _, kr, dr = orr.extract(frames[k][1])
        u, d, _ = oracle.stereo_match(ol, orr, kl, dl, kr, dr, bf, b)
        return kl, dl, u, d

    def cpu_track(k, feats, last, pred):
        r = oracle.track_motion_model(feats[0], feats[1], feats[2], W, H, scales, inv_sigma2, pred, last["pose7"], cam5, b, 7.0, last["has_point"],
                                      last["outlier"], last["Xw"], last["keys"], last["descriptors"])
        return r[0], int(r[2]), int(r[3])

    want, want_stats = drive(cpu_frontend, cpu_track, synthetic)

    assert got_stats == want_stats
    truth = np.stack([np.concatenate([[0, 0, 0, 1], -camera_position(k)]) for k in range(N_FRAMES)])
    files = {}
    for name, poses in (("product", got), ("oracle", want), ("truth", truth)):
        files[name] = str(tmp_path / (name + "_kitti.txt"))
        T.save_kitti(files[name], poses)
        T.save_tum(str(tmp_path / (name + "_tum.txt")), poses, 0.1 * np.arange(N_FRAMES))
    xyz = {n: T.load_positions(f) for n, f in files.items()}
    assert np.allclose(T.load_positions(str(tmp_path / "product_tum.txt")), xyz["product"], atol=1e-8)  # both writers agree
    length = float(np.linalg.norm(np.diff(xyz["truth"], axis=0), axis=1).sum())
    ate_ref = T.ate_rmse(xyz["product"], xyz["oracle"], align=False)
    ate_truth = T.ate_rmse(xyz["product"], xyz["truth"])
    print("path %.2f m: ATE(product, oracle) = %.3g m, ATE(product, truth) = %.4f m, matches/inliers of the last frame %s"
          % (length, ate_ref, ate_truth, got_stats[-1]))
    assert ate_ref <= 1e-4 and ate_ref <= 1e-4 * length
    assert ate_truth < 0.05
    assert min(i for _, i in got_stats) > 100
