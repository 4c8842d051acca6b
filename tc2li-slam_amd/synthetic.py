"""Deterministic synthetic workload of SURVEY.md section 8(d): KITTI-00-like stereo pairs and 64-beam scans of a
ground plane plus textured boxes.  There is no dataset in this environment; bench.py, the parity tests and
``__graft_entry__.smoke()`` all draw their inputs from here (numpy only, PCG64 streams, fixed seeds)."""
import numpy as np

# KITTI-00 pinhole (config/Camera-Lidar/KITTI00-02.yaml:9-28 of the reference)
FX = FY = 718.856
CX, CY = 607.1928, 185.2157
BF = 386.1448
BASELINE = BF / FX
WIDTH, HEIGHT = 1242, 375
CAM_HEIGHT = 1.65
SEED0 = 0x7C211000


def _hash01(ix, iy, salt):
    """Integer lattice hash -> float32 in [0, 1) (uint32 wrap-around arithmetic)."""
    h = (ix.astype(np.uint32) * np.uint32(73856093)) ^ (iy.astype(np.uint32) * np.uint32(19349663)) ^ np.uint32(
        (int(salt) * 83492791) & 0xFFFFFFFF)
    h = (h ^ (h >> np.uint32(16))) * np.uint32(0x45D9F3B)
    h = (h ^ (h >> np.uint32(16))) * np.uint32(0x45D9F3B)
    h = h ^ (h >> np.uint32(16))
    return (h & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def _value_noise(u, v, cell, salt):
    x, y = u / cell, v / cell
    ix, iy = np.floor(x), np.floor(y)
    fx, fy = (x - ix).astype(np.float32), (y - iy).astype(np.float32)
    ix, iy = ix.astype(np.int32), iy.astype(np.int32)
    a = _hash01(ix, iy, salt)
    b = _hash01(ix + 1, iy, salt)
    c = _hash01(ix, iy + 1, salt)
    d = _hash01(ix + 1, iy + 1, salt)
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    return (a * (1 - sx) + b * sx) * (1 - sy) + (c * (1 - sx) + d * sx) * sy


def _texture(u, v, salt, base_cell):
    """Band-limited 1/f value noise (6 octaves) plus hard-edged checker patches, in [0, 255]."""
    t = np.zeros_like(u, dtype=np.float32)
    amp, tot, cell = 1.0, 0.0, base_cell
    for o in range(6):
        t += amp * _value_noise(u, v, cell, salt * 16 + o)
        tot += amp
        amp *= 0.6
        cell *= 0.5
    t /= tot
    # blocky patches: piecewise-constant cells of random brightness (two sizes) give FAST-style corners
    for k, (cell, gate, mix) in enumerate(((0.35, 1.4, 0.45), (0.09, 0.6, 0.35))):
        blk = _hash01(np.floor(u / (base_cell * cell)).astype(np.int32), np.floor(v / (base_cell * cell)).astype(np.int32),
                      salt * 16 + 9 + 2 * k)
        sel = _hash01(np.floor(u / (base_cell * gate)).astype(np.int32), np.floor(v / (base_cell * gate)).astype(np.int32),
                      salt * 16 + 10 + 2 * k)
        t = np.where(sel > 0.45, (1 - mix) * t + mix * blk, t)
    return 255.0 * np.clip(0.1 + 0.9 * t, 0, 1)


class Scene:
    """Ground plane y = CAM_HEIGHT (camera frame: x right, y down, z forward) and fronto-parallel textured boxes."""

    def __init__(self, seed=0, n_boxes=60):
        rng = np.random.default_rng(SEED0 + seed)
        self.seed = seed
        z = rng.uniform(5.0, 60.0, n_boxes)
        x = rng.uniform(-1.0, 1.0, n_boxes) * (0.25 * z + 3.0)
        w = rng.uniform(1.0, 4.0, n_boxes)
        h = rng.uniform(1.0, 3.5, n_boxes)
        self.boxes = np.stack([x - w / 2, x + w / 2, CAM_HEIGHT - h, np.full(n_boxes, CAM_HEIGHT), z], 1)
        self.salts = rng.integers(1, 1 << 20, n_boxes + 1)

    def render(self, cam_x=0.0, width=WIDTH, height=HEIGHT, noise_seed=0, cam_z=0.0):
        """8-bit image and depth (z) map seen from a camera translated by cam_x along +x and by cam_z along +z (forward)."""
        uu, vv = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
        dx, dy = (uu - np.float32(CX)) / np.float32(FX), (vv - np.float32(CY)) / np.float32(FY)  # ray (dx, dy, 1)
        depth = np.full((height, width), 1e6, np.float32)
        img = (200.0 - 40.0 * (vv / np.float32(height))).astype(np.float32)  # sky with a faint gradient
        r0 = min(max(int(np.ceil(CY + FY * CAM_HEIGHT / 120.0)), 0), height)  # first row whose ground hit is < 120 m
        if r0 < height:
            zg = (np.float32(CAM_HEIGHT) / dy[r0:]).astype(np.float32)
            X = (np.float32(cam_x) + dx[r0:] * zg).astype(np.float32)
            img[r0:] = _texture(X, (zg + np.float32(cam_z)).astype(np.float32), int(self.salts[-1]), 2.0)
            depth[r0:] = zg
        order = np.argsort(-self.boxes[:, 4], kind="stable")  # far to near: nearer boxes overwrite
        for k in order:
            x0, x1, y0, y1, zb = self.boxes[k]
            zb = zb - cam_z  # depth of the box in front of the moved camera
            if zb < 0.5:
                continue
            c0 = int(np.floor(CX + FX * (x0 - cam_x) / zb)) - 1
            c1 = int(np.ceil(CX + FX * (x1 - cam_x) / zb)) + 2
            q0 = int(np.floor(CY + FY * y0 / zb)) - 1
            q1 = int(np.ceil(CY + FY * y1 / zb)) + 2
            c0, c1, q0, q1 = max(c0, 0), min(c1, width), max(q0, 0), min(q1, height)
            if c0 >= c1 or q0 >= q1:
                continue
            sl = (slice(q0, q1), slice(c0, c1))
            X = (np.float32(cam_x) + dx[sl] * np.float32(zb)).astype(np.float32)
            Y = (dy[sl] * np.float32(zb)).astype(np.float32)
            hit = (X >= x0) & (X <= x1) & (Y >= y0) & (Y <= y1) & (np.float32(zb) < depth[sl])
            if not hit.any():
                continue
            tex = _texture((X - np.float32(x0)).astype(np.float32), (Y - np.float32(y0)).astype(np.float32),
                           int(self.salts[k]), 0.8)
            img[sl] = np.where(hit, tex, img[sl])
            depth[sl] = np.where(hit, np.float32(zb), depth[sl])
        rng = np.random.default_rng([SEED0 + self.seed, noise_seed])
        img = img + rng.normal(0.0, 2.0, img.shape).astype(np.float32)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth


def stereo_pair(seed=0, width=WIDTH, height=HEIGHT):
    """Left/right 8-bit images of scene `seed` (right camera at +baseline along x)."""
    sc = Scene(seed)
    left, _ = sc.render(0.0, width, height, noise_seed=1)
    right, _ = sc.render(BASELINE, width, height, noise_seed=2)
    return left, right


def stereo_batch(n_frames, seed=0, width=WIDTH, height=HEIGHT):
    """[n_frames, 2, height, width] uint8: frame f is stereo_pair(seed + f)."""
    out = np.empty((n_frames, 2, height, width), np.uint8)
    for f in range(n_frames):
        out[f, 0], out[f, 1] = stereo_pair(seed + f, width, height)
    return out


# ---- 64-beam scans ---------------------------------------------------------------------
# LiDAR axes: x forward, y left, z up; the sensor sits at the left camera centre.  With camera axes x right,
# y down, z forward:  p_cam = R_CL p_lidar,  R_CL = [[0,-1,0],[0,0,-1],[1,0,0]].
N_BEAMS, N_AZIMUTH = 64, 2048
R_CAM_FROM_LIDAR = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
VELODYNE_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("intensity", "<f4"),
                           ("time", "<f4"), ("ring", "<u2"), ("pad1", "<u2"), ("pad2", "<f4")])
POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("normal_x", "<f4"),
                        ("normal_y", "<f4"), ("normal_z", "<f4"), ("pad1", "<f4"), ("intensity", "<f4"),
                        ("curvature", "<f4"), ("pad2", "<f4"), ("pad3", "<f4")])
assert VELODYNE_DTYPE.itemsize == 32 and POINT_DTYPE.itemsize == 48


def sensor_pose(frame, speed=10.0, rate=10.0):
    """World-from-LiDAR pose of frame `frame`: 10 m/s forward with a slow yaw sinusoid.  Returns (R 3x3, t 3)."""
    t = frame / rate
    yaw = 0.05 * np.sin(0.1 * 2 * np.pi * t)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    return R, np.array([speed * t, 0.3 * np.sin(0.2 * t), 0.0])


def lidar_scan(scene, frame=0, max_range=100.0, noise=0.02):
    """One HDL-64E-like sweep of `scene` from sensor_pose(frame): structured array of velodyne_ros::Point (32 B),
    azimuth-major firing order, no-return rays dropped; `time` in microseconds over the 0.1 s sweep."""
    R, t = sensor_pose(frame)
    elev = np.deg2rad(np.linspace(2.0, -24.8, N_BEAMS))
    azim = np.linspace(0.0, 2 * np.pi, N_AZIMUTH, endpoint=False)
    az, el = np.meshgrid(azim, elev, indexing="ij")  # [azimuth, beam]
    d_l = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
    d_w = d_l @ R.T  # world frame = LiDAR frame of frame 0 (x forward, y left, z up)
    best = np.full(len(d_w), np.inf)
    # ground z = -CAM_HEIGHT
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (-CAM_HEIGHT - t[2]) / d_w[:, 2]
        tg = np.where((d_w[:, 2] < 0) & (tg > 0), tg, np.inf)
        best = np.minimum(best, tg)
        # boxes: fronto-parallel faces; camera (x right, y down, z fwd) -> world x = z_cam, y = -x_cam, z = -y_cam
        for x0, x1, y0, y1, zb in scene.boxes:
            tb = (zb - t[0]) / d_w[:, 0]
            hy = t[1] + tb * d_w[:, 1]
            hz = t[2] + tb * d_w[:, 2]
            ok = (tb > 0) & (-hy >= x0) & (-hy <= x1) & (-hz >= y0) & (-hz <= y1)
            best = np.minimum(best, np.where(ok, tb, np.inf))
        # side walls y = +-15 m, 6 m high (a street canyon), so that there are vertical planes all along the way
        for wy in (-15.0, 15.0):
            tw = (wy - t[1]) / d_w[:, 1]
            hz = t[2] + tw * d_w[:, 2]
            ok = (tw > 0) & (hz >= -CAM_HEIGHT) & (hz <= 6.0)
            best = np.minimum(best, np.where(ok, tw, np.inf))
    rng = np.random.default_rng([SEED0 + scene.seed, 77, frame])
    rngd = best + rng.normal(0.0, noise, best.shape)
    keep = np.isfinite(best) & (best <= max_range)
    pts = d_l * rngd[:, None]
    out = np.zeros(int(keep.sum()), VELODYNE_DTYPE)
    out["x"], out["y"], out["z"] = pts[keep, 0], pts[keep, 1], pts[keep, 2]
    out["intensity"] = (rng.random(len(best))[keep] * 100).astype(np.float32)
    idx = np.nonzero(keep)[0]
    out["time"] = ((idx // N_BEAMS) / N_AZIMUTH * 1.0e5).astype(np.float32)
    out["ring"] = (idx % N_BEAMS).astype(np.uint16)
    return out


def lidar_map(scene, x_from=-700.0, x_to=300.0, voxel=0.5, seed=0):
    """The accumulated LiDAR map of a drive along the street of `scene`, as map_incremental leaves it: one point per `voxel`-sized cell
    (LidarFrontEnd.cpp:387-435 keeps the point nearest the cell centre) on the ground, the two side walls and the box faces between
    x_from and x_to in the LiDAR world frame (x forward, y left, z up), each jittered inside its cell plus the range noise.  The reference
    holds 10^5 - 10^6 such points; 1000 m of street gives about 1.9 * 10^5."""
    rng = np.random.default_rng([SEED0 + scene.seed, 991, seed])
    parts = []

    def lattice(a0, a1, b0, b1):
        a = np.arange(np.floor(a0 / voxel), np.ceil(a1 / voxel)) * voxel
        b = np.arange(np.floor(b0 / voxel), np.ceil(b1 / voxel)) * voxel
        A, B = np.meshgrid(a, b, indexing="ij")
        A = A.reshape(-1) + rng.uniform(0.05, voxel - 0.05, A.size)
        B = B.reshape(-1) + rng.uniform(0.05, voxel - 0.05, B.size)
        return A, B

    gx, gy = lattice(x_from, x_to, -15.0, 15.0)  # ground z = -CAM_HEIGHT
    parts.append(np.stack([gx, gy, np.full(gx.size, -CAM_HEIGHT) + rng.normal(0, 0.02, gx.size)], 1))
    for wy in (-15.0, 15.0):                      # side walls, 6 m high
        wx, wz = lattice(x_from, x_to, -CAM_HEIGHT, 6.0)
        parts.append(np.stack([wx, np.full(wx.size, wy) + rng.normal(0, 0.02, wx.size), wz], 1))
    for x0, x1, y0, y1, zb in scene.boxes:        # box faces: plane x = zb, y in [-x1, -x0], z in [-y1, -y0]
        if not (x_from <= zb <= x_to):
            continue
        by, bz = lattice(-x1, -x0, -y1, -y0)
        parts.append(np.stack([np.full(by.size, zb) + rng.normal(0, 0.02, by.size), by, bz], 1))
    xyz = np.concatenate(parts).astype(np.float32)
    out = np.zeros(len(xyz), POINT_DTYPE)
    out["x"], out["y"], out["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out["pad0"] = 1.0
    return out[rng.permutation(len(out))]


def lidar_state(frame):
    """state_point of LidarFrontEnd.cpp (rot, pos of the body in the LiDAR world frame; identity LiDAR-IMU offset)
    as four float64 arrays: rot[9] (row-major), pos[3], offset_R_L_I[9], offset_T_L_I[3]."""
    R, t = sensor_pose(frame)
    return R.reshape(-1).copy(), t.copy(), np.eye(3).reshape(-1).copy(), np.zeros(3)


# ---- bundle-adjustment windows -------------------------------------------------------------
def _quat_from_rot(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = (R[2, 1] - R[1, 2]) / (4 * w); y = (R[0, 2] - R[2, 0]) / (4 * w); z = (R[1, 0] - R[0, 1]) / (4 * w)
    return np.array([x, y, z, w])


def _rot_from_rvec(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def ba_window(seed=0, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.5, 0.05), point_noise=0.01, outlier_frac=0.05,
              mono_frac=0.1, pixel_sigma=1.0):
    """Synthetic local-BA window: keyframes on a forward trajectory (camera frame: x right, y down, z forward), points
    in front of them, stereo observations with pixel noise scaled by the pyramid level, a few gross outliers and
    monocular observations.  Returns a dict of flat arrays in the layout the C ABI takes:
    poses [K,7] (qx,qy,qz,qw,tx,ty,tz = Tcw), fixed [K], points [P,3], edges [E,6] (point, pose, u, v, uR, invSigma2),
    cam (fx, fy, cx, cy, bf), plus the noise-free truth."""
    rng = np.random.default_rng([SEED0, 0xBA, seed])
    K = n_opt + n_fix
    # fixed keyframes first in time (older), optimisable ones last; ids ascending = vertex order in g2o
    Rs, ts = [], []
    for k in range(K):
        yaw = 0.03 * np.sin(0.4 * k)
        Rwc = _rot_from_rvec(np.array([0.0, yaw, 0.0]))
        twc = np.array([0.3 * np.sin(0.2 * k), 0.02 * np.cos(0.3 * k), 1.0 * k])
        Rcw = Rwc.T
        Rs.append(Rcw); ts.append(-Rcw @ twc)
    # points: in a corridor ahead of the trajectory
    z = rng.uniform(2.0, K + 40.0, n_points)
    pts = np.stack([rng.uniform(-12, 12, n_points), rng.uniform(-3, 1.65, n_points), z], 1)
    scale = 1.2 ** np.arange(8)
    edges, truth_uv = [], []
    for p in range(n_points):
        for k in range(K):
            pc = Rs[k] @ pts[p] + ts[k]
            if pc[2] < 1.0 or pc[2] > 45.0:
                continue
            u = FX * pc[0] / pc[2] + CX
            v = FY * pc[1] / pc[2] + CY
            if not (20 < u < WIDTH - 20 and 20 < v < HEIGHT - 20):
                continue
            if rng.random() > 0.55:  # not every keyframe that could see a point has matched it
                continue
            lvl = int(min(7, max(0, np.floor(np.log(max(pc[2], 1.0) / 6.0) / np.log(1.2) + 3))))
            s = pixel_sigma * scale[lvl]
            uo, vo = u + rng.normal(0, s), v + rng.normal(0, s)
            ur = uo - BF / pc[2] + rng.normal(0, s)
            if rng.random() < outlier_frac:
                uo += rng.choice([-1, 1]) * rng.uniform(8, 20); vo += rng.choice([-1, 1]) * rng.uniform(8, 20)
            if rng.random() < mono_frac:
                ur = -1.0
            edges.append((p, k, np.float32(uo), np.float32(vo), np.float32(ur), np.float32(1.0) / np.float32(scale[lvl] ** 2)))
    edges = np.array(edges, np.float64)
    # keep points with at least 2 observations (as local mapping would)
    cnt = np.bincount(edges[:, 0].astype(int), minlength=n_points)
    keep = cnt >= 2
    remap = -np.ones(n_points, int); remap[keep] = np.arange(keep.sum())
    edges = edges[keep[edges[:, 0].astype(int)]]
    edges[:, 0] = remap[edges[:, 0].astype(int)]
    pts_true = pts[keep]
    poses_true = np.array([np.concatenate([_quat_from_rot(Rs[k]), ts[k]]) for k in range(K)])
    fixed = np.zeros(K, np.uint8); fixed[:n_fix] = 1
    poses = poses_true.copy()
    for k in range(n_fix, K):
        dR = _rot_from_rvec(rng.normal(0, np.deg2rad(pose_noise[0]), 3))
        Rn = dR @ Rs[k]
        poses[k] = np.concatenate([_quat_from_rot(Rn), ts[k] + rng.normal(0, pose_noise[1], 3)])
    # estimates live in float on the map (Sophus::SE3f / Eigen::Vector3f): round through float32
    poses = poses.astype(np.float32).astype(np.float64)
    pts_noisy = (pts_true * (1 + rng.normal(0, point_noise, (len(pts_true), 1)))).astype(np.float32).astype(np.float64)
    cam = np.array([np.float32(FX), np.float32(FY), np.float32(CX), np.float32(CY), np.float32(BF)], np.float64)
    return dict(poses=poses, fixed=fixed, points=pts_noisy, edges=edges, cam=cam, poses_true=poses_true, points_true=pts_true)


def ba_window_varied(seed=0):
    """A local-BA window drawn from the spread the reference's windows have (round 6): the keyframe that triggers local
    mapping takes ALL its covisible keyframes (OptimizerWithLidar.cc:63-76), fixes whoever else sees their points (:106-123), and gets the
    LiDAR edge only if more than two of them carry a surface cloud, over at most six (:226-241).  Drawn per window: 4-24 free and 2-40 fixed
    keyframes, 500-6000 points (log-uniform), 0-15 % gross outliers, pose noise 0.2-2 x the benched window's (0.1 deg, 1 cm), a LiDAR window of
    0 (no edge) or 3-6 clouds of 1500-3000 points, its weight 1 -- or, one window in sixteen, 1000: the optimiser then rejects steps (the edge's
    gradient lacks the residual) -- and one in sixteen is INTERRUPTED after 2-7 iterations: the reference's tracking thread raises
    mbAbortBA when the next keyframe waits (LocalMapping.cc:302, 908) and g2o's terminate() ends the loop; the count stands in for the moment the
    flag arrives, so that the run stays deterministic.  (Windows that end by the `_nBad >= 3` rule do not occur at these sizes: from
    tau = 1e-5 the damping starts at ~500 and the robust cost still falls by 0.3 % in the tenth iteration -- measured with the oracle.)
    The same layout as ba_window() plus win_pose / clouds / weight; vectorised, its own random stream (ba_window's stays as the fixtures know it)."""
    rng = np.random.default_rng([SEED0, 0xBA5, seed])
    n_opt, n_fix = int(rng.integers(4, 25)), int(rng.integers(2, 41))
    n_points = int(np.exp(rng.uniform(np.log(500.0), np.log(6000.0))))
    outlier_frac = float(rng.uniform(0.0, 0.15))
    kind = min(int(rng.integers(0, 16)), 2)   # 0: heavy LiDAR edge (one window in 16), 1: interrupted (one in 16), 2: ordinary
    noise_scale = float(rng.uniform(0.2, 2.0))
    iterations = int(rng.integers(2, 8)) if kind == 1 else 10
    pose_noise = (0.1 * noise_scale, 0.01 * noise_scale)
    W = int(rng.choice([0, 3, 4, 5, 6], p=[0.15, 0.15, 0.2, 0.2, 0.3]))
    W = W if W <= n_opt else (n_opt if n_opt >= 3 else 0)
    if kind == 0 and W == 0:
        W = min(n_opt, 5)
    cloud_points = int(rng.integers(1500, 3001))
    K = n_opt + n_fix
    k = np.arange(K)
    yaw = 0.03 * np.sin(0.4 * k)
    Rcw = np.stack([_rot_from_rvec(np.array([0.0, y, 0.0])).T for y in yaw])
    twc = np.stack([0.3 * np.sin(0.2 * k), 0.02 * np.cos(0.3 * k), 1.0 * k], 1)
    tcw = -np.einsum("kij,kj->ki", Rcw, twc)
    pts = np.stack([rng.uniform(-12, 12, n_points), rng.uniform(-3, 1.65, n_points), rng.uniform(2.0, K + 40.0, n_points)], 1)
    pc = np.einsum("kij,pj->pki", Rcw, pts) + tcw[None]                      # [P, K, 3]
    z = pc[..., 2]
    zs = np.where(z > 0.5, z, 1.0)
    u, v = FX * pc[..., 0] / zs + CX, FY * pc[..., 1] / zs + CY
    seen = (z >= 1.0) & (z <= 45.0) & (u > 20) & (u < WIDTH - 20) & (v > 20) & (v < HEIGHT - 20) & (rng.random((n_points, K)) <= 0.55)
    lvl = np.clip(np.floor(np.log(np.maximum(zs, 1.0) / 6.0) / np.log(1.2) + 3), 0, 7).astype(int)
    sig = (1.2 ** np.arange(8))[lvl]
    uo, vo = u + rng.normal(0, 1, u.shape) * sig, v + rng.normal(0, 1, u.shape) * sig
    ur = uo - BF / zs + rng.normal(0, 1, u.shape) * sig
    out = rng.random(u.shape) < outlier_frac
    uo = uo + out * rng.choice([-1.0, 1.0], u.shape) * rng.uniform(8, 20, u.shape)
    vo = vo + out * rng.choice([-1.0, 1.0], u.shape) * rng.uniform(8, 20, u.shape)
    ur = np.where(rng.random(u.shape) < 0.1, -1.0, ur)
    keep = seen.sum(1) >= 2                                                  # points with at least two observations, as local mapping keeps them
    seen &= keep[:, None]
    remap = -np.ones(n_points, int); remap[keep] = np.arange(keep.sum())
    pi, ki = np.nonzero(seen)                                                # point-major, keyframes ascending: ba_window's edge order
    edges = np.stack([remap[pi], ki, np.float32(uo[pi, ki]), np.float32(vo[pi, ki]), np.float32(ur[pi, ki]),
                      np.float32(1.0) / np.float32(sig[pi, ki] ** 2)], 1).astype(np.float64)
    pts_true = pts[keep]
    poses_true = np.array([np.concatenate([_quat_from_rot(Rcw[j]), tcw[j]]) for j in range(K)])
    fixed = np.zeros(K, np.uint8); fixed[:n_fix] = 1
    poses = poses_true.copy()
    for j in range(n_fix, K):
        dR = _rot_from_rvec(rng.normal(0, np.deg2rad(pose_noise[0]), 3))
        poses[j] = np.concatenate([_quat_from_rot(dR @ Rcw[j]), tcw[j] + rng.normal(0, pose_noise[1], 3)])
    poses = poses.astype(np.float32).astype(np.float64)
    pts_noisy = (pts_true * (1 + rng.normal(0, 0.01 * min(noise_scale, 1.0), (len(pts_true), 1)))).astype(np.float32).astype(np.float64)
    cam = np.array([np.float32(FX), np.float32(FY), np.float32(CX), np.float32(CY), np.float32(BF)], np.float64)
    w = dict(poses=poses, fixed=fixed, points=pts_noisy, edges=edges, cam=cam, poses_true=poses_true, points_true=pts_true,
             params=dict(n_opt=n_opt, n_fix=n_fix, n_points=int(keep.sum()), outlier_frac=outlier_frac, noise_scale=noise_scale, lidar_keyframes=W,
                         iterations=iterations, kind=("heavy LiDAR edge", "interrupted")[kind] if kind < 2 else "ordinary"))
    w["win_pose"] = list(range(K - 1, K - 1 - W, -1))
    w["clouds"] = ba_window_clouds(w, w["win_pose"], n_points=cloud_points, seed=seed) if W else []
    w["weight"] = 1000.0 if kind == 0 else 1.0
    w["iterations"] = iterations
    return w


# ---- LiDAR clouds for the BALM term of the local BA -----------------------------------------------------------------
# KITTI-00 camera <- LiDAR extrinsic used by the synthetic windows: LiDAR axes x forward / y left / z up, camera axes
# x right / y down / z forward, LiDAR 0.27 m behind and 0.08 m above the camera.
TCL7 = np.array([0.5, -0.5, 0.5, 0.5, 0.0, -0.08, -0.27], np.float32)  # (qx, qy, qz, qw, tx, ty, tz) of Tcl


def _quat_R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def ba_window_clouds(w, win_pose, n_points=3000, noise=0.02, seed=0):
    """Surface clouds (LiDAR frame, float32 [n, 3]) for the keyframes `win_pose` of a ba_window(): samples of the ground
    plane y = CAM_HEIGHT and of two walls x = -7 / x = +9 (camera-world axes) around each TRUE keyframe pose."""
    rng = np.random.default_rng([SEED0, 0xC10D, seed])
    Rcl, tcl = _quat_R(TCL7[:4]), TCL7[4:].astype(np.float64)
    clouds = []
    for k in win_pose:
        q, t = w["poses_true"][k][:4], w["poses_true"][k][4:]
        Rcw = _quat_R(q)
        cz = -(Rcw.T @ t)  # camera centre in the world
        m = n_points // 3
        g = np.stack([rng.uniform(-4, 4, m) + cz[0], np.full(m, CAM_HEIGHT), rng.uniform(2, 12, m) + cz[2]], 1)
        wl = np.stack([np.full(m, -7.0), rng.uniform(-1.5, CAM_HEIGHT, m), rng.uniform(2, 12, m) + cz[2]], 1)
        wr = np.stack([np.full(n_points - 2 * m, 9.0), rng.uniform(-1.5, CAM_HEIGHT, n_points - 2 * m), rng.uniform(2, 12, n_points - 2 * m) + cz[2]], 1)
        Xw = np.concatenate([g, wl, wr]) + rng.normal(0, noise, (n_points, 3))
        Xc = Xw @ Rcw.T + t
        Xl = (Xc - tcl) @ Rcl  # Tlc * Xc
        clouds.append(Xl.astype(np.float32))
    return clouds


IMU_SAMPLE_DTYPE = np.dtype([("t", "<f8"), ("a", "<f4", (3,)), ("w", "<f4", (3,))])
# config/Camera-Inertial-Lidar/KITTI04-12.yaml:50-54 (NoiseGyro, NoiseAcc, GyroWalk, AccWalk, Frequency 100 Hz): Calib::Set gets
# ng = NoiseGyro * sqrt(f), na = NoiseAcc * sqrt(f), ngw = GyroWalk / sqrt(f), naw = AccWalk / sqrt(f) (SF/src/Tracking.cc ParseIMUParamFile)
IMU_NOISE = (1.6968e-04 * 10.0, 2.0e-03 * 10.0, 1.9393e-05 / 10.0, 3.0e-03 / 10.0)


def imu_samples(t0, t1, rate=100.0, seed=0, noise=True, omega=(0.02, -0.1, 0.05), acc_body=(0.3, 0.1, 9.7), jitter=0.0):
    """IMU samples covering [t0, t1] with one sample before t0 and one after t1 (what mvImuFromLastFrame holds): a smooth
    angular velocity / specific force signal plus white noise."""
    rng = np.random.default_rng([SEED0, 0x1A0, seed])
    k0, k1 = int(np.floor(t0 * rate)) - (0 if jitter else 0), int(np.ceil(t1 * rate)) + 1
    t = np.arange(k0, k1 + 1) / rate + (rng.uniform(-jitter, jitter, k1 - k0 + 1) / rate if jitter else 0.0)
    out = np.zeros(len(t), IMU_SAMPLE_DTYPE)
    out["t"] = t
    ph = 2 * np.pi * 0.7 * t
    w = np.stack([omega[0] + 0.05 * np.sin(ph), omega[1] + 0.03 * np.cos(1.3 * ph), omega[2] + 0.04 * np.sin(0.6 * ph)], 1)
    a = np.stack([acc_body[0] + 0.5 * np.sin(1.1 * ph), acc_body[1] + 0.3 * np.cos(ph), acc_body[2] + 0.2 * np.sin(0.4 * ph)], 1)
    if noise:
        w += rng.normal(0, IMU_NOISE[0], w.shape)
        a += rng.normal(0, IMU_NOISE[1], a.shape)
    out["w"], out["a"] = w.astype(np.float32), a.astype(np.float32)
    return out


# ---- visual-inertial local BA windows -------------------------------------------------------------------------------------
# body frame: x forward, y left, z up (world gravity = (0, 0, -9.81)); camera: z forward, x right, y down
RBC = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
TBC = np.array([0.3, 0.05, 0.1])


def _traj(t):
    """Analytic body trajectory -> (p, R) at time t (scalar)."""
    from scipy.spatial.transform import Rotation
    p = np.array([8.0 * t + 0.5 * np.sin(0.8 * t), 1.5 * np.sin(0.5 * t), 0.2 * np.sin(0.7 * t)])
    R = Rotation.from_euler("zyx", [0.25 * np.sin(0.4 * t), 0.05 * np.sin(0.9 * t), 0.04 * np.cos(0.6 * t)]).as_matrix()
    return p, R


def inertial_window(seed=0, n_opt=8, n_points=600, kf_dt=0.4, rate=200.0, pose_noise=(0.3, 0.03), vel_noise=0.05, pixel_noise=0.5,
                    outlier_frac=0.02):
    """A LocalInertialBA problem: 1 fixed keyframe + n_opt optimisable ones (vertex-id order = time order), pre-integrated
    IMU between consecutive keyframes (from noisy samples of the analytic trajectory), stereo observations of random points.
    Returns a dict: kf33 [K, 33] (noisy estimate), kf33_true, fixed, has_imu, calib24, points, points_true, edges [E, 6],
    link4 [L, 4], samples (per link: IMU_SAMPLE_DTYPE array, t1, t2), bias6 (used for the integration), cam."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng([SEED0, 0x1BA, seed])
    K = n_opt + 1
    times = 2.0 + kf_dt * np.arange(K)
    h = 1e-4
    Rcb, tcb = RBC.T, -RBC.T @ TBC
    calib24 = np.concatenate([Rcb.ravel(), tcb, RBC.ravel(), TBC])
    bg_true, ba_true = np.array([0.002, -0.001, 0.0015]), np.array([0.03, -0.02, 0.01])

    def state(t):
        p, R = _traj(t)
        v = (_traj(t + h)[0] - _traj(t - h)[0]) / (2 * h)
        Rcw = Rcb @ R.T
        tcw = Rcb @ (-R.T @ p) + tcb
        return p, R, v, Rcw, tcw

    def pack(p, R, v, Rcw, tcw, bg, ba):
        return np.concatenate([Rcw.ravel(), tcw, R.ravel(), p, v, bg, ba])

    kf_true = np.stack([pack(*state(t), bg_true, ba_true) for t in times])
    kf = kf_true.copy()
    for k in range(1, K):  # keyframe 0 is the fixed one: exact
        p, R, v, _, _ = state(times[k])
        Rn = R @ Rotation.from_rotvec(rng.normal(0, np.deg2rad(pose_noise[0]), 3)).as_matrix()
        pn = p + rng.normal(0, pose_noise[1], 3)
        Rcw = Rcb @ Rn.T
        tcw = Rcb @ (-Rn.T @ pn) + tcb
        kf[k] = pack(pn, Rn, v + rng.normal(0, vel_noise, 3), Rcw, tcw, bg_true + rng.normal(0, 2e-4, 3), ba_true + rng.normal(0, 3e-3, 3))
    # the map keeps floats
    kf = kf.astype(np.float32).astype(np.float64)
    kf_true32 = kf_true.astype(np.float32).astype(np.float64)
    kf[0] = kf_true32[0]
    # IMU samples between consecutive keyframes
    g = np.array([0.0, 0.0, -9.81])
    samples, link4 = [], []
    for k in range(1, K):
        t1, t2 = times[k - 1], times[k]
        ts = np.arange(np.floor(t1 * rate) - 0, np.ceil(t2 * rate) + 1) / rate
        ts = ts[(ts >= t1 - 1.0 / rate - 1e-9) & (ts <= t2 + 1.0 / rate + 1e-9)]
        out = np.zeros(len(ts), IMU_SAMPLE_DTYPE)
        for i, t in enumerate(ts):
            p0, R0 = _traj(t)
            acc_w = (_traj(t + h)[0] - 2 * p0 + _traj(t - h)[0]) / (h * h)
            Rm, Rp = _traj(t - h)[1], _traj(t + h)[1]
            W = R0.T @ (Rp - Rm) / (2 * h)
            w_b = np.array([W[2, 1], W[0, 2], W[1, 0]])
            out["t"][i] = t
            out["a"][i] = R0.T @ (acc_w - g) + ba_true + rng.normal(0, IMU_NOISE[1] * 0.1, 3)
            out["w"][i] = w_b + bg_true + rng.normal(0, IMU_NOISE[0] * 0.1, 3)
        samples.append((out, t1, t2))
        link4.append([k - 1, k, 1.0 if k == 1 else 0.0, 1e-2 if k == 1 else 1.0])  # the link to the fixed keyframe is robust and down-weighted
    # points in front of the trajectory and their stereo observations
    cam = np.array([FX, FY, CX, CY, BF], np.float64)
    cam = np.float32(cam).astype(np.float64)
    pts_true = np.stack([rng.uniform(10, 8.0 * times[-1] + 40, n_points), rng.uniform(-12, 12, n_points), rng.uniform(-1.5, 4, n_points)], 1)
    edges = []
    used = np.zeros(n_points, bool)
    for k in range(K):
        _, _, _, Rcw, tcw = state(times[k])
        Xc = pts_true @ Rcw.T + tcw
        z = Xc[:, 2]
        u = cam[0] * Xc[:, 0] / z + cam[2]
        v = cam[1] * Xc[:, 1] / z + cam[3]
        vis = (z > 2) & (z < 60) & (u > 20) & (u < WIDTH - 20) & (v > 20) & (v < HEIGHT - 20)
        for i in np.nonzero(vis)[0]:
            octave = int(min(7, max(0, np.log(z[i] / 6.0) / np.log(1.2))))
            s2 = 1.2 ** (2 * octave)
            du, dv = rng.normal(0, pixel_noise * np.sqrt(s2), 2)
            if rng.random() < outlier_frac:
                du += rng.choice([-1, 1]) * rng.uniform(8, 20)
            ur = u[i] + du - cam[4] / z[i] + rng.normal(0, 0.3)
            mono = z[i] > 45
            edges.append([i, k, np.float32(u[i] + du), np.float32(v[i] + dv), -1.0 if mono else np.float32(ur), 1.0 / s2])
            used[i] = True
    edges = np.array(edges, np.float64)
    nobs = np.bincount(edges[:, 0].astype(int), minlength=n_points)
    keep = nobs >= 2
    remap = np.cumsum(keep) - 1
    edges = edges[keep[edges[:, 0].astype(int)]]
    edges[:, 0] = remap[edges[:, 0].astype(int)]
    pts_true = pts_true[keep]
    pts = (pts_true + rng.normal(0, 0.05, pts_true.shape) * (1 + np.linalg.norm(pts_true - _traj(times[K // 2])[0], axis=1, keepdims=True) / 20)).astype(np.float32).astype(np.float64)
    fixed = np.zeros(K, np.uint8); fixed[0] = 1
    return dict(kf33=kf, kf33_true=kf_true32, fixed=fixed, has_imu=np.ones(K, np.uint8), calib24=calib24, points=pts, points_true=pts_true,
                edges=edges, link4=np.array(link4), samples=samples, bias6=np.concatenate([ba_true, bg_true]).astype(np.float32), cam=cam,
                times=times)


def pose_inertial_problem(seed=0, n_points=500, frame_dt=0.1, last_frame=False, outlier_frac=0.04, mono_frac=0.15):
    """One Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame problem: the frame (noisy state, kf33 layout), the other state (last
    keyframe: exact, fixed; previous frame: slightly off, free, with a prior mpcpi), the IMU samples between them, the map points the
    frame holds (fixed), their observations, the close flags (mTrackDepth < 10) and the camera.
    -> dict(cur33, other33, cur33_true, prior246 | None, calib24, samples, t1, t2, bias6, Xw, edges [E, 6], close, cam)."""
    from scipy.spatial.transform import Rotation
    w = inertial_window(1000 + seed, n_opt=1, n_points=n_points, kf_dt=frame_dt, pose_noise=(0.4, 0.05), vel_noise=0.08, outlier_frac=0.0)
    rng = np.random.default_rng([SEED0, 0x9014, seed])
    e = w["edges"][w["edges"][:, 1] == 1].copy()
    ids, inv = np.unique(e[:, 0].astype(int), return_inverse=True)
    e[:, 0], e[:, 1] = inv, 0
    Xw = (w["points_true"][ids] + rng.normal(0, 0.02, (len(ids), 3))).astype(np.float32).astype(np.float64)
    bad = rng.random(len(e)) < outlier_frac
    e[bad, 2] += rng.choice([-1, 1], bad.sum()) * rng.uniform(8, 25, bad.sum())
    e[bad, 3] += rng.choice([-1, 1], bad.sum()) * rng.uniform(8, 25, bad.sum())
    mono = rng.random(len(e)) < mono_frac
    e[mono, 4] = -1.0
    Rcw, tcw = w["kf33_true"][1][:9].reshape(3, 3), w["kf33_true"][1][9:12]
    depth = (Xw[e[:, 0].astype(int)] @ Rcw.T + tcw)[:, 2]
    close = (depth < 10.0).astype(np.uint8)
    other = w["kf33_true"][0].copy()
    prior = None
    if last_frame:
        # the previous frame is an estimate too (and moves a little); its prior = that estimate with a plausible information matrix
        Rcb, tcb = w["calib24"][:9].reshape(3, 3), w["calib24"][9:12]
        R = other[12:21].reshape(3, 3) @ Rotation.from_rotvec(rng.normal(0, np.deg2rad(0.05), 3)).as_matrix()
        p = other[21:24] + rng.normal(0, 0.01, 3)
        other[12:21], other[21:24] = R.ravel(), p
        other[:9], other[9:12] = (Rcb @ R.T).ravel(), Rcb @ (-R.T @ p) + tcb
        other[24:27] += rng.normal(0, 0.02, 3)
        other = other.astype(np.float32).astype(np.float64)
        A = rng.normal(0, 1, (15, 15))
        d = np.concatenate([np.full(3, 3e4), np.full(3, 3e3), np.full(3, 4e2), np.full(3, 1e6), np.full(3, 1e4)])
        H = np.diag(d) + 0.02 * np.sqrt(np.outer(d, d)) * (A + A.T) / 2
        H = (H + H.T) / 2
        prior = np.concatenate([other[12:21], other[21:24], other[24:27], other[27:30], other[30:33], H.ravel()])
    s, t1, t2 = w["samples"][0]
    return dict(cur33=w["kf33"][1].copy(), other33=other, cur33_true=w["kf33_true"][1].copy(), prior246=prior, calib24=w["calib24"], samples=s, t1=t1, t2=t2,
                bias6=w["bias6"], Xw=Xw, edges=e, close=close, cam=w["cam"], gross=bad)


def imu_init_problem(seed=0, n_kf=12, kf_dt=0.4, tilt_deg=25.0, pos_noise=0.01):
    """An IMU initialisation problem (LocalMapping::InitializeIMU): n_kf keyframes of the analytic trajectory seen in a visual world W that
    is rotated against the gravity-aligned frame (the map before the IMU is initialised knows nothing of gravity), slightly noisy
    positions, IMU samples between consecutive keyframes integrated at ZERO bias (the true biases are what the optimisation finds).
    -> dict(Rwb [N, 3, 3], twb [N, 3] in W, samples (per link: array, t1, t2), Rwg_true (gravity in W = Rwg_true @ (0, 0, -9.81)),
    vel_true [N, 3] in W, bg_true, ba_true)."""
    from scipy.spatial.transform import Rotation
    w = inertial_window(2000 + seed, n_opt=n_kf - 1, n_points=20, kf_dt=kf_dt)
    rng = np.random.default_rng([SEED0, 0x1217, seed])
    axis = rng.normal(0, 1, 3)
    axis /= np.linalg.norm(axis)
    Rwg = Rotation.from_rotvec(np.deg2rad(tilt_deg) * axis).as_matrix()
    kt = w["kf33_true"]
    Rwb = np.stack([Rwg @ k[12:21].reshape(3, 3) for k in kt])
    twb = np.stack([Rwg @ k[21:24] for k in kt]) + rng.normal(0, pos_noise, (len(kt), 3))
    vel = np.stack([Rwg @ k[24:27] for k in kt])
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    return dict(Rwb=f32(Rwb), twb=f32(twb), samples=w["samples"], Rwg_true=Rwg, vel_true=vel, bg_true=kt[0][27:30].copy(), ba_true=kt[0][30:33].copy())


def tbl7():
    """mLidarParam->mTbl = Tbc * Tcl of the synthetic rig as (qx, qy, qz, qw, tx, ty, tz) float32."""
    from scipy.spatial.transform import Rotation
    Rcl, tcl = _quat_R(TCL7[:4]), TCL7[4:].astype(np.float64)
    R, t = RBC @ Rcl, RBC @ tcl + TBC
    return np.concatenate([Rotation.from_matrix(R).as_quat(), t]).astype(np.float32)


def inertial_window_clouds(w, win_kf, n_points=2400, noise=0.02, seed=0):
    """Surface clouds (LiDAR frame, float32 [n, 3]) for the keyframes `win_kf` of an inertial_window(): samples of the ground
    z = -1.7 and of two walls y = -8 / y = +9 (world axes: x forward, z up) ahead of each TRUE keyframe pose."""
    rng = np.random.default_rng([SEED0, 0xC11D, seed])
    Rcl, tcl = _quat_R(TCL7[:4]), TCL7[4:].astype(np.float64)
    clouds = []
    for k in win_kf:
        Rcw, tcw, twb = w["kf33_true"][k][:9].reshape(3, 3), w["kf33_true"][k][9:12], w["kf33_true"][k][21:24]
        m = n_points // 3
        r = n_points - 2 * m
        g = np.stack([rng.uniform(2, 12, m) + twb[0], rng.uniform(-4, 4, m) + twb[1], np.full(m, -1.7)], 1)
        wl = np.stack([rng.uniform(2, 12, m) + twb[0], np.full(m, -8.0), rng.uniform(-1.7, 1.5, m)], 1)
        wr = np.stack([rng.uniform(2, 12, r) + twb[0], np.full(r, 9.0), rng.uniform(-1.7, 1.5, r)], 1)
        Xw = np.concatenate([g, wl, wr]) + rng.normal(0, noise, (n_points, 3))
        Xc = Xw @ Rcw.T + tcw
        clouds.append(((Xc - tcl) @ Rcl).astype(np.float32))
    return clouds
