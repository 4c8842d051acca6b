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

    def render(self, cam_x=0.0, width=WIDTH, height=HEIGHT, noise_seed=0):
        """8-bit image and depth (z) map seen from a camera translated by cam_x along +x."""
        uu, vv = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
        dx, dy = (uu - np.float32(CX)) / np.float32(FX), (vv - np.float32(CY)) / np.float32(FY)  # ray (dx, dy, 1)
        depth = np.full((height, width), 1e6, np.float32)
        img = (200.0 - 40.0 * (vv / np.float32(height))).astype(np.float32)  # sky with a faint gradient
        r0 = min(max(int(np.ceil(CY + FY * CAM_HEIGHT / 120.0)), 0), height)  # first row whose ground hit is < 120 m
        if r0 < height:
            zg = (np.float32(CAM_HEIGHT) / dy[r0:]).astype(np.float32)
            X = (np.float32(cam_x) + dx[r0:] * zg).astype(np.float32)
            img[r0:] = _texture(X, zg, int(self.salts[-1]), 2.0)
            depth[r0:] = zg
        order = np.argsort(-self.boxes[:, 4], kind="stable")  # far to near: nearer boxes overwrite
        for k in order:
            x0, x1, y0, y1, zb = self.boxes[k]
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
