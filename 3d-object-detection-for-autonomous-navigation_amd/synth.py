"""Seeded synthetic point clouds of the shapes BASELINE.json names (no datasets here).

d435i cloud: N float32 xyz points; depth x ~ 0.3 + 6.1*Beta(2,3),
y ~ U(-0.62 x, 0.62 x) (about 64 deg HFOV), z ~ N(0, 0.45) clipped to (-1, 1.2)
plus a ground band, ~2 % of points outside the voxel range, 1-3 pedestrian
blobs (0.5 x 0.5 x 1.7 m, 400-1500 points each), and a share of points lifted
above z = 1.0 so that the second z-cell of the shipped config is populated
(SURVEY Appendix B).  KITTI-shaped cloud: N xyzi points, 1/r^2 density.
"""
import numpy as np


def d435i_cloud(frame, n_points=16384, num_features=3, seed=1234):
    rng = np.random.default_rng(seed + int(frame))
    n_blobs = int(rng.integers(1, 4))
    blob_sizes = [int(rng.integers(400, 1501)) for _ in range(n_blobs)]
    n_out = int(0.02 * n_points)
    n_ground = int(0.15 * n_points)
    n_bg = n_points - sum(blob_sizes) - n_out - n_ground
    if n_bg < 0:
        blob_sizes = [max(1, n_points // 16)] * n_blobs
        n_out, n_ground = n_points // 50, n_points // 8
        n_bg = n_points - sum(blob_sizes) - n_out - n_ground
    parts = []
    x = 0.3 + 6.1 * rng.beta(2.0, 3.0, n_bg)
    y = np.clip(rng.uniform(-0.62, 0.62, n_bg) * x, -2.559, 2.559)
    z = np.clip(rng.normal(0.0, 0.45, n_bg), -1.0, 1.2)
    parts.append(np.stack([x, y, z], axis=1))
    xg = 0.3 + 6.1 * rng.beta(2.0, 2.0, n_ground)
    yg = np.clip(rng.uniform(-0.62, 0.62, n_ground) * xg, -2.559, 2.559)
    zg = -0.95 + 0.03 * rng.standard_normal(n_ground)
    parts.append(np.stack([xg, yg, zg], axis=1))
    for sz in blob_sizes:
        cx = rng.uniform(1.0, 5.5)
        cy = rng.uniform(-0.5, 0.5) * cx
        px = cx + rng.uniform(-0.25, 0.25, sz)
        py = cy + rng.uniform(-0.25, 0.25, sz)
        pz = rng.uniform(-0.95, 0.75, sz)
        # a d435i cloud is lifted by +1 m before voxelisation (load_data.py:2443):
        # let some blobs reach above z = 1.0 into the second z-cell
        if rng.random() < 0.5:
            pz = pz + 0.6
        parts.append(np.stack([px, py, pz], axis=1))
    xo = rng.uniform(-1.0, 8.0, n_out)
    yo = rng.uniform(-4.0, 4.0, n_out)
    zo = rng.uniform(-4.0, 4.0, n_out)
    parts.append(np.stack([xo, yo, zo], axis=1))
    pts = np.concatenate(parts, axis=0)
    pts = pts[rng.permutation(pts.shape[0])]
    if num_features > 3:
        extra = rng.uniform(0.0, 1.0, (pts.shape[0], num_features - 3))
        pts = np.concatenate([pts, extra], axis=1)
    return np.ascontiguousarray(pts[:n_points].astype(np.float32))


def kitti_cloud(frame, n_points=20000, num_features=4, seed=4321):
    rng = np.random.default_rng(seed + int(frame))
    # 1/r^2 density out to ~70 m within the forward half-plane
    r = 2.0 / (rng.uniform(2.0 / 75.0, 1.0, n_points))
    th = rng.uniform(-0.85, 0.85, n_points)
    x = r * np.cos(th)
    y = r * np.sin(th)
    z = np.clip(-1.6 + 0.25 * rng.standard_normal(n_points) + rng.uniform(0, 1.8, n_points) *
                (rng.random(n_points) < 0.3), -2.99, 0.99)
    cols = [x, y, z]
    for _ in range(num_features - 3):
        cols.append(rng.uniform(0.0, 1.0, n_points))
    return np.ascontiguousarray(np.stack(cols, axis=1).astype(np.float32))


def default_calib():
    """rect = I, Trv2c as in the reference's production path (train.py:681-682)."""
    rect = np.eye(4, dtype=np.float32)
    trv2c = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    p2 = np.eye(4, dtype=np.float32)
    return rect, trv2c, p2
