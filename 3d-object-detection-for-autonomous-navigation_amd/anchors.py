"""Static anchor grid and its integer BEV footprint, built once on the host.

The reference rebuilds the anchors for every frame on the CPU
(load_data.py:3029 -> generate_anchors :1641-1685 -> create_anchors_3d_stride
:1598-1638) although they depend only on the config.  Here they are built once
at engine creation with the same float32 arithmetic (centre = f32(i) * f32(stride)
+ f32(offset), order (y, x, size, rot), row = [x,y,z,w,l,h,r]) and cached on
the device, together with the per-anchor integer corner cells the anchor-mask
kernel looks up in the occupancy integral image.  The cells follow
rbbox2d_to_near_bbox (load_data.py:535-547) and the float64 floor / clamp of
fused_get_anchors_area (load_data.py:567-578); only the 4 integral-image
gathers per anchor remain as per-frame work.
"""
import numpy as np


def build_anchors(derived):
    """Returns anchors [A,7] float32 in the reference's flattened order."""
    ag = derived.anchor_cfg
    f32 = np.float32
    sizes = np.asarray(ag["sizes"], dtype=f32).reshape(-1, 3)
    rots = np.asarray(ag["rotations"], dtype=f32)
    sx, sy, sz = (f32(v) for v in ag["strides"])
    ox, oy, oz = (f32(v) for v in ag["offsets"])
    H, W = derived.head_h, derived.head_w
    xs = np.arange(W, dtype=f32) * sx + ox
    ys = np.arange(H, dtype=f32) * sy + oy
    z = f32(0) * sz + oz
    S, R = sizes.shape[0], rots.shape[0]
    a = np.empty((H, W, S, R, 7), dtype=f32)
    a[..., 0] = xs.reshape(1, W, 1, 1)
    a[..., 1] = ys.reshape(H, 1, 1, 1)
    a[..., 2] = z
    a[..., 3:6] = sizes.reshape(1, 1, S, 1, 3)
    a[..., 6] = rots.reshape(1, 1, 1, R)
    return a.reshape(-1, 7)


def build_anchor_cells(anchors, derived):
    """Returns int32 [A,4] = (x0, y0, x1, y1) clamped integral-image cells."""
    f32 = np.float32
    r = anchors[:, 6]
    # limit_period(r, 0.5, pi) in float32 (load_data.py:805-806, :543)
    lp = r - np.floor(r / f32(np.pi) + f32(0.5)) * f32(np.pi)
    swap = np.abs(lp) > np.pi / 4
    dx = np.where(swap, anchors[:, 4], anchors[:, 3])
    dy = np.where(swap, anchors[:, 3], anchors[:, 4])
    x, y = anchors[:, 0], anchors[:, 1]
    lo_x, hi_x = x - dx / 2, x + dx / 2   # float32, as center_to_minmax_2d_0_5
    lo_y, hi_y = y - dy / 2, y + dy / 2
    vx, vy = derived.voxel_size[0], derived.voxel_size[1]
    x_min, y_min = derived.pc_range[0], derived.pc_range[1]

    def cell(v, lo, step):
        return np.floor((v.astype(np.float64) - lo) / step).astype(np.int32)

    x0 = np.maximum(cell(lo_x, x_min, vx), 0)
    y0 = np.maximum(cell(lo_y, y_min, vy), 0)
    x1 = np.minimum(cell(hi_x, x_min, vx), derived.nx - 1)
    y1 = np.minimum(cell(hi_y, y_min, vy), derived.ny - 1)
    return np.ascontiguousarray(np.stack([x0, y0, x1, y1], axis=1).astype(np.int32))
