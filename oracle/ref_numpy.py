"""numpy restatement of the reference's host-side hot-path functions.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function cites the
reference file:line it follows (paths relative to /root/reference).  The
functions marked PINNED are checked bit-for-bit against outputs of the
reference's own code in tests/test_oracle_golden.py.

Old-numpy idiom: the reference indexes with list-wrapped index arrays
(`x[[idx]]`, model/voxelnet.py:1127-1137, 1209-1214;
libraries/eval_helper_functions.py:477-478, 490).  Under the reference's pinned
numpy 1.19 that means `x[idx]`; it is restated as `x[idx]` here.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# a1  voxel generator                                                 PINNED
# --------------------------------------------------------------------------
def grid_size(coors_range, voxel_size):
    """load_data.py:612-615, 730-731, 2599-2601: round((max-min)/voxel)."""
    coors_range = np.asarray(coors_range, dtype=np.float64)
    voxel_size = np.asarray(voxel_size, dtype=np.float64)
    g = (coors_range[3:] - coors_range[:3]) / voxel_size
    return np.round(g).astype(np.int64)  # (nx, ny, nz); round-half-even (1.5 -> 2)


def points_to_voxel(points, voxel_size, coors_range, max_points, reverse_index, max_voxels):
    """load_data.py:695-771 + :593-641 (reverse kernel) / :643-692.

    Sequential scan of the points in input order; the cell of a point is
    floor((p - min) / voxel) evaluated in float64 (float32 point promoted
    against the float64 range / voxel arrays, load_data.py:622); a pillar id is
    the order of first appearance of its cell; the slot in the pillar is the
    arrival order, truncated at max_points; when pillar number max_voxels+1
    would be opened the scan stops and every later point is dropped (:632-633).
    Returns (voxels[P,T,F] f32, coors[P,3] i32 (zyx if reverse_index), num[P] i32).
    """
    points = np.asarray(points)
    if not isinstance(voxel_size, np.ndarray):  # load_data.py:726-729
        voxel_size = np.array(voxel_size, dtype=points.dtype)
    if not isinstance(coors_range, np.ndarray):
        coors_range = np.array(coors_range, dtype=points.dtype)
    g = np.round((coors_range[3:] - coors_range[:3]) / voxel_size).astype(np.int32)
    shape = tuple(int(v) for v in g)
    if reverse_index:
        shape = shape[::-1]
    n_feat = points.shape[-1]
    num = np.zeros((max_voxels,), dtype=np.int32)
    cell_to_pillar = -np.ones(shape, dtype=np.int32)
    voxels = np.zeros((max_voxels, max_points, n_feat), dtype=points.dtype)
    coors = np.zeros((max_voxels, 3), dtype=np.int32)
    n_pillars = 0
    cell = [0, 0, 0]
    for i in range(points.shape[0]):
        ok = True
        for j in range(3):
            c = np.floor((points[i, j] - coors_range[j]) / voxel_size[j])
            if c < 0 or c >= g[j]:
                ok = False
                break
            cell[2 - j if reverse_index else j] = int(c)
        if not ok:
            continue
        pid = cell_to_pillar[cell[0], cell[1], cell[2]]
        if pid == -1:
            pid = n_pillars
            if n_pillars >= max_voxels:
                break
            n_pillars += 1
            cell_to_pillar[cell[0], cell[1], cell[2]] = pid
            coors[pid] = cell
        k = num[pid]
        if k < max_points:
            voxels[pid, k] = points[i]
            num[pid] += 1
    return voxels[:n_pillars], coors[:n_pillars], num[:n_pillars]


def merge_batch(frames):
    """load_data.py:2164-2224 (merge_second_batch): concat voxels / num_points,
    prepend the batch index to each frame's coors, stack the rest.
    frames: list of dicts with voxels, num_points, coordinates."""
    voxels = np.concatenate([f["voxels"] for f in frames], axis=0)
    num_points = np.concatenate([f["num_points"] for f in frames], axis=0)
    coors = np.concatenate(
        [np.pad(f["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=b)
         for b, f in enumerate(frames)], axis=0)
    return voxels, num_points, coors


# --------------------------------------------------------------------------
# a3  anchors                                                         PINNED
# --------------------------------------------------------------------------
def create_anchors_3d_stride(feature_size, sizes, anchor_strides, anchor_offsets, rotations,
                             dtype=np.float32):
    """load_data.py:1598-1638.  feature_size = [D,H,W] (zyx).  Returns
    [D,H,W,num_sizes,num_rots,7] with rows [x,y,z,w,l,h,r]; centres are
    arange(n, f32) * stride + offset evaluated in f32."""
    xs, ys, zs = anchor_strides
    xo, yo, zo = anchor_offsets
    zc = np.arange(feature_size[0], dtype=dtype) * zs + zo
    yc = np.arange(feature_size[1], dtype=dtype) * ys + yo
    xc = np.arange(feature_size[2], dtype=dtype) * xs + xo
    sizes = np.reshape(np.array(sizes, dtype=dtype), [-1, 3])
    rots = np.array(rotations, dtype=dtype)
    D, H, W, S, R = len(zc), len(yc), len(xc), sizes.shape[0], len(rots)
    out = np.zeros((D, H, W, S, R, 7), dtype=dtype)
    out[..., 0] = xc[None, None, :, None, None]
    out[..., 1] = yc[None, :, None, None, None]
    out[..., 2] = zc[:, None, None, None, None]
    out[..., 3:6] = sizes[None, None, None, :, None, :]
    out[..., 6] = rots[None, None, None, None, :]
    return out


def generate_anchors(feature_map_size, cfg):
    """load_data.py:1641-1685.  Returns anchors [D,H,W,S*R,7] f32."""
    a = create_anchors_3d_stride(feature_map_size, cfg["sizes"], cfg["strides"],
                                 cfg["offsets"], cfg["rotations"])
    return a.reshape([*a.shape[:3], -1, 7])


def feature_map_size(grid, layer_strides, upsample_strides):
    """load_data.py:3019-3023: [1, ny//f, nx//f], f = layer_strides[0]//upsample_strides[0]."""
    f = int(layer_strides[0]) // int(upsample_strides[0])
    return [1, int(grid[1]) // f, int(grid[0]) // f]


# --------------------------------------------------------------------------
# a4  anchor mask                                                     PINNED
# --------------------------------------------------------------------------
def limit_period(val, offset=0.5, period=np.pi):
    """load_data.py:805-806."""
    return val - np.floor(val / period + offset) * period


def rbbox2d_to_near_bbox(rbboxes):
    """load_data.py:535-556.  [N,5](x,y,xdim,ydim,rad) -> [N,4] min/max AABB,
    dims swapped when |limit_period(r)| > pi/4."""
    rots = rbboxes[..., -1]
    swap = (np.abs(limit_period(rots, 0.5, np.pi)) > np.pi / 4)[..., None]
    cd = np.where(swap, rbboxes[:, [0, 1, 3, 2]], rbboxes[:, :4])
    c, d = cd[:, :2], cd[:, 2:]
    return np.concatenate([c - d / 2, c + d / 2], axis=-1)


def occupancy_map(coors, shape_yx):
    """load_data.py:586-591 (sparse_sum_for_anchors_mask): += 1 at (y, x) per pillar."""
    m = np.zeros(shape_yx, dtype=np.float32)
    np.add.at(m, (coors[:, 1], coors[:, 2]), F32(1))
    return m


def anchors_area(dense_map, anchors_bv, stride, offset, grid):
    """load_data.py:558-584 (fused_get_anchors_area): integral-image lookups at
    floor((box - range_min)/voxel) (float64 math, stored to int32), clamped."""
    stride = np.asarray(stride, dtype=np.float64)
    offset = np.asarray(offset, dtype=np.float64)
    bv = anchors_bv.astype(np.float64)
    x0 = np.floor((bv[:, 0] - offset[0]) / stride[0]).astype(np.int32)
    y0 = np.floor((bv[:, 1] - offset[1]) / stride[1]).astype(np.int32)
    x1 = np.floor((bv[:, 2] - offset[0]) / stride[0]).astype(np.int32)
    y1 = np.floor((bv[:, 3] - offset[1]) / stride[1]).astype(np.int32)
    x0 = np.maximum(x0, 0)
    y0 = np.maximum(y0, 0)
    x1 = np.minimum(x1, int(grid[0]) - 1)
    y1 = np.minimum(y1, int(grid[1]) - 1)
    return dense_map[y1, x1] - dense_map[y1, x0] - dense_map[y0, x1] + dense_map[y0, x0]


def anchor_cells(anchors_bv, stride, offset, grid):
    """Integer corner cells (x0,y0,x1,y1) exactly as anchors_area computes them."""
    stride = np.asarray(stride, dtype=np.float64)
    offset = np.asarray(offset, dtype=np.float64)
    bv = anchors_bv.astype(np.float64)
    x0 = np.maximum(np.floor((bv[:, 0] - offset[0]) / stride[0]).astype(np.int32), 0)
    y0 = np.maximum(np.floor((bv[:, 1] - offset[1]) / stride[1]).astype(np.int32), 0)
    x1 = np.minimum(np.floor((bv[:, 2] - offset[0]) / stride[0]).astype(np.int32), int(grid[0]) - 1)
    y1 = np.minimum(np.floor((bv[:, 3] - offset[1]) / stride[1]).astype(np.int32), int(grid[1]) - 1)
    return np.stack([x0, y0, x1, y1], axis=1)


def anchors_mask(coors, anchors, voxel_size, coors_range, threshold):
    """load_data.py:3043-3072.  coors [P,3] zyx of ONE frame; anchors [A,7]."""
    grid = grid_size(coors_range, voxel_size)
    bv = rbbox2d_to_near_bbox(anchors[:, [0, 1, 3, 4, 6]])
    dense = occupancy_map(coors, tuple(int(v) for v in grid[::-1][1:]))
    dense = dense.cumsum(0).cumsum(1)
    area = anchors_area(dense, bv, np.asarray(voxel_size, dtype=np.float64),
                        np.asarray(coors_range, dtype=np.float64), grid)
    return area > threshold


# --------------------------------------------------------------------------
# a9  box decode                                                      PINNED
# --------------------------------------------------------------------------
def second_box_decode(enc, anchors):
    """libraries/eval_helper_functions.py:388-461 (default flags)."""
    xa, ya, za, wa, la, ha, ra = np.split(anchors, 7, axis=-1)
    xt, yt, zt, wt, lt, ht, rt = np.split(enc, 7, axis=-1)
    za = za + ha / 2
    diag = np.sqrt(la ** 2 + wa ** 2)
    xg = xt * diag + xa
    yg = yt * diag + ya
    zg = zt * ha + za
    lg = np.exp(lt) * la
    wg = np.exp(wt) * wa
    hg = np.exp(ht) * ha
    rg = rt + ra
    zg = zg - hg / 2
    return np.concatenate([xg, yg, zg, wg, lg, hg, rg], axis=-1)


# --------------------------------------------------------------------------
# a8  corners / stand-up boxes                                        PINNED
# --------------------------------------------------------------------------
def center_to_corner_box2d(centers, dims, angles):
    """load_data.py:1525-1593: corner order (-,-),(-,+),(+,+),(+,-) * dims/2,
    rotated by [[c,-s],[s,c]] (einsum 'aij,jka->aik'), plus centre."""
    unit = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=dims.dtype) - np.array(0.5, dtype=dims.dtype)
    corners = dims.reshape([-1, 1, 2]) * unit.reshape([1, 4, 2])
    s, c = np.sin(angles), np.cos(angles)
    rot = np.stack([[c, -s], [s, c]])
    corners = np.einsum("aij,jka->aik", corners, rot)
    corners += centers.reshape([-1, 1, 2])
    return corners


def corner_to_standup(corners):
    """load_data.py:1330-1340: [N,4,2] -> [N,4] (xmin,ymin,xmax,ymax)."""
    return np.concatenate([corners.min(axis=1), corners.max(axis=1)], axis=1).astype(corners.dtype)


# --------------------------------------------------------------------------
# a10/a11  NMS      nms(), nms_gpu, nms_kernel, host sweep PINNED (ref_cuda_kernels.npz: the reference kernels on an emulator)
# --------------------------------------------------------------------------
def nms_iou(a, b):
    """libraries/eval_helper_functions.py:553-564 (iou_device).  Inputs f32; the
    differences are f32, the `+ 1` promotes to f64 (numba types f32+int64 as
    f64), everything after is f64."""
    left = max(a[0], b[0])
    right = min(a[2], b[2])
    top = max(a[1], b[1])
    bottom = min(a[3], b[3])
    w = max(np.float64(F32(right - left)) + 1.0, 0.0)
    h = max(np.float64(F32(bottom - top)) + 1.0, 0.0)
    inter = w * h
    sa = (np.float64(F32(a[2] - a[0])) + 1.0) * (np.float64(F32(a[3] - a[1])) + 1.0)
    sb = (np.float64(F32(b[2] - b[0])) + 1.0) * (np.float64(F32(b[3] - b[1])) + 1.0)
    return inter / (sa + sb - inter)


def nms_mask(boxes, thresh):
    """libraries/eval_helper_functions.py:567-598 (nms_kernel): row i, column
    block cb: bit j set iff j > i (same block: start=tx+1) and IoU > thresh.
    boxes [n,5] f32 sorted by score descending.  Returns uint64[n*col_blocks]."""
    n = boxes.shape[0]
    cb = -(-n // 64)
    mask = np.zeros((n * cb,), dtype=np.uint64)
    thr = np.float64(F32(thresh))
    for i in range(n):
        for blk in range(cb):
            t = 0
            lo = blk * 64
            hi = min(n, lo + 64)
            start = (i % 64) + 1 if blk == i // 64 else 0
            for k in range(start, hi - lo):
                if nms_iou(boxes[i, :4], boxes[lo + k, :4]) > thr:
                    t |= 1 << k
            mask[i * cb + blk] = np.uint64(t)
    return mask


def nms_postprocess(mask, n):
    """libraries/eval_helper_functions.py:529-546: greedy sweep over the bitmask."""
    cb = -(-n // 64)
    remv = [0] * cb
    keep = []
    for i in range(n):
        blk, bit = divmod(i, 64)
        if not (remv[blk] >> bit) & 1:
            keep.append(i)
            for j in range(blk, cb):
                remv[j] |= int(mask[i * cb + j])
    return keep


def nms_gpu(dets, thresh):
    """libraries/eval_helper_functions.py:494-527."""
    order = dets[:, 4].argsort()[::-1].astype(np.int32)
    boxes = dets[order, :]
    keep = nms_postprocess(nms_mask(boxes, thresh), boxes.shape[0])
    return list(order[keep])


def nms(bboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """libraries/eval_helper_functions.py:463-492."""
    if pre_max_size is not None:
        pre = min(scores.shape[0], pre_max_size)
        indices = np.argpartition(scores, -pre)[-pre:]
        scores = scores[indices]
        bboxes = bboxes[indices]
    dets = np.concatenate([bboxes, scores[:, None]], axis=1)
    if len(dets) == 0:
        keep = np.array([], dtype=np.int64)
    else:
        keep = np.array(nms_gpu(dets, iou_threshold), dtype=np.int64)[:post_max_size]
    if keep.shape[0] == 0:
        return None
    return indices[keep] if pre_max_size is not None else keep


# --------------------------------------------------------------------------
# a12  lidar -> camera                                                PINNED
# --------------------------------------------------------------------------
def lidar_to_camera(points, r_rect, velo2cam):
    """libraries/eval_helper_functions.py:728-733 (np.ones is f64 => f64 out)."""
    shape = list(points.shape[:-1])
    if points.shape[-1] == 3:
        points = np.concatenate([points, np.ones(shape + [1])], axis=-1)
    return (points @ (r_rect @ velo2cam).T)[..., :3]


def box_lidar_to_camera(data, r_rect, velo2cam):
    """libraries/eval_helper_functions.py:735-740: xyz transformed, (w,l,h)->(l,h,w)."""
    xyz = lidar_to_camera(data[:, 0:3], r_rect, velo2cam)
    w, l, h, r = data[:, 3:4], data[:, 4:5], data[:, 5:6], data[:, 6:7]
    return np.concatenate([xyz, l, h, w, r], axis=1)


# --------------------------------------------------------------------------
# a8 + a12  VoxelNet.predict                       restated (PARITY UNPINNED)
# --------------------------------------------------------------------------
def sigmoid_array(x):
    """model/voxelnet.py:722-723."""
    return 1 / (1 + np.exp(-x))


def predict(example, preds, cfg):
    """model/voxelnet.py:1060-1390 for encode_background_as_zeros and no multi-class NMS
    (the shipped config; those other branches are TF stubs in the reference).
    cfg["num_class"] > 1 follows the stub at :1183-1185 (top_scores = reduce_max,
    top_labels = argmax over the class scores) restated in numpy -- beyond what the
    reference can run.  cfg["use_direction_classifier"] False: no dir head, no flip
    (:1093-1096, :1297).

    example: 10-tuple (voxels, num_points, coors, rect, Trv2c, P2, anchors,
    anchors_mask, image_idx, image_shape) of numpy arrays;  preds: dict of
    numpy arrays box_preds/cls_preds/dir_cls_preds [B,H,W,*].
    cfg: dict with nms_score_threshold, nms_pre_max_size, nms_post_max_size,
    nms_iou_threshold.
    """
    anchors_b = example[6]
    B = anchors_b.shape[0]
    rect_b, trv_b, mask_b, idx_b = example[3], example[4], example[7], example[8]
    box_b = np.reshape(preds["box_preds"], (B, -1, 7))
    ncls = int(cfg.get("num_class", 1))
    use_dir = bool(cfg.get("use_direction_classifier", True))
    cls_b = np.reshape(preds["cls_preds"], (B, -1, ncls))
    dir_b = np.reshape(preds["dir_cls_preds"], (B, -1, 2)) if use_dir else [None] * B
    out = []
    for b in range(B):
        sel = np.where(mask_b[b] == 1)[0]
        box, cls, anc = box_b[b][sel], cls_b[b][sel], anchors_b[b][sel]
        if use_dir:
            dir_labels = np.argmax(dir_b[b][sel], axis=-1)
        else:
            dir_labels = np.zeros(box.shape[0], dtype=np.int64)
        total = sigmoid_array(cls)
        if ncls == 1:
            scores = np.squeeze(total, axis=-1)
            labels = np.zeros(scores.shape[0], dtype=int)
        else:
            scores = total.max(axis=-1)
            labels = np.argmax(total, axis=-1)
        thr = cfg["nms_score_threshold"]
        if thr > 0.0:
            k = scores >= thr
            scores, box, anc, dir_labels, labels = scores[k], box[k], anc[k], dir_labels[k], labels[k]
        n_top = np.minimum(len(scores), 100)
        top = np.argpartition(scores, -n_top)[-n_top:] if len(scores) else np.zeros((0,), dtype=np.int64)
        scores, box, anc, dir_labels, labels = scores[top], box[top], anc[top], dir_labels[top], labels[top]
        selected = None
        if scores.shape[0] != 0:
            box = second_box_decode(box, anc)
            bev = box[..., [0, 1, 3, 4, 6]]
            standup = corner_to_standup(center_to_corner_box2d(bev[:, :2], bev[:, 2:4], bev[:, 4]))
            selected = nms(standup, scores, pre_max_size=cfg["nms_pre_max_size"],
                           post_max_size=cfg["nms_post_max_size"],
                           iou_threshold=cfg["nms_iou_threshold"])
        if selected is not None:
            fbox = box[selected]
            fdir = dir_labels[selected]
            if use_dir:
                opp = ((fbox[..., -1] > 0) ^ fdir) > 0  # model/voxelnet.py:1305 precedence
                fbox[..., -1] += np.where(opp, np.pi, 0.0)
            cam = box_lidar_to_camera(fbox, rect_b[b], trv_b[b])
            out.append({
                "bbox": np.tile(np.array([[400., 200., 500., 400.]]), (fbox.shape[0], 1)),
                "box3d_camera": cam,
                "box3d_lidar": fbox,
                "scores": scores[selected],
                "label_preds": labels[selected],
                "batch_idx": idx_b[b],
            })
        else:
            out.append({"bbox": None, "box3d_camera": None, "box3d_lidar": None,
                        "scores": None, "label_preds": None, "batch_idx": idx_b[b]})
    return out
