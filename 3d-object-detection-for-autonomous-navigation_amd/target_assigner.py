"""Training-side target assignment (SURVEY section 8f, row f3 -- data half).

Host numpy, as in the reference (it runs in the tf.data thread, load_data.py:3085-3101):
  second_box_encode          load_data.py:125-203   (the inverse of the decode on the predict path)
  iou_jit                    load_data.py:206-235   axis-aligned IoU, eps = 0
  nearest_iou_similarity     load_data.py:238-256   rotated bev box -> nearest standing / lying box
  similarity_fn              load_data.py:258-262
  create_target_np / assign  load_data.py:267-293, :331-532
The matching is the FAIR-detectron rule: every ground-truth box takes the anchors that tie for its best
overlap, every anchor at or above `matched_threshold` takes its best box, anchors below
`unmatched_threshold` are background, the rest are ignored (-1).
"""
import numpy as np
import numpy.random as npr


def limit_period(val, offset=0.5, period=np.pi):
    """load_data.py:805-806."""
    return val - np.floor(val / period + offset) * period


def rbbox2d_to_near_bbox(rbboxes):
    """[N, 5] (x, y, xdim, ydim, rad) -> nearest standing / lying box [N, 4] (xmin, ymin, xmax, ymax)
    (load_data.py:535-551)."""
    rots = rbboxes[..., -1]
    cond = (np.abs(limit_period(rots, 0.5, np.pi)) > np.pi / 4)[..., np.newaxis]
    c = np.where(cond, rbboxes[:, [0, 1, 3, 2]], rbboxes[:, :4])
    return np.concatenate([c[:, :2] - c[:, 2:] / 2, c[:, :2] + c[:, 2:] / 2], axis=-1)


def second_box_encode(boxes, anchors, encode_angle_to_vector=False, smooth_dim=False):
    """boxes (ground truth) and anchors are [N, 7] x, y, z, w, l, h, r; float32 arithmetic like the
    reference's tf.float32 graph."""
    a = np.asarray(anchors, dtype=np.float32)
    g = np.asarray(boxes, dtype=np.float32)
    xa, ya, za, wa, la, ha, ra = (a[..., i:i + 1] for i in range(7))
    xg, yg, zg, wg, lg, hg, rg = (g[..., i:i + 1] for i in range(7))
    za = za + ha / np.float32(2)
    zg = zg + hg / np.float32(2)
    diagonal = np.sqrt(la ** 2 + wa ** 2)
    xt = (xg - xa) / diagonal
    yt = (yg - ya) / diagonal
    zt = (zg - za) / ha
    if smooth_dim:
        lt, wt, ht = lg / la - 1, wg / wa - 1, hg / ha - 1
    else:
        lt, wt, ht = np.log(lg / la), np.log(wg / wa), np.log(hg / ha)
    if encode_angle_to_vector:
        rtx = np.cos(rg) - np.cos(ra)
        rty = np.sin(rg) - np.sin(ra)
        return np.concatenate([xt, yt, zt, wt, lt, ht, rtx, rty], axis=-1)
    return np.concatenate([xt, yt, zt, wt, lt, ht, rg - ra], axis=-1)


def iou_jit(boxes, query_boxes, eps=0.0):
    """[N, 4] x [K, 4] (xmin, ymin, xmax, ymax) -> [N, K], in the dtype of `boxes`; zero where the
    boxes do not overlap (strictly positive width and height of the intersection required)."""
    boxes = np.asarray(boxes)
    q = np.asarray(query_boxes).astype(boxes.dtype, copy=False)
    eps = boxes.dtype.type(eps)
    iw = np.minimum(boxes[:, None, 2], q[None, :, 2]) - np.maximum(boxes[:, None, 0], q[None, :, 0]) + eps
    ih = np.minimum(boxes[:, None, 3], q[None, :, 3]) - np.maximum(boxes[:, None, 1], q[None, :, 1]) + eps
    area_b = (boxes[:, 2] - boxes[:, 0] + eps) * (boxes[:, 3] - boxes[:, 1] + eps)
    area_q = (q[:, 2] - q[:, 0] + eps) * (q[:, 3] - q[:, 1] + eps)
    inter = iw * ih
    ua = area_b[:, None] + area_q[None, :] - inter
    ok = (iw > 0) & (ih > 0)
    out = np.zeros(inter.shape, dtype=boxes.dtype)
    np.divide(inter, ua, out=out, where=ok)
    return out


def nearest_iou_similarity(boxes1, boxes2):
    return iou_jit(rbbox2d_to_near_bbox(boxes1), rbbox2d_to_near_bbox(boxes2), eps=0.0)


def similarity_fn(anchors, gt_boxes):
    return nearest_iou_similarity(anchors[:, [0, 1, 3, 4, 6]], gt_boxes[:, [0, 1, 3, 4, 6]])


def box_encoding_fn(boxes, anchors):
    return second_box_encode(boxes, anchors)


def unmap(data, count, inds, fill=0):
    if count == len(inds):
        return data
    ret = np.empty((count,) + data.shape[1:], dtype=data.dtype)
    ret.fill(fill)
    ret[inds] = data
    return ret


def _match(iou, gt_classes, hi_thr, lo_thr):
    """The matching rule on an [n_anchor, n_gt] similarity matrix.  Returns (labels, gt index per anchor (-1: none),
    best overlap per anchor, background anchor indices, forced anchors, their boxes)."""
    n = iou.shape[0]
    labels = np.full((n,), -1, dtype=np.int32)
    owner = np.full((n,), -1, dtype=np.int32)
    best_gt = iou.argmax(axis=1)                       # per anchor: its best box ...
    best_iou = iou[np.arange(n), best_gt]              # ... and how well it fits
    top_anchor = iou.argmax(axis=0)                    # per box: an anchor with its best overlap
    top_iou = iou[top_anchor, np.arange(iou.shape[1])]
    top_iou[top_iou == 0] = -1                         # a box that touches no anchor forces nothing
    forced = np.where(iou == top_iou)[0]               # every anchor tying for a box's best overlap
    forced_gt = best_gt[forced]
    labels[forced] = gt_classes[forced_gt]
    owner[forced] = forced_gt
    good = best_iou >= hi_thr
    labels[good] = gt_classes[best_gt[good]]
    owner[good] = best_gt[good]
    background = np.where(best_iou < lo_thr)[0]
    return labels, owner, best_gt, best_iou, background, forced, forced_gt


def create_target_np(all_anchors, gt_boxes, prune_anchor_fn, gt_classes, matched_threshold, unmatched_threshold,
                     positive_fraction, rpn_batch_size, norm_by_num_examples, box_code_size,
                     bbox_inside_weight=None):
    """load_data.py:331-532 (same arguments and returned keys)."""
    total = all_anchors.shape[0]
    keep = prune_anchor_fn(all_anchors) if prune_anchor_fn is not None else None
    anchors = all_anchors if keep is None else all_anchors[keep, :]
    if keep is not None:
        if not isinstance(matched_threshold, float):
            matched_threshold = matched_threshold[keep]
        if not isinstance(unmatched_threshold, float):
            unmatched_threshold = unmatched_threshold[keep]
    n = anchors.shape[0] if keep is None else len(keep)
    if gt_classes is None:
        gt_classes = np.ones([gt_boxes.shape[0]], dtype=np.int32)
    matched = len(gt_boxes) > 0 and anchors.shape[0] > 0
    if matched:
        labels, owner, best_gt, best_iou, background, forced, forced_gt = _match(
            similarity_fn(anchors, gt_boxes), gt_classes, matched_threshold, unmatched_threshold)
    else:
        labels = np.full((n,), -1, dtype=np.int32)
        owner = np.full((n,), -1, dtype=np.int32)
        background = np.arange(n)
    positives = np.where(labels > 0)[0]
    positive_overlap = best_iou[positives] if matched else None
    positive_gt = owner[positives]
    if positive_fraction is not None:
        # subsampling with numpy's global generator, in the reference's order of draws
        quota = int(positive_fraction * rpn_batch_size)
        if len(positives) > quota:
            labels[npr.choice(positives, size=(len(positives) - quota), replace=False)] = -1
            positives = np.where(labels > 0)[0]
        room = rpn_batch_size - np.sum(labels > 0)
        if len(background) > room:
            labels[background[npr.randint(len(background), size=room)]] = 0
    elif not matched:
        labels[:] = 0
    else:
        labels[background] = 0
        labels[forced] = gt_classes[forced_gt]          # forced matches win over background
    targets = np.zeros((n, box_code_size), dtype=all_anchors.dtype)
    if matched:
        targets[positives, :] = box_encoding_fn(gt_boxes[best_gt[positives], :], anchors[positives, :])
    weights = np.zeros((n,), dtype=all_anchors.dtype)
    weights[labels > 0] = 1.0 / np.maximum(1.0, np.sum(labels >= 0)) if norm_by_num_examples else 1.0
    if keep is not None:
        labels = unmap(labels, total, keep, fill=-1)
        targets = unmap(targets, total, keep, fill=0)
        weights = unmap(weights, total, keep, fill=0)
    return {
        "labels": labels, "bbox_targets": targets, "bbox_outside_weights": weights,
        "assigned_anchors_overlap": positive_overlap, "positive_gt_id": positive_gt,
        "assigned_anchors_inds": keep[positives] if keep is not None else positives,
    }


def assign(anchors, gt_boxes, anchors_mask, gt_classes, matched_thresholds, unmatched_thresholds,
           config_target_assigner):
    prune_anchor_fn = (lambda _: np.where(anchors_mask)[0]) if anchors_mask is not None else None
    frac = config_target_assigner["sample_positive_fraction"]
    if frac == "None":
        frac = None
    return create_target_np(anchors, gt_boxes, prune_anchor_fn=prune_anchor_fn, gt_classes=gt_classes,
                            matched_threshold=matched_thresholds, unmatched_threshold=unmatched_thresholds,
                            positive_fraction=frac, rpn_batch_size=config_target_assigner["rpn_batch_size"],
                            norm_by_num_examples=False, box_code_size=7)
