"""KITTI-style AP evaluator behind `train.py evaluate`.

Mirrors the call surface of the reference's second/utils/eval.py:
  get_official_eval_result  :836-917   (bev / 3d / aos [/ bbox] AP at the 6 custom IoU tiers)
  get_coco_eval_result      :920-997   do_coco_style_eval :764-781
  do_eval_v2                :693-755   eval_class_v3 :552-660
  calculate_iou_partly      :329-418   clean_data :40-93   compute_statistics_jit :167-287
  get_thresholds            :18-37     image_box_overlap :97-124
The rotated-rectangle overlaps (bev_box_overlap :127-129, d3_box_overlap :159-163; numba-CUDA in
the reference) run on the GPU through the C-ABI (`pp_rotate_iou_eval`, `pp_d3_box_overlap`,
csrc/rotate_iou.hip) -- there is no CPU fallback for them; everything else is host bookkeeping
on a few boxes per frame.

Differences from the reference, none of which change a result it can produce:
  * the greedy gt<->detection matching is evaluated per ground-truth box with array operations
    instead of a scalar double loop (same winner, same tie-breaks: first maximum);
  * statistics are accumulated frame by frame, not over concatenated "parts" (the parts only exist
    in the reference to batch the IoU kernel; results are per frame either way), so fewer than
    `num_parts` frames work here (the reference raises on an empty part).
"""
import io

import numpy as np

from . import _lib

CLASS_NAMES = ['car', 'pedestrian', 'cyclist', 'van', 'person_sitting', 'car', 'tractor', 'trailer']
MIN_HEIGHT = (40, 25, 25)
MAX_OCCLUSION = (0, 1, 2)
MAX_TRUNCATION = (0.15, 0.3, 0.5)
N_SAMPLE_PTS = 41
_NO_DET = -10000000

CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting', 5: 'car', 6: 'tractor',
                 7: 'trailer'}


# ------------------------------------------------------------------------------------------
# overlaps
# ------------------------------------------------------------------------------------------
def _check(status, who):
    if status != 0:
        raise RuntimeError(f"{who}: {_lib.lib().pp_last_error(None).decode()}")


def rotate_iou_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """[N,5] x [K,5] (x, y, x size, y size, angle) -> [N,K]; rotate_iou_gpu_eval, nms_gpu.py:618-653."""
    boxes = np.asarray(boxes)
    dtype = boxes.dtype
    b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 5)
    q = np.ascontiguousarray(query_boxes, dtype=np.float32).reshape(-1, 5)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    if b.shape[0] and q.shape[0]:
        _check(_lib.lib().pp_rotate_iou_eval(int(device_id), b.ctypes.data, b.shape[0], q.ctypes.data, q.shape[0],
                                              int(criterion), out.ctypes.data), "rotate_iou_eval")
    return out.astype(dtype)


def bev_box_overlap(boxes, qboxes, criterion=-1, device_id=0):
    return rotate_iou_eval(boxes, qboxes, criterion, device_id)


def d3_box_overlap(boxes, qboxes, criterion=-1, device_id=0):
    """Camera-frame [N,7] x [K,7] (x, y, z, l, h, w, ry) -> [N,K] 3D overlap."""
    b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 7)
    q = np.ascontiguousarray(qboxes, dtype=np.float64).reshape(-1, 7)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float64)
    if b.shape[0] and q.shape[0]:
        _check(_lib.lib().pp_d3_box_overlap(int(device_id), b.ctypes.data, b.shape[0], q.ctypes.data, q.shape[0],
                                             int(criterion), out.ctypes.data), "d3_box_overlap")
    return out


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """Axis-aligned 2D boxes [N,4] x [K,4] (x1, y1, x2, y2) -> [N,K]."""
    boxes = np.asarray(boxes)
    query_boxes = np.asarray(query_boxes)
    N, K = boxes.shape[0], query_boxes.shape[0]
    out = np.zeros((N, K), dtype=boxes.dtype)
    if N == 0 or K == 0:
        return out
    b, q = boxes[:, None, :], query_boxes[None, :, :]
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0])
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1])
    barea = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    qarea = (q[..., 2] - q[..., 0]) * (q[..., 3] - q[..., 1])
    inter = iw * ih
    if criterion == -1:
        ua = barea + qarea - inter
    elif criterion == 0:
        ua = np.broadcast_to(barea, inter.shape)
    elif criterion == 1:
        ua = np.broadcast_to(qarea, inter.shape)
    else:
        ua = np.ones_like(inter)
    hit = (iw > 0) & (ih > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        out[hit] = (inter / ua)[hit]
    return out


def _bev_rects(annos):
    loc = np.concatenate([a["location"][:, [0, 2]] for a in annos], 0)
    dims = np.concatenate([a["dimensions"][:, [0, 2]] for a in annos], 0)
    rots = np.concatenate([a["rotation_y"] for a in annos], 0)
    return np.concatenate([loc, dims, rots[..., np.newaxis]], axis=1)


def _cam_boxes(annos):
    loc = np.concatenate([a["location"] for a in annos], 0)
    dims = np.concatenate([a["dimensions"] for a in annos], 0)
    rots = np.concatenate([a["rotation_y"] for a in annos], 0)
    return np.concatenate([loc, dims, rots[..., np.newaxis]], axis=1)


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50, overlap_fns=None):
    """Per-frame overlap matrices [len(gt_annos[i]), len(dt_annos[i])]; metric 0 bbox, 1 bev, 2 3d.
    Frames are batched `frames // num_parts` at a time into one kernel launch (cross-frame pairs are
    computed and discarded, as in the reference).  `overlap_fns` = {1: fn, 2: fn} overrides the
    GPU-backed overlap functions (used by the CPU tests to inject the oracle)."""
    assert len(gt_annos) == len(dt_annos)
    fns = {0: image_box_overlap, 1: bev_box_overlap, 2: d3_box_overlap}
    if overlap_fns:
        fns.update(overlap_fns)
    if metric not in fns:
        raise ValueError("unknown metric")
    n = len(gt_annos)
    total_gt = np.array([len(a["name"]) for a in gt_annos], dtype=np.int64)
    total_dt = np.array([len(a["name"]) for a in dt_annos], dtype=np.int64)
    step = max(1, n // max(1, num_parts))
    overlaps, parted = [], []
    for s in range(0, n, step):
        g, d = gt_annos[s:s + step], dt_annos[s:s + step]
        if metric == 0:
            part = fns[0](np.concatenate([a["bbox"] for a in g], 0), np.concatenate([a["bbox"] for a in d], 0))
        elif metric == 1:
            part = fns[1](_bev_rects(g), _bev_rects(d)).astype(np.float64)
        else:
            part = fns[2](_cam_boxes(g), _cam_boxes(d)).astype(np.float64)
        parted.append(part)
        gi = di = 0
        for i in range(s, min(s + step, n)):
            overlaps.append(part[gi:gi + total_gt[i], di:di + total_dt[i]])
            gi += total_gt[i]
            di += total_dt[i]
    return overlaps, parted, total_gt, total_dt


# ------------------------------------------------------------------------------------------
# per-frame bookkeeping
# ------------------------------------------------------------------------------------------
def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """Which gt / dt boxes count (0), are neutral (1) or belong to another class (-1) at this difficulty."""
    cls = CLASS_NAMES[current_class].lower()
    ignored_gt, ignored_dt, dc_bboxes = [], [], []
    num_valid_gt = 0
    for i in range(len(gt_anno["name"])):
        name = gt_anno["name"][i].lower()
        bbox = gt_anno["bbox"][i]
        if name == cls:
            valid = 1
        elif (cls == "pedestrian" and name == "person_sitting") or (cls == "car" and name == "van"):
            valid = 0
        else:
            valid = -1
        hard = (gt_anno["occluded"][i] > MAX_OCCLUSION[difficulty] or gt_anno["truncated"][i] > MAX_TRUNCATION[difficulty]
                or (bbox[3] - bbox[1]) <= MIN_HEIGHT[difficulty])
        if valid == 1 and not hard:
            ignored_gt.append(0)
            num_valid_gt += 1
        elif valid == 0 or (hard and valid == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
        if gt_anno["name"][i] == "DontCare":
            dc_bboxes.append(bbox)
    for i in range(len(dt_anno["name"])):
        height = abs(dt_anno["bbox"][i, 3] - dt_anno["bbox"][i, 1])
        if height < MIN_HEIGHT[difficulty]:
            ignored_dt.append(1)
        elif dt_anno["name"][i].lower() == cls:
            ignored_dt.append(0)
        else:
            ignored_dt.append(-1)
    return num_valid_gt, ignored_gt, ignored_dt, dc_bboxes


def compute_statistics(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap,
                       thresh=0, compute_fp=False, compute_aos=False):
    """Greedy matching of one frame.  overlaps [D,G]; gt_datas [G,5] (bbox, alpha); dt_datas [D,6]
    (bbox, alpha, score).  Returns tp, fp, fn, similarity, scores of the true positives."""
    D, G = dt_datas.shape[0], gt_datas.shape[0]
    scores, dt_alpha, gt_alpha = dt_datas[:, -1], dt_datas[:, 4], gt_datas[:, 4]
    ignored_det = np.asarray(ignored_det)
    assigned = np.zeros(D, dtype=bool)
    below = (scores < thresh) if compute_fp else np.zeros(D, dtype=bool)
    usable = (ignored_det != -1) & ~below
    tp = fp = fn = 0
    similarity = 0
    tp_scores, deltas = [], []
    for i in range(G):
        if ignored_gt[i] == -1:
            continue
        cand = usable & ~assigned & (overlaps[:, i] > min_overlap) if D else np.zeros(0, dtype=bool)
        det = -1
        if cand.any():
            if not compute_fp:
                # highest score wins, first one on ties; scores at or below the sentinel never match
                sc = np.where(cand & (scores > _NO_DET), scores, -np.inf)
                if np.isfinite(sc).any():
                    det = int(np.argmax(sc))
            else:
                real = cand & (ignored_det == 0)
                if real.any():       # a countable detection: the largest overlap, first one on ties
                    det = int(np.argmax(np.where(real, overlaps[:, i], -np.inf)))
                else:                # only neutral detections overlap: the first of them
                    det = int(np.argmax(cand & (ignored_det == 1)))
        if det < 0:
            if ignored_gt[i] == 0:
                fn += 1
        elif ignored_gt[i] == 1 or ignored_det[det] == 1:
            assigned[det] = True
        else:
            tp += 1
            tp_scores.append(scores[det])
            if compute_aos:
                deltas.append(gt_alpha[i] - dt_alpha[det])
            assigned[det] = True
    if compute_fp:
        fp = int(np.count_nonzero(~(assigned | (ignored_det == -1) | (ignored_det == 1) | below)))
        nstuff = 0
        if metric == 0 and dc_bboxes.shape[0]:
            ov = image_box_overlap(dt_datas[:, :4], dc_bboxes, 0)
            for c in range(dc_bboxes.shape[0]):
                hit = ~assigned & (ignored_det == 0) & ~below & (ov[:, c] > min_overlap)
                nstuff += int(np.count_nonzero(hit))
                assigned |= hit
        fp -= nstuff
        if compute_aos:
            if tp > 0 or fp > 0:
                tmp = np.zeros((fp + len(deltas),))
                for j, dlt in enumerate(deltas):
                    tmp[j + fp] = (1.0 + np.cos(dlt)) / 2.0
                similarity = np.sum(tmp)
            else:
                similarity = -1
    return tp, fp, fn, similarity, np.array(tp_scores, dtype=np.float64)


def get_thresholds(scores, num_gt, num_sample_pts=41):
    """Scores at which recall crosses the 41 sample points."""
    scores = np.sort(np.asarray(scores))[::-1]
    current_recall = 0
    out = []
    last = len(scores) - 1
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < last else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < last:
            continue
        out.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return out


def _prepare_data(gt_annos, dt_annos, current_class, difficulty):
    gt_list, dt_list, ign_gt, ign_dt, dcs = [], [], [], [], []
    total_valid = 0
    for g, d in zip(gt_annos, dt_annos):
        nvalid, ig, idt, dc = clean_data(g, d, current_class, difficulty)
        ign_gt.append(np.array(ig, dtype=np.int64))
        ign_dt.append(np.array(idt, dtype=np.int64))
        dcs.append(np.stack(dc, 0).astype(np.float64) if len(dc) else np.zeros((0, 4), dtype=np.float64))
        total_valid += nvalid
        gt_list.append(np.concatenate([g["bbox"], g["alpha"][..., np.newaxis]], 1))
        dt_list.append(np.concatenate([d["bbox"], d["alpha"][..., np.newaxis], d["score"][..., np.newaxis]], 1))
    return gt_list, dt_list, ign_gt, ign_dt, dcs, total_valid


def eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False,
                  num_parts=50, overlap_fns=None):
    """precision / recall / orientation arrays [class, difficulty, overlap tier, 41]."""
    assert len(gt_annos) == len(dt_annos)
    overlaps, _, _, _ = calculate_iou_partly(dt_annos, gt_annos, metric, num_parts, overlap_fns)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    nframes = len(gt_annos)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            gt_list, dt_list, ign_gt, ign_dt, dcs, total_valid = _prepare_data(gt_annos, dt_annos, current_class, difficulty)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                tp_scores = []
                for i in range(nframes):
                    tp_scores += compute_statistics(overlaps[i], gt_list[i], dt_list[i], ign_gt[i], ign_dt[i], dcs[i],
                                                    metric, min_overlap, 0.0, False)[4].tolist()
                thresholds = np.array(get_thresholds(np.array(tp_scores), total_valid))
                pr = np.zeros([len(thresholds), 4])
                for i in range(nframes):
                    for t, thresh in enumerate(thresholds):
                        tp, fp, fn, sim, _ = compute_statistics(overlaps[i], gt_list[i], dt_list[i], ign_gt[i], ign_dt[i],
                                                                dcs[i], metric, min_overlap, thresh, True, compute_aos)
                        pr[t, 0] += tp
                        pr[t, 1] += fp
                        pr[t, 2] += fn
                        if sim != -1:
                            pr[t, 3] += sim
                nt = len(thresholds)
                with np.errstate(divide="ignore", invalid="ignore"):
                    recall[m, l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, l, k, :nt] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                for i in range(nt):   # monotone envelope from the right (over all 41 slots, zeros included)
                    precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                    recall[m, l, k, i] = np.max(recall[m, l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP_v2(prec):
    """11-point AP (every 4th of the 41 recall samples) in percent."""
    return prec[..., 0::4].sum(-1) / 11 * 100


def do_eval_v2(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, difficultys=(0, 1, 2),
               compute_bbox=True, overlap_fns=None):
    """min_overlaps [tier, metric, class] -> mAP arrays [class, difficulty, tier] for bbox, bev, 3d, aos."""
    mAP_bbox = None
    if compute_bbox:
        ret = eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps, compute_aos,
                            overlap_fns=overlap_fns)
        mAP_bbox = get_mAP_v2(ret["precision"])
    ret = eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps, compute_aos,
                        overlap_fns=overlap_fns)
    mAP_bev = get_mAP_v2(ret["precision"])
    mAP_aos = get_mAP_v2(ret["orientation"]) if compute_aos else None
    ret = eval_class_v3(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps, overlap_fns=overlap_fns)
    mAP_3d = get_mAP_v2(ret["precision"])
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


def do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos, overlap_fns=None):
    """overlap_ranges [3 = (lo, hi, count), metric, class] -> means over the `count` tiers."""
    min_overlaps = np.zeros([10, *overlap_ranges.shape[1:]])
    for i in range(overlap_ranges.shape[1]):
        for j in range(overlap_ranges.shape[2]):
            lo, hi, cnt = overlap_ranges[:, i, j]
            min_overlaps[:, i, j] = np.linspace(lo, hi, int(cnt))
    res = do_eval_v2(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, overlap_fns=overlap_fns)
    return tuple(None if r is None else r.mean(-1) for r in res)


def _line(value):
    s = io.StringIO()
    print(value, file=s)
    return s.getvalue()


def _class_ids(current_classes):
    name_to_class = {v: n for n, v in CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [name_to_class[c] if isinstance(c, str) else c for c in current_classes]


def _has_alpha(dt_annos):
    for anno in dt_annos:
        if anno['alpha'].shape[0] != 0:
            return bool(anno['alpha'][0] != -10)
    return False


def official_min_overlaps():
    """The reference's 6 IoU tiers [tier, metric (bbox, bev, 3d), class] (eval.py:843-861): Pedestrian
    (column 1) runs 0.50 ... 0.75 in bev / 3d and 0.70 ... 0.95 for 2D boxes."""
    base = np.array([[0.7, 0.0, 0.5, 0.7, 0.5, 0.7, 0.7, 0.7]] * 3)
    tiers = []
    lo = np.array([[0.7, 0.7, 0.5, 0.7, 0.5, 0.5, 0.5, 0.5],
                   [0.5, 0.5, 0.25, 0.5, 0.25, 0.5, 0.5, 0.5],
                   [0.5, 0.5, 0.25, 0.5, 0.25, 0.5, 0.5, 0.5]])
    tiers.append(lo)
    for ped in (0.55, 0.60, 0.65, 0.70, 0.75):
        t = base.copy()
        t[0, 1] = round(ped + 0.20, 2)
        t[1, 1] = t[2, 1] = ped
        tiers.append(t)
    return np.stack(tiers, axis=0)


def get_official_eval_result(gt_annos, dt_annos, current_classes, difficultys=[0, 1, 2], return_data=True,
                             compute_bbox=True, overlap_fns=None):
    """The report `train.py evaluate` prints (train.py:899-901) plus the mAP arrays."""
    current_classes = _class_ids(current_classes)
    min_overlaps = official_min_overlaps()[:, :, current_classes]
    compute_aos = _has_alpha(dt_annos)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval_v2(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos,
                                                difficultys, compute_bbox=compute_bbox, overlap_fns=overlap_fns)
    result = ''
    for j, curcls in enumerate(current_classes):
        for i in range(min_overlaps.shape[0]):
            result += _line(f"{CLASS_TO_NAME[curcls]} " + "AP@{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j]))
            rows = []
            if compute_bbox:
                rows.append(("bbox", mAPbbox))
            rows += [("bev ", mAPbev), ("3d  ", mAP3d)]
            if compute_aos:
                rows.append(("aos ", mAPaos))
            for tag, arr in rows:
                result += _line(f"{tag} AP:{arr[j, 0, i]:.2f}, {arr[j, 1, i]:.2f}, {arr[j, 2, i]:.2f}")
    if return_data:
        return result, mAPbbox, mAPbev, mAP3d, mAPaos
    return result


_COCO_RANGE = {0: [0.5, 0.95, 10], 1: [0.25, 0.7, 10], 2: [0.25, 0.7, 10], 3: [0.5, 0.95, 10], 4: [0.25, 0.7, 10],
               5: [0.5, 0.95, 10], 6: [0.5, 0.95, 10], 7: [0.5, 0.95, 10]}


def get_coco_eval_result(gt_annos, dt_annos, current_classes, overlap_fns=None):
    current_classes = _class_ids(current_classes)
    overlap_ranges = np.zeros([3, 3, len(current_classes)])
    for i, curcls in enumerate(current_classes):
        overlap_ranges[:, :, i] = np.array(_COCO_RANGE[curcls])[:, np.newaxis]
    compute_aos = _has_alpha(dt_annos)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges,
                                                        compute_aos, overlap_fns=overlap_fns)
    result = ''
    for j, curcls in enumerate(current_classes):
        o_range = np.array(_COCO_RANGE[curcls])[[0, 2, 1]]
        o_range[1] = (o_range[2] - o_range[0]) / (o_range[1] - 1)
        result += _line(f"{CLASS_TO_NAME[curcls]} " + "coco AP@{:.2f}:{:.2f}:{:.2f}:".format(*o_range))
        rows = [("bbox", mAPbbox), ("bev ", mAPbev), ("3d  ", mAP3d)]
        if compute_aos:
            rows.append(("aos ", mAPaos))
        for tag, arr in rows:
            result += _line(f"{tag} AP:{arr[j, 0]:.2f}, {arr[j, 1]:.2f}, {arr[j, 2]:.2f}")
    return result
