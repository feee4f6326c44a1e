"""Golden vectors for the AP-evaluator row, produced by RUNNING the reference's
own second/utils/eval.py and the rotated-IoU device functions of nms_gpu.py as plain Python
(build container only; numba decorators are identity stubs, tools/ref_shim.py).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_eval.py
      -> tests/golden/ref_rotate_iou.npz   boxes / query boxes / [N,K] results for criterion -1, 0, 1, 2
      -> tests/golden/ref_kitti_eval.npz   synthetic gt / dt annos, overlaps, per-metric mAP arrays, report text

rotate_iou_gpu_eval launches a numba-CUDA kernel (nms_gpu.py:618-653) that cannot run here; the generator
loops the kernel's own indexing (:611-613: out[n, k] = devRotateIoUEval(query[k], boxes[n])) over the
reference's device function and installs that loop as eval.rotate_iou_gpu_eval.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ev, ng = ref_shim.load_reference_eval()


def rotate_iou_eval_loop(boxes, query_boxes, criterion=-1, device_id=0):
    box_dtype = boxes.dtype
    b = boxes.astype(np.float32)
    q = query_boxes.astype(np.float32)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    for n in range(b.shape[0]):
        for k in range(q.shape[0]):
            out[n, k] = ng.devRotateIoUEval(q[k], b[n], criterion)
    return out.astype(box_dtype)


ev.rotate_iou_gpu_eval = rotate_iou_eval_loop
rng = np.random.default_rng(2024)


def rboxes(n, spread=3.0):
    return np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(0.3, 2.5, (n, 2)),
                           rng.uniform(-3.5, 3.5, (n, 1))], axis=1).astype(np.float32)


# ---- rotated IoU: random clouds of boxes + hand-made edge cases ----
b = rboxes(40)
q = rboxes(33)
edge = np.array([[0, 0, 2, 1, 0.0], [0, 0, 2, 1, 0.0],            # identical, axis aligned
                 [0, 0, 2, 1, 0.3], [0, 0, 1, 0.5, 0.3],          # contained, same angle
                 [5, 5, 1, 1, 0.0], [6, 5, 1, 1, 0.0],            # touching edges
                 [0, 0, 2, 2, np.pi / 4], [0, 0, 2, 2, 0.0],      # octagon
                 [10, 10, 1, 1, 0.2], [-10, -10, 1, 1, 0.2]],     # disjoint
                dtype=np.float32)
b = np.concatenate([b, edge[0::2]], 0)
q = np.concatenate([q, edge[1::2]], 0)
out = {"boxes": b, "qboxes": q}
for crit in (-1, 0, 1, 2):
    out[f"iou_c{crit}"] = rotate_iou_eval_loop(b, q, crit)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_rotate_iou.npz"), **out)
print("rotate iou:", out["iou_c-1"].shape, float(out["iou_c-1"].max()), int((out["iou_c-1"] > 0).sum()))


# ---- KITTI-style AP on synthetic annotations (camera coordinates) ----
def make_annos(nframes):
    gts, dts = [], []
    for f in range(nframes):
        ng_ = int(rng.integers(0, 5))
        loc = np.stack([rng.uniform(-2.4, 2.4, ng_), rng.uniform(0.2, 1.4, ng_), rng.uniform(0.8, 6.0, ng_)], 1)
        dims = np.stack([rng.uniform(0.5, 0.9, ng_), rng.uniform(1.5, 1.9, ng_), rng.uniform(0.5, 0.9, ng_)], 1)
        rot = rng.uniform(-np.pi, np.pi, ng_)
        names = np.array(["Pedestrian"] * ng_)
        if ng_ >= 3:
            names[-1] = "Cyclist" if f % 2 else "Person_sitting"
        occ = rng.integers(0, 3, ng_)
        trunc = rng.choice([0.0, 0.1, 0.25, 0.4], ng_)
        bbox_gt = np.stack([100.0 + 40 * np.arange(ng_), np.full(ng_, 100.0), 180.0 + 40 * np.arange(ng_),
                            200.0 + 20 * (f % 3) + rng.uniform(-70, 10, ng_)], 1) if ng_ else np.zeros((0, 4))
        if ng_ >= 2 and f % 3 == 0:
            names[0] = "DontCare"               # exercises the dc_bboxes branch of the 2D metric
        gt = {"name": names, "truncated": trunc, "occluded": occ, "alpha": -np.arctan2(-loc[:, 0], loc[:, 2]) + rot,
              "bbox": bbox_gt, "dimensions": dims, "location": loc, "rotation_y": rot}
        # detections: jittered copies of most gts + false positives
        keep = rng.uniform(size=ng_) < 0.8
        nd_fp = int(rng.integers(0, 3))
        dloc = np.concatenate([loc[keep] + rng.normal(0, 0.05, (int(keep.sum()), 3)),
                               np.stack([rng.uniform(-2.4, 2.4, nd_fp), rng.uniform(0.2, 1.4, nd_fp),
                                         rng.uniform(0.8, 6.0, nd_fp)], 1)], 0)
        ddims = np.concatenate([dims[keep] * rng.uniform(0.95, 1.05, (int(keep.sum()), 3)),
                                np.tile(np.array([[0.7, 1.7, 0.7]]), (nd_fp, 1))], 0)
        drot = np.concatenate([rot[keep] + rng.normal(0, 0.1, int(keep.sum())), rng.uniform(-np.pi, np.pi, nd_fp)], 0)
        nd = dloc.shape[0]
        dt = {"name": np.array(["Pedestrian"] * nd), "truncated": np.zeros(nd), "occluded": np.zeros(nd, dtype=np.int64),
              "alpha": -np.arctan2(-dloc[:, 0], dloc[:, 2]) + drot if nd else np.zeros(0),
              "bbox": (np.concatenate([bbox_gt[keep] + rng.normal(0, 4, (int(keep.sum()), 4)),
                                       np.tile(np.array([[400.0, 200.0, 500.0, 215.0 + 60 * (f % 2)]]), (nd_fp, 1))], 0)
                       if nd else np.zeros((0, 4))),
              "dimensions": ddims, "location": dloc,
              "rotation_y": drot, "score": rng.uniform(0.05, 0.99, nd).astype(np.float32)}
        gts.append(gt)
        dts.append(dt)
    return gts, dts


gts, dts = make_annos(57)   # >= 50: get_split_parts(n, 50) yields empty parts (and the reference fails) below that
res = {}
for i, (g, d) in enumerate(zip(gts, dts)):
    for k, v in g.items():
        res[f"gt_{i}_{k}"] = np.asarray(v)
    for k, v in d.items():
        res[f"dt_{i}_{k}"] = np.asarray(v)
res["nframes"] = np.array(len(gts))
for metric in (1, 2):
    overlaps, parted, tg, td = ev.calculate_iou_partly(dts, gts, metric, num_parts=5)
    for i, o in enumerate(overlaps):
        res[f"ov_m{metric}_{i}"] = o
text, mbbox, mbev, m3d, maos = ev.get_official_eval_result(gts, dts, ["Pedestrian"], compute_bbox=False)
res["official_text"] = np.array(text)
res["official_bev"], res["official_3d"], res["official_aos"] = mbev, m3d, maos
text2, mbbox2, mbev2, m3d2, maos2 = ev.get_official_eval_result(gts, dts, ["Pedestrian", "Cyclist"],
                                                               difficultys=[0, 1, 2], compute_bbox=True)
res["official2_text"] = np.array(text2)
res["official2_bbox"], res["official2_bev"], res["official2_3d"], res["official2_aos"] = mbbox2, mbev2, m3d2, maos2
res["coco_text"] = np.array(ev.get_coco_eval_result(gts, dts, ["Pedestrian"]))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_kitti_eval.npz"), **res)
print(text)
print("coco:", str(res["coco_text"])[:200])

# ---- the predict path's axis-aligned `+1` IoU (libraries/eval_helper_functions.py:553-564, iou_device) ----
# Run as plain Python the float32 + integer-literal sums stay float32 (numpy promotion), whereas numba types
# them float64: the fixture pins the formula (the `+1`, the clamps, the union) to ~1e-7, not the last bit.
_, ehf = ref_shim.load_reference()
ab = np.concatenate([rng.uniform(0, 6, (300, 2)), rng.uniform(0, 6, (300, 2))], 1).astype(np.float32)
ab[:, 2:] = ab[:, :2] + rng.uniform(0.2, 1.5, (300, 2)).astype(np.float32)
ious = np.array([[ehf.iou_device(ab[i], ab[j]) for j in range(0, 300, 7)] for i in range(0, 300, 5)], dtype=np.float64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_iou_device.npz"), boxes=ab, iou=ious,
                    rows=np.arange(0, 300, 5), cols=np.arange(0, 300, 7))
print("iou_device:", ious.shape, float(ious.min()), float(ious.max()))
