"""Golden vectors for the training-side target assignment (SURVEY section 8f, f3 -- data half), produced by
RUNNING the reference's own create_target_np / assign / iou_jit / rbbox2d_to_near_bbox (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_targets.py   ->  tests/golden/ref_targets.npz

The reference's `second_box_encode` is written with TensorFlow ops (load_data.py:125-203) and cannot run
here; for this run `load_data.box_encoding_fn` is pointed at the numpy restatement below (same formulas,
float32), so `bbox_targets` in the fixture pins the ROWS that receive a target and the reference's
gather order, while the encode arithmetic itself is pinned by the round trip through the reference's
own numpy `second_box_decode` (also recorded here).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ld, ehf = ref_shim.load_reference()


def encode_np(boxes, anchors):
    a = np.asarray(anchors, np.float32)
    g = np.asarray(boxes, np.float32)
    xa, ya, za, wa, la, ha, ra = (a[..., i:i + 1] for i in range(7))
    xg, yg, zg, wg, lg, hg, rg = (g[..., i:i + 1] for i in range(7))
    za = za + ha / np.float32(2)
    zg = zg + hg / np.float32(2)
    d = np.sqrt(la ** 2 + wa ** 2)
    return np.concatenate([(xg - xa) / d, (yg - ya) / d, (zg - za) / ha, np.log(wg / wa), np.log(lg / la),
                           np.log(hg / ha), rg - ra], axis=-1)


ld.box_encoding_fn = encode_np

# the shipped anchor grid (64 x 80 x 2 rotations) through the reference's own generator
feature_size = [1, 64, 80]
anchors = ld.create_anchors_3d_stride(feature_size, [0.6, 0.8, 1.73], [0.08, 0.08, 0.0], [0.08, -2.56, -1.465],
                                      [0, 1.57], np.float32).reshape(-1, 7)
rng = np.random.default_rng(31)
out = {"anchors": anchors}
cfg_ta = {"sample_positive_fraction": "None", "rpn_batch_size": 512}
cases = {
    "three": np.array([[2.0, 0.3, -0.6, 0.62, 0.85, 1.75, 0.1], [4.1, -1.2, -0.55, 0.55, 0.7, 1.6, 1.5],
                       [5.9, 2.2, -0.7, 0.7, 0.9, 1.8, -2.9]], np.float32),
    "offgrid": np.array([[2.0, 0.3, -0.6, 0.62, 0.85, 1.75, 0.1], [30.0, 30.0, 0.0, 0.6, 0.8, 1.7, 0.0]], np.float32),
    "empty": np.zeros((0, 7), np.float32),
    "tie": np.array([[1.00, 0.00, -0.6, 0.6, 0.8, 1.73, 0.0], [1.04, 0.00, -0.6, 0.6, 0.8, 1.73, 0.0]], np.float32),
}
for name, gt in cases.items():
    for masked in (False, True):
        mask = None
        if masked:
            mask = rng.uniform(size=anchors.shape[0]) < 0.6
        gt_classes = np.ones(len(gt), np.int32)
        r = ld.assign(anchors, gt, mask, gt_classes, 0.5, 0.35, cfg_ta)
        tag = f"{name}_{'mask' if masked else 'all'}"
        out[tag + "_gt"] = gt
        if masked:
            out[tag + "_anchors_mask"] = mask
        for k, v in r.items():
            out[tag + "_" + k] = np.zeros(0) if v is None else np.asarray(v)
        out[tag + "_overlap_is_none"] = np.array(r["assigned_anchors_overlap"] is None)
# the similarity pieces on their own
b1 = np.concatenate([rng.uniform(0, 6, (40, 2)), rng.uniform(0.3, 2.0, (40, 2)), rng.uniform(-4, 4, (40, 1))], 1).astype(np.float32)
b2 = np.concatenate([rng.uniform(0, 6, (9, 2)), rng.uniform(0.3, 2.0, (9, 2)), rng.uniform(-4, 4, (9, 1))], 1).astype(np.float32)
out["rb1"], out["rb2"] = b1, b2
out["near1"] = ld.rbbox2d_to_near_bbox(b1)
out["sim"] = ld.nearest_iou_similarity(b1, b2)
# encode -> the reference's numpy decode gives the boxes back
pos = out["three_all_assigned_anchors_inds"]
enc = out["three_all_bbox_targets"][pos]
out["three_all_decoded"] = ehf.second_box_decode(enc, anchors[pos])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_targets.npz"), **out)
print(len(out), {k: out[k].shape for k in sorted(out) if k.startswith("three_all")})
