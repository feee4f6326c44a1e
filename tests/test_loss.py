"""Training loss at the head maps: HIP kernel vs the torch-CPU restatement of the reference's
TensorFlow loss graph (oracle/loss_ref.py; parity unpinned -- TF is not installable), and the restatement
itself against a float64 numpy evaluation of the formulas and against finite differences."""
import numpy as np
import pytest

import pp_amd as pp
from oracle import loss_ref

SECOND = pp.config.pedestrian_d435i_config(1)["model"]["second"]


def _random_problem(rng, B=2, H=6, W=5, na=2, npos=7):
    A = H * W * na
    box = rng.normal(0, 0.4, (B, H, W, na * 7)).astype(np.float32)
    cls = rng.normal(-1.0, 1.5, (B, H, W, na)).astype(np.float32)
    dr = rng.normal(0, 1.0, (B, H, W, na * 2)).astype(np.float32)
    labels = rng.choice([-1, 0, 0, 0], size=(B, A)).astype(np.int32)
    reg = np.zeros((B, A, 7), np.float32)
    for b in range(B):
        pos = rng.choice(A, npos if b == 0 else 0, replace=False)      # frame 1: no positives (normaliser clip)
        labels[b, pos] = 1
        reg[b, pos] = rng.normal(0, 0.5, (len(pos), 7)).astype(np.float32)
    anchors = np.zeros((A, 7), np.float32)
    anchors[:, 6] = np.tile([0.0, 1.57], A // 2)
    return box, cls, dr, labels, reg, anchors


def _numpy64_loss(box, cls, dr, labels, reg, anchors):
    """The same formulas written independently in float64 numpy (values only)."""
    s = SECOND
    B = labels.shape[0]
    bp, cp, dl = box.reshape(B, -1, 7).astype(np.float64), cls.reshape(B, -1).astype(np.float64), dr.reshape(B, -1, 2).astype(np.float64)
    pos, neg = (labels > 0).astype(np.float64), (labels == 0).astype(np.float64)
    norm = np.clip(pos.sum(1, keepdims=True), 1.0, 1e5)
    t = pos
    ce = np.clip(cp, 0, 10000) - cp * t + np.log1p(np.exp(-np.abs(cp)))
    p = 1 / (1 + np.exp(-cp))
    p_t = t * p + (1 - t) * (1 - p)
    cls_l = (1 - p_t) ** 2.0 * (t * 0.25 + (1 - t) * 0.75) * ce * (pos + neg) / norm
    rg = reg.astype(np.float64)
    d = bp - rg
    d[..., 6] = np.sin(bp[..., 6] - rg[..., 6])
    ad = np.abs(d)
    l1 = np.where(ad <= 1 / 9.0, 0.5 * (ad * 3.0) ** 2, ad - 0.5 / 9.0) * (pos / norm)[..., None]
    c = ((rg[..., 6] + anchors[None, :, 6]) > 0).astype(np.int64)
    lse = np.log(np.exp(dl).sum(-1))
    dir_l = (lse - np.take_along_axis(dl, c[..., None], -1)[..., 0]) * pos / np.clip(pos.sum(1, keepdims=True), 1.0, 9999999.0)
    loc, cl, di = l1.sum() / B * 1.5, cls_l.sum() / B * 1.0, dir_l.sum() / B * 0.5
    return {"loss": loc + cl + di, "loc_loss_reduced": loc, "cls_loss_reduced": cl, "dir_loss_reduced": di,
            "cls_pos_loss": (cls_l * pos).sum() / B, "cls_neg_loss": (cls_l * neg).sum() / B}


def test_restatement_values_match_float64_formulas():
    rng = np.random.default_rng(5)
    prob = _random_problem(rng)
    vals, grads = loss_ref.training_loss(SECOND, *prob)
    want = _numpy64_loss(*prob)
    for k, v in want.items():
        assert abs(vals[k] - v) <= 2e-6 * max(1.0, abs(v)), (k, vals[k], v)
    assert vals["num_positives"] == 7 and vals["loc_loss_reduced"] > 0 and vals["dir_loss_reduced"] > 0
    # ignored anchors (-1) and background anchors get no box / direction gradient
    B, A = prob[3].shape
    gbox = grads["box_preds_grad"].reshape(B, A, 7)
    assert not gbox[prob[3] <= 0].any() and gbox[prob[3] > 0].any()
    assert not grads["cls_preds_grad"].reshape(B, A)[prob[3] < 0].any()


def test_restatement_gradient_matches_finite_differences():
    import torch
    rng = np.random.default_rng(6)
    box, cls, dr, labels, reg, anchors = _random_problem(rng, B=1, H=3, W=2, npos=4)
    _, g = loss_ref.training_loss(SECOND, box, cls, dr, labels, reg, anchors, dtype=torch.float64)
    eps = 1e-6
    for name, arr, key in (("box", box, "box_preds_grad"), ("cls", cls, "cls_preds_grad"), ("dir", dr, "dir_cls_preds_grad")):
        a64 = arr.astype(np.float64)
        for idx in rng.choice(a64.size, 6, replace=False):
            up, dn = a64.copy().ravel(), a64.copy().ravel()
            up[idx] += eps
            dn[idx] -= eps
            def run(v):
                rep = {"box": (v.reshape(arr.shape), cls, dr), "cls": (box, v.reshape(arr.shape), dr),
                       "dir": (box, cls, v.reshape(arr.shape))}[name]
                return loss_ref.training_loss(SECOND, *rep, labels, reg, anchors, dtype=torch.float64)[0]["loss"]
            num = (run(up) - run(dn)) / (2 * eps)
            assert abs(num - g[key].ravel()[idx]) <= 1e-6 + 1e-5 * abs(num), (name, idx, num, g[key].ravel()[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["boxes", "no_positives"])
def test_head_loss_kernel_matches_restatement(case):
    B = 2
    eng = pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=20000)
    d = eng.d
    eng.load_weights(pp.weights.init_weights(d, seed=7))
    frames = [pp.synth.d435i_cloud(300 + i) for i in range(B)]
    eng.detect(frames)
    im = eng.intermediates()
    anchors = pp.anchors.build_anchors(d)
    rng = np.random.default_rng(12)
    labels = np.zeros((B, d.num_anchors), np.int32)
    reg = np.zeros((B, d.num_anchors, 7), np.float32)
    if case == "boxes":
        for b in range(B):
            gt = np.concatenate([rng.uniform([0.8, -2.0, -0.9], [6.0, 2.0, -0.3], (3, 3)), rng.uniform(0.5, 0.9, (3, 2)),
                                 rng.uniform(1.5, 1.9, (3, 1)), rng.uniform(-3.1, 3.1, (3, 1))], 1).astype(np.float32)
            mask = im["anchors_mask"][b].astype(bool)
            r = pp.target_assigner.assign(anchors, gt, mask, np.ones(3, np.int32), 0.5, 0.35,
                                          {"sample_positive_fraction": "None", "rpn_batch_size": 512})
            labels[b], reg[b] = r["labels"], r["bbox_targets"]
        assert (labels > 0).sum() > 20 and (labels == -1).sum() > 100
    else:
        labels[0, ::3] = -1
    got = eng.head_loss(labels, reg)
    s = d.config["model"]["second"]
    vals, grads = loss_ref.training_loss(s, im["box_preds"], im["cls_preds"], im["dir_cls_preds"], labels, reg, anchors)
    for k in ("loss", "loc_loss_reduced", "cls_loss_reduced", "dir_loss_reduced", "cls_pos_loss", "cls_neg_loss"):
        assert abs(got[k] - vals[k]) <= 2e-5 * max(1e-3, abs(vals[k])), (k, got[k], vals[k])
    assert got["num_positives"] == vals["num_positives"]
    for k in ("box_preds_grad", "cls_preds_grad", "dir_cls_preds_grad"):
        np.testing.assert_allclose(got[k], grads[k], rtol=2e-4, atol=1e-8, err_msg=k)
    assert not got["head_grad"][:, :, 20:].any(), "pad columns of the head map get no gradient"
    again = eng.head_loss(labels, reg)
    assert again["loss"] == got["loss"] and again["head_grad"].tobytes() == got["head_grad"].tobytes(), "bit-reproducible"
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["two_classes", "three_classes_no_direction"])
def test_head_loss_kernel_config_branches(variant):
    """The training branch of the reference is config-driven (model/voxelnet.py:74-155, :922-1049): several classes
    (one sigmoid logit per class and anchor, one-hot targets without the background column) and a model without the
    direction head -- the kernel against the restatement, values and gradients."""
    import copy
    B = 2
    cfg = copy.deepcopy(pp.config.tiny_config(B))
    s = cfg["model"]["second"]
    if variant == "two_classes":
        s["num_class"] = 2
    else:
        s["num_class"] = 3
        s["use_direction_classifier"] = False
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=4096)
    d = eng.d
    eng.load_weights(pp.weights.init_weights(d, seed=3))
    rng = np.random.default_rng(31)
    frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (700, 400)]
    eng.detect(frames)
    im = eng.intermediates()
    anchors = pp.anchors.build_anchors(d)
    labels = rng.choice([-1, 0, 0, 0, 0], size=(B, d.num_anchors)).astype(np.int32)
    reg = np.zeros((B, d.num_anchors, 7), np.float32)
    for b in range(B):
        pos = rng.choice(d.num_anchors, 25, replace=False)
        labels[b, pos] = rng.integers(1, s["num_class"] + 1, 25)          # every class appears
        reg[b, pos] = rng.normal(0, 0.4, (25, 7)).astype(np.float32)
    got = eng.head_loss(labels, reg)
    vals, grads = loss_ref.training_loss(s, im["box_preds"], im["cls_preds"], im.get("dir_cls_preds"), labels, reg, anchors)
    keys = ["loss", "loc_loss_reduced", "cls_loss_reduced", "cls_pos_loss", "cls_neg_loss"]
    if s["use_direction_classifier"]:
        keys.append("dir_loss_reduced")
    for k in keys:
        assert abs(got[k] - vals[k]) <= 2e-5 * max(1e-3, abs(vals[k])), (k, got[k], vals[k])
    assert got["num_positives"] == vals["num_positives"] == 50
    for k in ("box_preds_grad", "cls_preds_grad") + (("dir_cls_preds_grad",) if s["use_direction_classifier"] else ()):
        np.testing.assert_allclose(got[k], grads[k], rtol=2e-4, atol=1e-8, err_msg=k)
    na = d.num_anchor_per_loc
    used = na * (7 + s["num_class"] + (2 if s["use_direction_classifier"] else 0))
    assert not got["head_grad"][:, :, used:].any(), "columns past the heads get no gradient"
    eng.close()


@pytest.mark.gpu
def test_head_loss_argument_errors():
    eng = pp.Engine(pp.config.tiny_config(1), max_batch=1)
    eng.load_weights(pp.weights.init_weights(eng.d, seed=1))
    with pytest.raises(ValueError):
        eng.head_loss(np.zeros((1, 3), np.int32), np.zeros((1, 3, 7), np.float32))
    with pytest.raises(RuntimeError):
        eng.head_loss(np.zeros((2, eng.d.num_anchors), np.int32), np.zeros((2, eng.d.num_anchors, 7), np.float32))
    eng.close()
