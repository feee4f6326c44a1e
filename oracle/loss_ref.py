"""TEST INFRASTRUCTURE: CPU restatement of the reference's training loss at the head maps.

Follows model/voxelnet.py:922-1049 (VoxelNet.call, training branch) and the functions it calls --
prepare_loss_weights :461-512, create_loss :74-155, add_sin_difference :63-69,
_sigmoid_cross_entropy_with_logits :237-260, sigmoid_focal_classification_loss :262-364,
WeightedSmoothL1LocalizationLoss.call :407-459, get_direction_target :38-46,
weighted_softmax_classification_loss :180-235 -- op for op in torch float32 on the CPU; the gradient with
respect to the head maps comes from autograd.  PARITY UNPINNED: the reference's versions are TensorFlow
graphs (TF 2.2 is not installable here) and the repository ships no loss values to compare with; a float64
numpy evaluation of the same formulas cross-checks the values in tests/test_loss.py.
"""
import numpy as np
import torch


def loss_tensors(second_cfg, box, cls, dr, labels, reg_targets, anchors, dtype=torch.float32):
    """The loss graph on torch tensors (box / cls / dr may carry an autograd history: the whole-network gradient
    check of oracle/train_ref.py).  Returns a dict of 0-d tensors."""
    s = second_cfg
    B = labels.shape[0]
    labels_t = torch.tensor(np.asarray(labels), dtype=torch.int64)
    reg_t = torch.tensor(np.asarray(reg_targets), dtype=dtype).reshape(B, -1, 7)
    anc = torch.tensor(np.asarray(anchors), dtype=dtype).reshape(1, -1, 7)

    # prepare_loss_weights
    cared = labels_t >= 0
    positives = (labels_t > 0).to(dtype)
    negatives = (labels_t == 0).to(dtype)
    cls_weights = negatives * s["neg_class_weight"] + positives * s["pos_class_weight"]
    reg_weights = positives.clone()
    if s["loss_norm_type"] == "NormByNumPositives":
        pos_normalizer = positives.sum(1, keepdim=True)
        reg_weights = reg_weights / torch.clamp(pos_normalizer, 1.0, 100000.0)
        cls_weights = cls_weights / torch.clamp(pos_normalizer, 1.0, 100000.0)
    cls_targets = labels_t * cared.to(labels_t.dtype)

    # create_loss
    num_class = s["num_class"]
    bp = box.reshape(B, -1, 7)
    cp = cls.reshape(B, -1, num_class)
    one_hot = torch.nn.functional.one_hot(cls_targets, num_class + 1).to(dtype)[..., 1:]
    rt = reg_t
    if s["encode_rad_error_by_sin"]:
        rad_pred = torch.sin(bp[..., -1:]) * torch.cos(rt[..., -1:])
        rad_tg = torch.cos(bp[..., -1:]) * torch.sin(rt[..., -1:])
        bp = torch.cat([bp[..., :-1], rad_pred], -1)
        rt = torch.cat([rt[..., :-1], rad_tg], -1)
    l1 = s["loss"]["localization_loss"]["weighted_smooth_l1"]
    sigma = l1["sigma"]
    diff = torch.tensor(l1["code_weight"], dtype=dtype).reshape(1, 1, -1) * (bp - rt)
    ad = diff.abs()
    lt = (ad <= 1 / sigma ** 2).to(dtype)
    loc_loss = (lt * 0.5 * (ad * sigma) ** 2 + (ad - 0.5 / sigma ** 2) * (1.0 - lt)) * reg_weights.unsqueeze(-1)
    focal = s["loss"]["classification_loss"]["weighted_sigmoid_focal"]
    ce = torch.clamp(cp, 0, 10000) - cp * one_hot + torch.log1p(torch.exp(-cp.abs()))
    prob = torch.sigmoid(cp)
    p_t = one_hot * prob + (1 - one_hot) * (1 - prob)
    mod = torch.pow(1.0 - p_t, focal["gamma"]) if focal["gamma"] else 1.0
    aw = one_hot * focal["alpha"] + (1 - one_hot) * (1 - focal["alpha"]) if focal["alpha"] is not None else 1.0
    cls_loss = mod * aw * ce * cls_weights.unsqueeze(2)

    loc_red = loc_loss.sum() / B * s["loss"]["localization_weight"]
    cls_red = cls_loss.sum() / B * s["loss"]["classification_weight"]
    if num_class == 1:     # _get_pos_neg_loss, model/voxelnet.py:48-61: by the anchor's label ...
        cls_pos = ((labels_t > 0).to(dtype) * cls_loss.reshape(B, -1)).sum() / B
        cls_neg = ((labels_t == 0).to(dtype) * cls_loss.reshape(B, -1)).sum() / B
    else:                  # ... or column 0 against the other class columns
        cls_pos = cls_loss[..., 1:].sum() / B
        cls_neg = cls_loss[..., 0].sum() / B
    loss = loc_red + cls_red
    dir_red = torch.zeros((), dtype=dtype)
    if s["use_direction_classifier"]:
        rot_gt = reg_t[..., -1] + anc[..., -1]
        dir_t = (rot_gt > 0).to(torch.int64)
        logits = dr.reshape(B, -1, 2)
        w = (labels_t > 0).to(dtype)
        w = w / torch.clamp(w.sum(-1, keepdim=True), 1.0, 9999999.0)
        ce_dir = torch.nn.functional.cross_entropy(logits.reshape(-1, 2), dir_t.reshape(-1), reduction="none")
        dir_red = (ce_dir.reshape(B, -1) * w).sum() / B * s["direction_loss_weight"]
        loss = loss + dir_red
    return {"loss": loss, "loc_loss_reduced": loc_red, "cls_loss_reduced": cls_red, "dir_loss_reduced": dir_red,
            "cls_pos_loss": cls_pos, "cls_neg_loss": cls_neg, "num_positives": (labels_t > 0).sum()}


def training_loss(second_cfg, box_preds, cls_preds, dir_cls_preds, labels, reg_targets, anchors, dtype=torch.float32):
    """box_preds [B,H,W,na*7], cls_preds [B,H,W,na], dir_cls_preds [B,H,W,na*2] (numpy), labels [B,A] int,
    reg_targets [B,A,7], anchors [A,7].  Returns (dict of python floats, dict of numpy gradients)."""
    box = torch.tensor(np.asarray(box_preds), dtype=dtype, requires_grad=True)
    cls = torch.tensor(np.asarray(cls_preds), dtype=dtype, requires_grad=True)
    dr = torch.tensor(np.asarray(dir_cls_preds), dtype=dtype, requires_grad=True) if dir_cls_preds is not None else None
    t = loss_tensors(second_cfg, box, cls, dr, labels, reg_targets, anchors, dtype)
    t["loss"].backward()
    vals = {k: float(v.detach()) for k, v in t.items() if k != "num_positives"}
    vals["num_positives"] = int(t["num_positives"])
    grads = {"box_preds_grad": box.grad.numpy(), "cls_preds_grad": cls.grad.numpy(),
             "dir_cls_preds_grad": (dr.grad.numpy() if dr is not None and dr.grad is not None
                                    else (np.zeros_like(np.asarray(dir_cls_preds)) if dir_cls_preds is not None else None))}
    return vals, grads
