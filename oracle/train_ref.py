"""TEST INFRASTRUCTURE: CPU restatement of one training step of the reference (train.py:265-304) up to the
gradients -- VoxelNet.call in training mode (model/voxelnet.py:850-1049) written in torch float32 with autograd.

  PillarFeatureNet   Dense(no bias) -> BatchNormalization(training: batch statistics over ALL P*T rows of the
                     padded tensor, eps 1e-3) -> ReLU -> reduce_max over T      model/pointpillars.py:97-115,199-225
  PointPillarsScatter                                                           model/pointpillars.py:285-341
  RPN                (SeparableConv2D -> BatchNormalization(training) -> ReLU)*, Conv2DTranspose -> BN -> ReLU,
                     concat, 1x1 heads with bias                                model/voxelnet.py:573-717
  losses             oracle/loss_ref.py
PARITY UNPINNED: TensorFlow is not installable here and the reference ships no gradient values; the layer semantics
are those of oracle/nn_ref.py (itself cross-checked between torch and numpy), BatchNorm in training mode normalises
with the biased batch variance (Keras), and reduce_max routes the gradient to the maximal row (torch.amax splits it
between exact ties, as tf.reduce_max does).
"""
import numpy as np
import torch
import torch.nn.functional as Fn

from . import loss_ref, nn_ref

BN_EPS = 1e-3


def _bn_train(x, gamma, beta, dims):
    mean = x.mean(dim=dims, keepdim=True)
    var = x.var(dim=dims, unbiased=False, keepdim=True)
    shape = [1] * x.dim()
    cdim = [d for d in range(x.dim()) if d not in dims][0]
    shape[cdim] = -1
    return (x - mean) / torch.sqrt(var + BN_EPS) * gamma.reshape(shape) + beta.reshape(shape), mean.flatten(), var.flatten()


def training_step(d, w, example, labels, reg_targets, anchors, num_threads=None, dtype=torch.float32, margins=None,
                  forced=None, record=None):
    """d: config.Derived; w: Keras-layout dict of numpy arrays; example: the oracle's merged batch (padded voxels,
    num_points, coors[b,z,y,x], ...).  Returns (loss dict of floats, gradient dict name -> numpy for every trainable
    tensor, batch statistics dict name -> (mean, biased var)).  dtype=torch.float64: the same graph in double
    precision (the yardstick that tells float32 round-off of the restatement apart from an error of the kernels).
    margins: a dict that receives, per BatchNorm + ReLU layer, the smallest |pre-ReLU value| of the step (and for the PFN
    the smallest positive gap between a pillar's largest and second-largest row): ReLU and max are not differentiable
    there, so an implementation whose round-off puts such an element on the other side returns a gradient that differs
    by that element's whole contribution -- a property of the problem, not an error (tools/fuzz_train.py uses it).
    forced: the decisions another implementation took at those points (Trainer.decisions(): {"pfn": winning row per
    (pillar, channel) in this example's pillar order, -1 a padded row, -2 none; "<layer>/bn": ReLU mask [B, pixels, C],
    transposed convolutions [B, input pixels, k, k, C]}): the graph then takes the SAME branches (relu(x) -> x * mask, max ->
    the given row), which makes its gradient a smooth function of the inputs and the comparison sharp.
    record: a dict that receives this run's own decisions in that format (self-check of the forced path)."""
    if num_threads:
        torch.set_num_threads(num_threads)
    voxels, num_points, coors = example[0], example[1], example[2]
    B = int(example[6].shape[0])
    train_names = [k for k in w if not k.endswith(("moving_mean", "moving_variance"))]
    t = {k: torch.tensor(np.asarray(w[k], dtype=np.float32), dtype=dtype, requires_grad=True) for k in train_names}
    stats = {}
    # ---- PFN ----
    feats = torch.from_numpy(nn_ref.pfn_decorate_np(voxels, num_points, coors, d.voxel_size, d.pc_range, d.with_distance)).to(dtype)
    y = feats @ t["pfn/dense/kernel"]                                  # [P, T, C]
    yn, m, v = _bn_train(y, t["pfn/bn/gamma"], t["pfn/bn/beta"], (0, 1))
    stats["pfn/bn"] = (m.detach().numpy(), v.detach().numpy())
    if record is not None:
        with torch.no_grad():
            best, bi = yn.max(dim=1)
            bi = torch.where(bi >= torch.from_numpy(num_points.astype(np.int64))[:, None], torch.full_like(bi, -1), bi)
            record["pfn"] = torch.where(best > 0, bi, torch.full_like(bi, -2)).numpy().astype(np.int32)
    if forced is not None:
        arg = torch.from_numpy(np.ascontiguousarray(forced["pfn"]).astype(np.int64))          # [P, C]
        pad_row = torch.from_numpy(np.minimum(num_points.astype(np.int64), yn.shape[1] - 1))[:, None].expand_as(arg)
        idx = torch.where(arg >= 0, arg, pad_row)
        f = yn.gather(1, idx[:, None, :]).squeeze(1) * (arg != -2).to(dtype)
    else:
        f = torch.relu(yn).amax(dim=1)                                  # [P, C]
    if margins is not None:
        with torch.no_grad():
            top = torch.topk(torch.relu(yn), 2, dim=1).values if yn.shape[1] > 1 else None
            gap = (top[:, 0] - top[:, 1]) if top is not None else None
            win = yn.amax(dim=1)                                           # the winning row's pre-ReLU value
            margins["pfn/bn"] = float(min(win.abs().min(), gap[gap > 0].min() if gap is not None and bool((gap > 0).any()) else 1.0))
    idx = torch.from_numpy((coors[:, 0].astype(np.int64) * d.ny + coors[:, 2]) * d.nx + coors[:, 3])
    canvas = torch.zeros(B * d.ny * d.nx, f.shape[1], dtype=dtype).index_add(0, idx, f).reshape(B, d.ny, d.nx, -1)
    x = canvas.permute(0, 3, 1, 2)
    # ---- RPN ----
    ups = []
    for b in range(3):
        for j in range(d.layer_nums[b] + 1):
            pre = f"rpn/block{b + 1}/{j}"
            stride = d.layer_strides[b] if j == 0 else 1
            dw = t[pre + "/depthwise_kernel"].permute(2, 3, 0, 1)       # [3,3,Cin,1] -> [Cin,1,3,3]
            pw = t[pre + "/pointwise_kernel"].permute(3, 2, 0, 1)       # [1,1,Cin,Cout] -> [Cout,Cin,1,1]
            x = Fn.conv2d(x, dw, stride=stride, padding=1, groups=x.shape[1])
            x = Fn.conv2d(x, pw)
            x, m, v = _bn_train(x, t[pre + "/bn/gamma"], t[pre + "/bn/beta"], (0, 2, 3))
            stats[pre + "/bn"] = (m.detach().numpy(), v.detach().numpy())
            if margins is not None:
                margins[pre + "/bn"] = float(x.detach().abs().min())
                margins["#near-kink"] = margins.get("#near-kink", 0) + int((x.detach().abs() < 1e-6).sum())
                margins["#pre-relu"] = margins.get("#pre-relu", 0) + x.numel()
            if record is not None:
                record[pre + "/bn"] = (x.detach() > 0).permute(0, 2, 3, 1).reshape(x.shape[0], -1, x.shape[1]).numpy()
            if forced is not None:
                m = torch.from_numpy(np.ascontiguousarray(forced[pre + "/bn"])).reshape(x.shape[0], x.shape[2], x.shape[3], x.shape[1])
                x = x * m.permute(0, 3, 1, 2).to(dtype)
            else:
                x = torch.relu(x)
        pre = f"rpn/deconv{b + 1}"
        k = d.upsample_strides[b]
        in_h, in_w = x.shape[2], x.shape[3]
        u = Fn.conv_transpose2d(x, t[pre + "/kernel"].permute(3, 2, 0, 1), stride=k)   # [k,k,Cout,Cin] -> [Cin,Cout,k,k]
        u, m, v = _bn_train(u, t[pre + "/bn/gamma"], t[pre + "/bn/beta"], (0, 2, 3))
        stats[pre + "/bn"] = (m.detach().numpy(), v.detach().numpy())
        if margins is not None:
            margins[pre + "/bn"] = float(u.detach().abs().min())
            margins["#near-kink"] = margins.get("#near-kink", 0) + int((u.detach().abs() < 1e-6).sum())
            margins["#pre-relu"] = margins.get("#pre-relu", 0) + u.numel()
        if record is not None:
            record[pre + "/bn"] = (u.detach() > 0).reshape(u.shape[0], u.shape[1], in_h, k, in_w, k).permute(0, 2, 4, 3, 5, 1) \
                .reshape(u.shape[0], in_h * in_w, k, k, u.shape[1]).numpy()
        if forced is not None:       # [B, input pixels, ti, tj, C] -> [B, C, in_h * k + ti, in_w * k + tj]
            m = torch.from_numpy(np.ascontiguousarray(forced[pre + "/bn"])).reshape(u.shape[0], in_h, in_w, k, k, u.shape[1])
            ups.append(u * m.permute(0, 5, 1, 3, 2, 4).reshape(u.shape).to(dtype))
        else:
            ups.append(torch.relu(u))
    cat = torch.cat(ups, dim=1)

    def head(name):
        return (Fn.conv2d(cat, t[name + "/kernel"].permute(3, 2, 0, 1), bias=t[name + "/bias"])).permute(0, 2, 3, 1)

    use_dir = bool(d.config["model"]["second"]["use_direction_classifier"])      # model/voxelnet.py:690
    box, cls = head("rpn/conv_box"), head("rpn/conv_cls")
    dr = head("rpn/conv_dir_cls") if use_dir else None
    lt = loss_ref.loss_tensors(d.config["model"]["second"], box, cls, dr, labels, reg_targets, anchors, dtype)
    lt["loss"].backward()
    vals = {k: float(vv.detach()) for k, vv in lt.items() if k != "num_positives"}
    vals["num_positives"] = int(lt["num_positives"])
    grads = {k: (tt.grad.numpy() if tt.grad is not None else np.zeros(tt.shape, np.float32)) for k, tt in t.items()}
    preds = {"box_preds": box.detach().numpy(), "cls_preds": cls.detach().numpy()}
    if dr is not None:
        preds["dir_cls_preds"] = dr.detach().numpy()
    return vals, grads, stats, preds
