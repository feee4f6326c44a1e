"""CPU restatement of the reference's network: PillarFeatureNet, PointPillarsScatter, RPN.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: these are
TensorFlow-2.2 Keras layers in the reference (model/pointpillars.py:65-341,
model/voxelnet.py:517-717); TensorFlow is not installable here and the
reference ships no golden vectors, so the layer semantics are restated from
the Keras definitions and cross-checked between two independent
implementations in this file: `*_torch` (torch-CPU fp32, fast) and `*_np`
(plain numpy loops/einsum, tiny shapes only).

Weights are a dict name -> float32 ndarray in KERAS layouts:
  pfn/dense/kernel [Fa,C]                  pfn/bn/{gamma,beta,moving_mean,moving_variance} [C]
  rpn/block{b}/{j}/depthwise_kernel [3,3,Cin,1]   rpn/block{b}/{j}/pointwise_kernel [1,1,Cin,Cout]
  rpn/block{b}/{j}/bn/{gamma,beta,moving_mean,moving_variance}
  rpn/deconv{b}/kernel [k,k,Cout,Cin]      rpn/deconv{b}/bn/...
  rpn/conv_box/{kernel [1,1,384,2*7], bias}   rpn/conv_cls/...   rpn/conv_dir_cls/...
b = 1..3, j = 0..layer_nums[b-1] (j = 0 is the strided ZeroPadding2D(1)+valid layer).
"""
import numpy as np

F32 = np.float32
BN_EPS = 1e-3  # explicit at model/pointpillars.py:109; Keras default for the RPN BNs (model/voxelnet.py:577...)


# --------------------------------------------------------------------------
# a5  PillarFeatureNet  (model/pointpillars.py:128-225)
# --------------------------------------------------------------------------
def pfn_decorate_np(voxels, num_points, coors, voxel_size, pc_range, with_distance=False):
    """Feature decoration + pad mask, fp32 (model/pointpillars.py:143-203).
    voxels [P,T,F], num_points [P], coors [P,4] (b,z,y,x) -> [P,T,F+5] (+1 with_distance:
    tf.norm of the raw xyz appended last, model/pointpillars.py:185-188)."""
    voxels = voxels.astype(F32)
    vx, vy = F32(voxel_size[0]), F32(voxel_size[1])
    # Python-float64 arithmetic, then used as an f32 constant (pointpillars.py:121-124)
    x_off = F32(voxel_size[0] / 2 + pc_range[0])
    y_off = F32(voxel_size[1] / 2 + pc_range[1])
    mean = voxels[:, :, :3].sum(axis=1, keepdims=True, dtype=F32) / num_points.astype(F32).reshape(-1, 1, 1)
    f_cluster = voxels[:, :, :3] - mean
    cx = coors[:, 3].astype(F32)[:, None] * vx + x_off
    cy = coors[:, 2].astype(F32)[:, None] * vy + y_off
    f_center = np.stack([voxels[:, :, 0] - cx, voxels[:, :, 1] - cy], axis=-1)
    parts = [voxels, f_cluster, f_center]
    if with_distance:
        parts.append(np.sqrt((voxels[:, :, :3] * voxels[:, :, :3]).sum(axis=2, keepdims=True, dtype=F32)).astype(F32))
    feats = np.concatenate(parts, axis=-1)
    T = voxels.shape[1]
    mask = (num_points.astype(np.int32)[:, None] > np.arange(T, dtype=np.int32)[None, :])
    return feats * mask[..., None].astype(F32)


def _bn_np(x, p):
    """Keras BatchNormalization inference: x*inv + (beta - mean*inv), inv = gamma*rsqrt(var+eps)."""
    inv = (p["gamma"] / np.sqrt(p["moving_variance"] + F32(BN_EPS))).astype(F32)
    return x * inv + (p["beta"] - p["moving_mean"] * inv)


def _bn_params(w, prefix):
    return {k: w[f"{prefix}/{k}"] for k in ("gamma", "beta", "moving_mean", "moving_variance")}


def pfn_np(voxels, num_points, coors, w, voxel_size, pc_range, with_distance=False):
    """Dense(no bias) -> BN -> ReLU -> max over ALL T rows, incl. zero-padded
    ones (model/pointpillars.py:211-219).  Returns [P,C] fp32."""
    feats = pfn_decorate_np(voxels, num_points, coors, voxel_size, pc_range, with_distance)
    y = feats @ w["pfn/dense/kernel"]
    y = _bn_np(y, _bn_params(w, "pfn/bn"))
    y = np.maximum(y, F32(0))
    return y.max(axis=1)


# --------------------------------------------------------------------------
# a6  PointPillarsScatter  (model/pointpillars.py:285-341)
# --------------------------------------------------------------------------
def scatter_np(features, coors, batch_size, ny, nx):
    """Per frame: canvas[ny*nx, C] zeros; index y*nx+x; duplicate indices ADD
    (tf.scatter_nd).  Returns NHWC [B,ny,nx,C] (the RPN transposes the
    reference's NCHW output to NHWC straight away, model/voxelnet.py:697)."""
    C = features.shape[1]
    out = np.zeros((batch_size, ny * nx, C), dtype=F32)
    for b in range(batch_size):
        m = coors[:, 0] == b
        idx = coors[m, 2].astype(np.int64) * nx + coors[m, 3].astype(np.int64)
        np.add.at(out[b], idx, features[m])
    return out.reshape(batch_size, ny, nx, C)


# --------------------------------------------------------------------------
# a7  RPN  (model/voxelnet.py:517-717) -- torch-CPU
# --------------------------------------------------------------------------
def rpn_torch(canvas_nhwc, w, rpn_cfg, num_threads=None):
    """canvas [B,H,W,C] fp32 numpy -> dict of NHWC numpy arrays.
    rpn_cfg: layer_nums, layer_strides, num_filters, upsample_strides."""
    import torch
    import torch.nn.functional as Fn
    if num_threads:
        torch.set_num_threads(num_threads)

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def bn(x, prefix):
        p = _bn_params(w, prefix)
        return Fn.batch_norm(x, t(p["moving_mean"]), t(p["moving_variance"]), t(p["gamma"]), t(p["beta"]),
                             training=False, eps=BN_EPS)

    def sep(x, prefix, stride):
        dw = t(w[f"{prefix}/depthwise_kernel"]).permute(2, 3, 0, 1)       # [3,3,Cin,1] -> [Cin,1,3,3]
        pw = t(w[f"{prefix}/pointwise_kernel"]).permute(3, 2, 0, 1)       # [1,1,Cin,Cout] -> [Cout,Cin,1,1]
        x = Fn.conv2d(x, dw.contiguous(), stride=stride, padding=1, groups=x.shape[1])
        x = Fn.conv2d(x, pw.contiguous())
        return torch.relu(bn(x, prefix + "/bn"))

    def deconv(x, prefix, k):
        wt = t(w[f"{prefix}/kernel"]).permute(3, 2, 0, 1)                 # [k,k,Cout,Cin] -> [Cin,Cout,k,k]
        x = Fn.conv_transpose2d(x, wt.contiguous(), stride=k)
        return torch.relu(bn(x, prefix + "/bn"))

    def head(x, prefix):
        k = t(w[f"{prefix}/kernel"]).permute(3, 2, 0, 1)
        return Fn.conv2d(x, k.contiguous(), bias=t(w[f"{prefix}/bias"]))

    with torch.no_grad():
        x = t(canvas_nhwc).permute(0, 3, 1, 2).contiguous()
        ups = []
        for b in range(3):
            x = sep(x, f"rpn/block{b + 1}/0", rpn_cfg["layer_strides"][b])
            for j in range(rpn_cfg["layer_nums"][b]):
                x = sep(x, f"rpn/block{b + 1}/{j + 1}", 1)
            ups.append(deconv(x, f"rpn/deconv{b + 1}", rpn_cfg["upsample_strides"][b]))
        cat = torch.cat(ups, dim=1)
        out = {
            "box_preds": head(cat, "rpn/conv_box"),
            "cls_preds": head(cat, "rpn/conv_cls"),
        }
        if "rpn/conv_dir_cls/kernel" in w:     # model/voxelnet.py:690,714: only with use_direction_classifier
            out["dir_cls_preds"] = head(cat, "rpn/conv_dir_cls")
        return {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in out.items()}


# --------------------------------------------------------------------------
# a7  RPN -- independent plain-numpy restatement (tiny shapes; cross-check)
# --------------------------------------------------------------------------
def _sep_np(x, dwk, pwk, stride):
    """ZeroPadding2D(1)+valid (stride s) == 'same' for s=1 with k=3: symmetric pad 1.
    x [B,H,W,Cin]; dwk [3,3,Cin,1]; pwk [1,1,Cin,Cout]."""
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    Ho = (H + 2 - 3) // stride + 1
    Wo = (W + 2 - 3) // stride + 1
    dw = np.zeros((B, Ho, Wo, C), dtype=F32)
    for i in range(3):
        for j in range(3):
            dw += xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :] * dwk[i, j, :, 0]
    return dw @ pwk[0, 0]


def _deconv_np(x, k):
    """Conv2DTranspose kernel==stride, valid: out[y*s+i, x*s+j, co] = sum_ci in[y,x,ci]*K[i,j,co,ci]."""
    B, H, W, Ci = x.shape
    s, _, Co, _ = k.shape
    y = np.einsum("bhwc,ijoc->bhiwjo", x, k)
    return y.reshape(B, H * s, W * s, Co).astype(F32)


def rpn_np(canvas_nhwc, w, rpn_cfg):
    x = canvas_nhwc.astype(F32)
    ups = []
    for b in range(3):
        names = [f"rpn/block{b + 1}/{j}" for j in range(rpn_cfg["layer_nums"][b] + 1)]
        for j, p in enumerate(names):
            s = rpn_cfg["layer_strides"][b] if j == 0 else 1
            x = _sep_np(x, w[p + "/depthwise_kernel"], w[p + "/pointwise_kernel"], s)
            x = np.maximum(_bn_np(x, _bn_params(w, p + "/bn")), F32(0))
        u = _deconv_np(x, w[f"rpn/deconv{b + 1}/kernel"])
        ups.append(np.maximum(_bn_np(u, _bn_params(w, f"rpn/deconv{b + 1}/bn")), F32(0)))
    cat = np.concatenate(ups, axis=-1)
    out = {}
    for name, key in (("box_preds", "rpn/conv_box"), ("cls_preds", "rpn/conv_cls"), ("dir_cls_preds", "rpn/conv_dir_cls")):
        if key + "/kernel" not in w:
            continue
        out[name] = (cat @ w[key + "/kernel"][0, 0] + w[key + "/bias"]).astype(F32)
    return out


# --------------------------------------------------------------------------
# a13  VoxelNet.call eval branch  (model/voxelnet.py:850-916, 1055-1056)
# --------------------------------------------------------------------------
def voxelnet_forward(voxels, num_points, coors, batch_size, w, model_cfg, num_threads=None):
    """model_cfg: voxel_size, pc_range, grid (nx,ny,nz), rpn{...}.  Returns
    (preds_dict, canvas_nhwc, pillar_features)."""
    feats = pfn_np(voxels, num_points, coors, w, model_cfg["voxel_size"], model_cfg["pc_range"],
                   bool(model_cfg.get("with_distance", False)))
    nx, ny = int(model_cfg["grid"][0]), int(model_cfg["grid"][1])
    canvas = scatter_np(feats, coors, batch_size, ny, nx)
    preds = rpn_torch(canvas, w, model_cfg["rpn"], num_threads=num_threads)
    return preds, canvas, feats
