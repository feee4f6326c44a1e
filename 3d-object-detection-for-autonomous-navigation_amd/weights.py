"""Weight containers in the reference's Keras layouts.

The released checkpoint (out/model_345/.../model_weights_48.h5) is not in the
reference tree (.MISSING_LARGE_BLOBS) and h5py is absent, so parity is
established with seeded synthetic weights that have NON-trivial BatchNorm
statistics (the PFN pad constant ReLU(beta - gamma*mean/sqrt(var+eps))
must matter).  Names / layouts (Keras):
  pfn/dense/kernel [Fa,C]                           model/pointpillars.py:103
  pfn/bn/{gamma,beta,moving_mean,moving_variance}   model/pointpillars.py:109
  rpn/block{b}/{j}/depthwise_kernel [3,3,Cin,1]     model/voxelnet.py:576,584,...
  rpn/block{b}/{j}/pointwise_kernel [1,1,Cin,Cout]
  rpn/block{b}/{j}/bn/*
  rpn/deconv{b}/kernel [k,k,Cout,Cin], rpn/deconv{b}/bn/*      model/voxelnet.py:591-599
  rpn/conv_box|conv_cls|conv_dir_cls/{kernel [1,1,Cin,Cout], bias}   model/voxelnet.py:684-691
"""
import numpy as np

BN_KEYS = ("gamma", "beta", "moving_mean", "moving_variance")


def layer_table(d):
    """Ordered list of (kind, name, shape-info) describing the network of config `d`."""
    layers = []
    cin = d.pfn_filters
    for b in range(3):
        cout = d.num_filters[b]
        for j in range(d.layer_nums[b] + 1):
            stride = d.layer_strides[b] if j == 0 else 1
            layers.append(("sep", f"rpn/block{b + 1}/{j}", {"cin": cin, "cout": cout, "stride": stride}))
            cin = cout
        layers.append(("deconv", f"rpn/deconv{b + 1}",
                       {"cin": cout, "cout": d.num_upsample_filters[b], "k": d.upsample_strides[b]}))
    cc = d.concat_channels
    layers.append(("head", "rpn/conv_box", {"cin": cc, "cout": d.num_anchor_per_loc * 7}))
    layers.append(("head", "rpn/conv_cls", {"cin": cc, "cout": d.num_anchor_per_loc * d.num_class}))
    if getattr(d, "use_direction_classifier", True):   # model/voxelnet.py:690
        layers.append(("head", "rpn/conv_dir_cls", {"cin": cc, "cout": d.num_anchor_per_loc * 2}))
    return layers


def _he_uniform(rng, shape, fan_in):
    lim = np.sqrt(6.0 / fan_in)
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _unit_gain_uniform(rng, shape, fan_in):
    """Variance-preserving uniform init.  The reference initialises with he_uniform and
    relies on TRAINED BatchNorm statistics to keep activations O(1); with random BN
    statistics a gain of 2 per conv would grow the signal by ~4x per separable layer
    (x 2^30 over the backbone), so the synthetic RPN kernels use gain 1 instead.  Head
    outputs then have the O(1) magnitude of a trained detector, which is what the
    1e-4 parity tolerance is meant for."""
    lim = np.sqrt(3.0 / fan_in)
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _bn(rng, c, prefix, out):
    out[prefix + "/gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    out[prefix + "/beta"] = rng.uniform(-0.5, 0.5, c).astype(np.float32)
    out[prefix + "/moving_mean"] = rng.uniform(-0.5, 0.5, c).astype(np.float32)
    out[prefix + "/moving_variance"] = rng.uniform(0.5, 2.0, c).astype(np.float32)


def init_weights(d, seed=7):
    """Seeded synthetic weights (SURVEY section 8c/8d distributions)."""
    rng = np.random.default_rng(seed)
    w = {}
    w["pfn/dense/kernel"] = _he_uniform(rng, (d.pfn_in, d.pfn_filters), d.pfn_in)
    _bn(rng, d.pfn_filters, "pfn/bn", w)
    for kind, name, s in layer_table(d):
        if kind == "sep":
            w[name + "/depthwise_kernel"] = _unit_gain_uniform(rng, (3, 3, s["cin"], 1), 9)
            w[name + "/pointwise_kernel"] = _unit_gain_uniform(rng, (1, 1, s["cin"], s["cout"]), s["cin"])
            _bn(rng, s["cout"], name + "/bn", w)
        elif kind == "deconv":
            k = s["k"]
            w[name + "/kernel"] = _unit_gain_uniform(rng, (k, k, s["cout"], s["cin"]), s["cin"])
            _bn(rng, s["cout"], name + "/bn", w)
        else:
            w[name + "/kernel"] = _unit_gain_uniform(rng, (1, 1, s["cin"], s["cout"]), s["cin"])
            w[name + "/bias"] = rng.uniform(-0.1, 0.1, s["cout"]).astype(np.float32)
    return w


def expected_shapes(d):
    shapes = {"pfn/dense/kernel": (d.pfn_in, d.pfn_filters)}
    for k in BN_KEYS:
        shapes["pfn/bn/" + k] = (d.pfn_filters,)
    for kind, name, s in layer_table(d):
        if kind == "sep":
            shapes[name + "/depthwise_kernel"] = (3, 3, s["cin"], 1)
            shapes[name + "/pointwise_kernel"] = (1, 1, s["cin"], s["cout"])
            for k in BN_KEYS:
                shapes[name + "/bn/" + k] = (s["cout"],)
        elif kind == "deconv":
            shapes[name + "/kernel"] = (s["k"], s["k"], s["cout"], s["cin"])
            for k in BN_KEYS:
                shapes[name + "/bn/" + k] = (s["cout"],)
        else:
            shapes[name + "/kernel"] = (1, 1, s["cin"], s["cout"])
            shapes[name + "/bias"] = (s["cout"],)
    return shapes


def check_weights(d, w):
    """Raises ValueError on a missing / mis-shaped tensor (the reference's
    net.load_weights raises on a layout mismatch too, train.py:731-734)."""
    for name, shp in expected_shapes(d).items():
        if name not in w:
            raise ValueError(f"missing weight tensor {name!r}")
        if tuple(w[name].shape) != tuple(shp):
            raise ValueError(f"weight {name!r}: shape {tuple(w[name].shape)} != expected {tuple(shp)}")


def save_npz(path, w):
    np.savez(path, **{k.replace("/", "."): v for k, v in w.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k.replace(".", "/"): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}


# ---------------------------------------------------------------------------------------------------------
# Keras .h5 checkpoints (net.save_weights / net.load_weights, train.py:407,436,731-734)
# ---------------------------------------------------------------------------------------------------------
_KINDS = ("depthwise_kernel", "pointwise_kernel", "gamma", "beta", "moving_mean", "moving_variance", "kernel", "bias")


def map_keras_weight_names(keras_names, d):
    """Maps the variable names of a Keras checkpoint of the reference's VoxelNet to this package's tensor names.

    The reference's model tree (train.py:62-113: net.layers = [loss, PillarFeatureNet, PointPillarsScatter, RPN];
    PillarFeatureNet.layers[0] = Sequential(Dense, BatchNormalization "batch", ReLU); RPN.layers = [block1, deconv1,
    block2, deconv2, block3, deconv3, conv_box, conv_cls, conv_dir_cls], the six Sequentials named "block1" ...
    "deconv3", model/voxelnet.py:573-660) fixes where a variable lives; Keras numbers the anonymous layers inside
    (separable_conv2d_7, batch_normalization_12, ...) by creation order.  The mapping therefore goes by the NAMED
    path component (block{b} / deconv{b} / conv_box / conv_cls / conv_dir_cls, anything else with a Dense kernel or
    the "batch" BatchNorm = the PFN), the variable kind (last path component) and the order of appearance of that
    kind inside the component -- `layer.weights` lists a Sequential's variables in layer order (trainable first, then
    the moving statistics, each in layer order).  Returns {package name: keras name}; shapes are checked by the caller.
    """
    counters = {}
    out = {}
    for kn in keras_names:
        parts = kn.split(":")[0].split("/")
        kind = parts[-1]
        if kind not in _KINDS:
            continue
        comp = None
        for tok in parts[:-1]:
            if tok.startswith(("block", "deconv")) and tok[-1].isdigit() or tok in ("conv_box", "conv_cls", "conv_dir_cls"):
                comp = tok
        if comp is None:
            comp = "pfn"
        idx = counters.get((comp, kind), 0)
        counters[(comp, kind)] = idx + 1
        if comp == "pfn":
            name = "pfn/dense/kernel" if kind == "kernel" else f"pfn/bn/{kind}"
        elif comp.startswith("block"):
            name = f"rpn/{comp}/{idx}/{kind}" if kind.endswith("_kernel") else f"rpn/{comp}/{idx}/bn/{kind}"
        elif comp.startswith("deconv"):
            name = f"rpn/{comp}/kernel" if kind == "kernel" else f"rpn/{comp}/bn/{kind}"
        else:
            name = f"rpn/{comp}/{kind}"
        if name in out:
            raise ValueError(f"two checkpoint variables map to {name!r}: {out[name]!r} and {kn!r}")
        out[name] = kn
    missing = [n for n in expected_shapes(d) if n not in out]
    if missing:
        raise ValueError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
    return out


def _h5_names(attrs, name):
    """A name list attribute the way Keras reads it back (`load_attributes_from_hdf5_group`): `name`, or -- when the list
    was too large for one object-header attribute -- its pieces `name0`, `name1`, ..."""
    def _s(x):
        return x.decode() if isinstance(x, bytes) else str(x)
    if name in attrs:
        return [_s(x) for x in np.asarray(attrs[name]).ravel().tolist()] if np.asarray(attrs[name]).size else []
    out, i = [], 0
    while f"{name}{i}" in attrs:
        out += [_s(x) for x in np.asarray(attrs[f"{name}{i}"]).ravel().tolist()]
        i += 1
    return out


def from_keras_h5(group, d):
    """`group`: an open HDF5 file of a Keras `save_weights` checkpoint (root attribute `layer_names`, one group per
    top-level layer with the attribute `weight_names` and one dataset per variable) through h5py.File or this package's
    h5lite.File -- or any object with the same mapping interface (a unit test uses plain dicts).  Returns this
    package's weight dict, shapes verified."""
    names, data = [], {}
    for ln in _h5_names(group.attrs, "layer_names"):
        g = group[ln]
        for wn in _h5_names(g.attrs, "weight_names"):
            names.append(wn)
            data[wn] = np.asarray(g[wn], dtype=np.float32)
    mapping = map_keras_weight_names(names, d)
    w = {name: np.ascontiguousarray(data[kn]) for name, kn in mapping.items() if name in expected_shapes(d)}
    check_weights(d, w)
    return w


def load_keras_h5(path, d):
    """Reads the reference's checkpoint file `model_weights_<epoch>.h5` (net.save_weights, train.py:407,436; read back
    by net.load_weights, train.py:731-734) into this package's weight dict.  h5py is used when the interpreter has it;
    otherwise the built-in reader (h5lite.py: the part of the HDF5 format such files use, checked against files written
    by the real library) -- the importer does not depend on a package the deployment box may lack."""
    try:
        import h5py
        opener = lambda p: h5py.File(p, "r")       # noqa: E731
    except ImportError:
        from . import h5lite
        opener = h5lite.File
    with opener(path) as f:
        return from_keras_h5(f, d)


def load_any(path, d):
    """.npz of this package, or a Keras .h5 checkpoint (recognised by the file signature, not the extension)."""
    from . import h5lite
    return load_keras_h5(path, d) if h5lite.is_hdf5(path) else load_npz(path)
