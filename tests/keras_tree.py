"""The variable tree Keras 2.3 / TF 2.2 `save_weights` writes for the reference's VoxelNet (train.py:62-113,
model/pointpillars.py:97-115, model/voxelnet.py:573-691): TEST INFRASTRUCTURE shared by tests/test_host.py (a dict
tree), tests/test_h5lite.py and tools/gen_golden_h5.py (a real HDF5 file written by h5py).

net.layers = [loss, PillarFeatureNet, PointPillarsScatter, RPN]; the anonymous layers inside are numbered by creation
order (separable_conv2d_7, batch_normalization_12, ...); per top-level layer Keras lists the trainable variables in
layer order, then the moving statistics (`_legacy_weights`)."""

LAYER_NAMES = ["weighted_smooth_l1_localization_loss", "pillar_feature_net", "point_pillars_scatter", "rpn"]


def keras_variables(d):
    """{top-level layer: [(keras variable name, package tensor name or None)]} in the order Keras writes them."""
    pn = "voxel_net/pillar_feature_net/sequential/"
    pfn = [(pn + "dense/kernel:0", "pfn/dense/kernel"), (pn + "batch/gamma:0", "pfn/bn/gamma"),
           (pn + "batch/beta:0", "pfn/bn/beta"), (pn + "batch/moving_mean:0", "pfn/bn/moving_mean"),
           (pn + "batch/moving_variance:0", "pfn/bn/moving_variance")]
    train, moving = [], []
    n_sep = n_bn = n_dec = 0

    def suffix(n):
        return "" if n == 0 else f"_{n}"
    for b in range(3):
        for j in range(d.layer_nums[b] + 1):
            base, ours = f"voxel_net/rpn/block{b + 1}/", f"rpn/block{b + 1}/{j}"
            sep, bn = f"separable_conv2d{suffix(n_sep)}", f"batch_normalization{suffix(n_bn)}"
            n_sep += 1
            n_bn += 1
            train += [(base + sep + "/depthwise_kernel:0", ours + "/depthwise_kernel"),
                      (base + sep + "/pointwise_kernel:0", ours + "/pointwise_kernel"),
                      (base + bn + "/gamma:0", ours + "/bn/gamma"), (base + bn + "/beta:0", ours + "/bn/beta")]
            moving += [(base + bn + "/moving_mean:0", ours + "/bn/moving_mean"),
                       (base + bn + "/moving_variance:0", ours + "/bn/moving_variance")]
        # the reference builds deconv{b} right after block{b}; Keras lists RPN.layers in attribute order
        base, ours = f"voxel_net/rpn/deconv{b + 1}/", f"rpn/deconv{b + 1}"
        dec, bn = f"conv2d_transpose{suffix(n_dec)}", f"batch_normalization{suffix(n_bn)}"
        n_dec += 1
        n_bn += 1
        train += [(base + dec + "/kernel:0", ours + "/kernel"), (base + bn + "/gamma:0", ours + "/bn/gamma"),
                  (base + bn + "/beta:0", ours + "/bn/beta")]
        moving += [(base + bn + "/moving_mean:0", ours + "/bn/moving_mean"),
                   (base + bn + "/moving_variance:0", ours + "/bn/moving_variance")]
    heads = ["conv_box", "conv_cls"] + (["conv_dir_cls"] if getattr(d, "use_direction_classifier", True) else [])
    for hname in heads:
        train += [(f"voxel_net/rpn/{hname}/kernel:0", f"rpn/{hname}/kernel"), (f"voxel_net/rpn/{hname}/bias:0", f"rpn/{hname}/bias")]
    return {"weighted_smooth_l1_localization_loss": [("code_weights:0", None)],      # model/voxelnet.py:404
            "pillar_feature_net": pfn, "point_pillars_scatter": [], "rpn": train + moving}
