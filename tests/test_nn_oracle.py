"""Cross-check of the two independent restatements of the network (oracle/nn_ref.py).

TensorFlow cannot run here, so the Keras layer semantics are unpinned by the
reference; the torch-CPU and plain-numpy restatements must at least agree with
each other, and hand-computed micro cases pin the layout conventions
(Conv2DTranspose [k,k,Cout,Cin], SeparableConv2D, NHWC).  CPU only.
"""
import numpy as np

from oracle import nn_ref


def _tiny(pp):
    d = pp.config.Derived(pp.config.tiny_config(batch_size=2))
    w = pp.weights.init_weights(d, seed=3)
    return d, w


def test_rpn_torch_vs_numpy(pp):
    d, w = _tiny(pp)
    rng = np.random.default_rng(0)
    canvas = rng.standard_normal((2, d.ny, d.nx, d.pfn_filters)).astype(np.float32)
    canvas[rng.random(canvas.shape[:3]) < 0.5] = 0
    a = nn_ref.rpn_torch(canvas, w, d.rpn_dict())
    b = nn_ref.rpn_np(canvas, w, d.rpn_dict())
    for k in a:
        assert a[k].shape == b[k].shape == (2, d.head_h, d.head_w, a[k].shape[-1])
        np.testing.assert_allclose(a[k], b[k], rtol=2e-5, atol=2e-5)


def test_deconv_layout_micro():
    # one input pixel, k=2: out[i, j, co] = sum_ci x[ci] * K[i, j, co, ci]
    x = np.array([[[[1.0, 2.0]]]], dtype=np.float32)           # [1,1,1,2]
    k = np.arange(2 * 2 * 3 * 2, dtype=np.float32).reshape(2, 2, 3, 2)
    y = nn_ref._deconv_np(x, k)
    assert y.shape == (1, 2, 2, 3)
    for i in range(2):
        for j in range(2):
            np.testing.assert_allclose(y[0, i, j], k[i, j] @ x[0, 0, 0])


def test_separable_stride2_padding_micro():
    # ZeroPadding2D(1) + valid stride 2 on a 4x4 ramp, identity pointwise, centre-tap depthwise
    x = np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1)
    dw = np.zeros((3, 3, 1, 1), np.float32)
    dw[1, 1, 0, 0] = 1
    pw = np.ones((1, 1, 1, 1), np.float32)
    y = nn_ref._sep_np(x, dw, pw, 2)
    assert y.shape == (1, 2, 2, 1)
    # output (oy, ox) reads padded (2oy+1, 2ox+1) = unpadded (2oy, 2ox)
    np.testing.assert_array_equal(y[0, :, :, 0], x[0, ::2, ::2, 0])
    dw[:] = 0
    dw[0, 0, 0, 0] = 1   # top-left tap -> unpadded (2oy-1, 2ox-1), zero on the border
    y = nn_ref._sep_np(x, dw, pw, 2)
    np.testing.assert_array_equal(y[0, :, :, 0], np.array([[0, 0], [0, 5]], np.float32))


def test_pfn_pad_constant_and_mask(pp):
    d, w = _tiny(pp)
    T, F = d.max_points, d.num_point_features
    vox = np.zeros((2, T, F), np.float32)
    vox[0, :3] = [[0.1, 0.1, 0.0], [0.12, 0.11, 0.2], [0.13, 0.1, -0.1]]
    vox[1, :] = np.random.default_rng(1).uniform(0.2, 0.25, (T, F))
    num = np.array([3, T], np.int32)
    coors = np.array([[0, 0, 9, 1], [0, 1, 10, 2]], np.int32)
    f = nn_ref.pfn_np(vox, num, coors, w, d.voxel_size, d.pc_range)
    inv = w["pfn/bn/gamma"] / np.sqrt(w["pfn/bn/moving_variance"] + np.float32(1e-3))
    pad = np.maximum(w["pfn/bn/beta"] - w["pfn/bn/moving_mean"] * inv, 0)
    assert (f[0] >= pad - 1e-6).all(), "a pillar with < T points maxes with the padded-row constant"
    assert (pad > 0).any(), "weights must make the pad constant matter (SURVEY fact 4)"
    assert not (f[1] >= pad - 1e-6).all(), "a full pillar does not see the pad constant"


def test_scatter_adds_duplicates():
    feats = np.array([[1.0, 2.0], [10.0, 20.0], [5.0, 5.0]], np.float32)
    coors = np.array([[0, 0, 1, 2], [0, 1, 1, 2], [1, 0, 0, 0]], np.int32)
    c = nn_ref.scatter_np(feats, coors, 2, 2, 3)
    np.testing.assert_array_equal(c[0, 1, 2], [11.0, 22.0])
    np.testing.assert_array_equal(c[1, 0, 0], [5.0, 5.0])
    assert c.sum() == 43.0
