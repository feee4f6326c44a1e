"""Parity of the HIP path (through the C-ABI) against the oracle and the golden fixtures.

All tests here need a real MI355X (`-m gpu`).  Tolerances: integer / index work
is bit-exact; floating point within 1e-4 (north_star), most stages far tighter.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle, nn_ref, ref_numpy as rn
import util_ref

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _cfg(pp, case):
    c = pp.config
    if case == "kitti5k":
        return c.kitti_shaped_config()
    if case == "tiny":
        return c.tiny_config()
    return c.pedestrian_d435i_config()


@pytest.fixture(scope="module")
def engines(pp, hip_lib):
    cache = {}

    def get(name, cfg, max_batch=1, nmax=32768, weights_seed=None):
        key = (name, max_batch, nmax)
        if key not in cache:
            eng = pp.Engine(cfg, max_batch=max_batch, max_points_per_frame=nmax)
            if weights_seed is not None:
                eng.load_weights(pp.weights.init_weights(eng.d, seed=weights_seed))
            cache[key] = eng
        return cache[key]

    yield get
    for e in cache.values():
        e.close()


# ------------------------------------------------------------------ a1
@pytest.mark.parametrize("case", ["a2k", "a16k", "brk", "t100", "kitti5k", "tiny"])
def test_voxelise_matches_reference_golden(pp, engines, case):
    g = load_golden("ref_voxel.npz")
    T, MV = (int(v) for v in g[case + "_params"])
    cfg = _cfg(pp, case)
    cfg["model"]["second"]["voxel_generator"].update(max_number_of_points_per_voxel=T, max_number_of_voxels=MV)
    eng = engines(f"vox-{case}", cfg)
    v, c, n = eng.points_to_voxel(g[case + "_points"])
    assert np.array_equal(c, g[case + "_coors"]), "pillar coordinates / order must be bit-exact"
    assert np.array_equal(n, g[case + "_num"])
    assert np.array_equal(v, g[case + "_voxels"])


def test_points_to_voxel_reference_signature(pp, hip_lib):
    g = load_golden("ref_voxel.npz")
    d = pp.config.Derived(pp.config.pedestrian_d435i_config())
    v, c, n = pp.points_to_voxel(g["a2k_points"], d.voxel_size, d.pc_range, 50, True, 12000)
    assert np.array_equal(c, g["a2k_coors"]) and np.array_equal(n, g["a2k_num"]) and np.array_equal(v, g["a2k_voxels"])


def test_voxelise_edge_cases(pp, engines):
    eng = engines("vox-a2k", _cfg(pp, "a2k"))
    d = eng.d
    for pts in (np.zeros((0, 3), np.float32),                       # empty
                np.full((7, 3), 100.0, np.float32),                 # everything out of range
                np.array([[1.0, 0.0, 0.0]], np.float32),            # one point
                np.tile(np.array([[2.0, 0.5, 0.1]], np.float32), (300, 1)),  # 300 duplicates > T
                np.array([[np.nan, 0, 0], [1, 0, 0], [np.inf, 0, 0], [1, -np.inf, 0]], np.float32)):
        v, c, n = eng.points_to_voxel(pts)
        finite = pts[np.isfinite(pts).all(axis=1)] if pts.size else pts
        ve, ce, ne = c_oracle.points_to_voxel(finite, d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
        assert np.array_equal(c, ce) and np.array_equal(n, ne) and np.array_equal(v, ve)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_voxelise_16k_vs_oracle(pp, engines, seed):
    eng = engines("vox-a2k", _cfg(pp, "a2k"))
    d = eng.d
    pts = pp.synth.d435i_cloud(100 + seed)
    v, c, n = eng.points_to_voxel(pts)
    ve, ce, ne = c_oracle.points_to_voxel(pts, d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
    assert np.array_equal(c, ce) and np.array_equal(n, ne) and np.array_equal(v, ve)


def test_voxelise_max_voxels_break_at_scale(pp, engines):
    cfg = pp.config.pedestrian_d435i_config(max_points=10, max_voxels=1000)
    eng = engines("vox-break1000", cfg)
    d = eng.d
    pts = pp.synth.d435i_cloud(7)
    v, c, n = eng.points_to_voxel(pts)
    ve, ce, ne = c_oracle.points_to_voxel(pts, d.voxel_size, d.pc_range, 10, 1000)
    assert v.shape[0] == 1000
    assert np.array_equal(c, ce) and np.array_equal(n, ne) and np.array_equal(v, ve)


# ------------------------------------------------------------------ a4
@pytest.mark.parametrize("case", ["a2k", "a16k"])
def test_anchor_mask_matches_reference_golden(pp, engines, case):
    gv, gm = load_golden("ref_voxel.npz"), load_golden("ref_mask.npz")
    eng = engines("vox-a2k", _cfg(pp, "a2k"))
    c = gv[case + "_coors"]
    c4 = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], axis=1)
    m = eng.anchor_mask(c4, 1)[0].astype(bool)
    assert np.array_equal(m, gm[case + "_mask"])


# ------------------------------------------------------------------ a5-a7, a13
@pytest.mark.parametrize("name,nframes,npts", [("tiny", 2, 600), ("A", 2, 16384)])
def test_forward_matches_oracle(pp, engines, name, nframes, npts):
    cfg = pp.config.tiny_config(nframes) if name == "tiny" else pp.config.pedestrian_d435i_config(nframes)
    eng = engines(f"net-{name}", cfg, max_batch=nframes, weights_seed=7)
    d = eng.d
    w = pp.weights.init_weights(d, seed=7)
    if name == "tiny":
        rng = np.random.default_rng(9)
        frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (npts, 3)).astype(np.float32) for _ in range(nframes)]
    else:
        frames = [pp.synth.d435i_cloud(20 + i, npts) for i in range(nframes)]
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    ex = ref["example"]
    out = eng.forward_voxels(ex[0], ex[1], ex[2], nframes, want_features=True, want_canvas=True)
    np.testing.assert_allclose(out["pillar_features"], ref["features"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["canvas"], ref["canvas"], rtol=1e-5, atol=1e-5)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        assert out[k].shape == ref["preds"][k].shape
        np.testing.assert_allclose(out[k], ref["preds"][k], rtol=0, atol=TOL)
    # shared (y, x) across z-cells must have been summed (tf.scatter_nd semantics)
    yx = ex[2][:, 0] * 10**6 + ex[2][:, 2] * 1000 + ex[2][:, 3]
    if name == "A":
        assert len(np.unique(yx)) < len(yx)


def test_small_map_split_k_kernel_ragged_tiles(pp, engines):
    """Few frames on a small map run the split-K separable kernel (k_sep_k4); a 28x20 grid makes the
    pixel counts of block1 / block2 (560 / 140 per frame) end inside a 32-pixel tile, and block3's odd
    width (7) takes the generic kernel.  Full-width layers (64/128/256 channels), against the oracle."""
    import copy
    B = 2
    cfg = copy.deepcopy(pp.config.pedestrian_d435i_config(B))
    cfg["eval_input_reader"]["feature_map_size"] = [1, 20, 28]
    s = cfg["model"]["second"]
    s["voxel_generator"].update(point_cloud_range=[0, -0.8, -3.0, 2.24, 0.8, 3.0], max_number_of_voxels=560)
    s["target_assigner"]["anchor_generators"]["anchor_generator_stride"].update(offsets=[0.08, -0.8, -1.465])
    eng = engines("net-ragged", cfg, max_batch=B, weights_seed=11)
    d = eng.d
    assert (d.nx, d.ny) == (28, 20)
    tags = eng.layer_tags()
    assert any(t.startswith("k_sep_k4") for t in tags) and any(t.startswith("k_deconv_k4") for t in tags), tags
    w = pp.weights.init_weights(d, seed=11)
    rng = np.random.default_rng(21)
    frames = [rng.uniform([0, -0.8, -3], [2.24, 0.8, 3], (n, 3)).astype(np.float32) for n in (3000, 1700)]
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    ex = ref["example"]
    out = eng.forward_voxels(ex[0], ex[1], ex[2], B)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        assert out[k].shape == ref["preds"][k].shape
        np.testing.assert_allclose(out[k], ref["preds"][k], rtol=0, atol=TOL)


# ------------------------------------------------------------------ a8-a12
def _assert_dets(pp_dicts, ref_dicts):
    assert len(pp_dicts) == len(ref_dicts)
    for a, b in zip(pp_dicts, ref_dicts):
        if b["scores"] is None:
            assert a["scores"] is None and a["box3d_lidar"] is None and a["bbox"] is None
            continue
        assert a["scores"] is not None
        assert a["scores"].shape == b["scores"].shape
        np.testing.assert_allclose(a["scores"], b["scores"], rtol=0, atol=TOL)
        np.testing.assert_allclose(a["box3d_lidar"], b["box3d_lidar"], rtol=0, atol=TOL)
        np.testing.assert_allclose(a["box3d_camera"], b["box3d_camera"], rtol=0, atol=TOL)
        assert a["box3d_camera"].dtype == np.float64 and a["box3d_lidar"].dtype == np.float32
        assert np.array_equal(a["label_preds"], b["label_preds"])
        assert np.array_equal(a["bbox"], b["bbox"])


def test_predict_matches_oracle_on_oracle_preds(pp, hip_lib):
    """VoxelNet.predict surface on head maps computed by the oracle (isolates a8-a12)."""
    cfg = pp.config.pedestrian_d435i_config(2)
    net = pp.VoxelNet(cfg, None, training=False, max_batch=2)
    d = net.d
    w = pp.weights.init_weights(d, seed=7)
    net.load_weights(w)
    frames = [pp.synth.d435i_cloud(40 + i) for i in range(2)]
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    got = net.predict(ref["example"], ref["preds"])
    assert all(r["scores"] is not None and len(r["scores"]) >= 2 for r in ref["dets"]), "fixture must produce detections"
    _assert_dets(got, ref["dets"])
    # empty mask -> the reference's all-None dict
    ex = list(ref["example"])
    ex[7] = np.zeros_like(ex[7])
    got = net.predict(tuple(ex), ref["preds"])
    assert all(g["scores"] is None and g["batch_idx"] == i for i, g in enumerate(got))
    net.engine.close()


def test_predict_nms_conventions(pp, engines):
    """Hand-built head maps: +1 IoU in metre space, strict >, pre/post caps, direction flip."""
    eng = engines("net-A1", pp.config.pedestrian_d435i_config(1), max_batch=1)
    d = eng.d
    A = d.num_anchors
    box = np.zeros((1, d.head_h, d.head_w, 14), np.float32)
    cls = np.full((1, d.head_h, d.head_w, 2), -9.0, np.float32)
    dr = np.zeros((1, d.head_h, d.head_w, 4), np.float32)
    mask = np.zeros((1, A), np.uint8)   # only the hand-placed anchors are candidates (no score ties)
    # three anchors: two 0.4 m apart (suppressed by the +1 convention), one 3 m away
    for (y, x, r, logit, dirbin) in ((10, 10, 0, 3.0, 1), (10, 15, 0, 2.0, 0), (40, 50, 1, 1.0, 1)):
        cls[0, y, x, r] = logit
        dr[0, y, x, 2 * r + dirbin] = 1.0
        mask[0, (y * d.head_w + x) * 2 + r] = 1
    rect, trv, _ = pp.synth.default_calib()
    dets, n = eng.predict(box, cls, dr, mask, rect[None], trv[None])
    ex = (None, None, None, rect[None], trv[None], None, eng.anchors[None], mask, np.array([0]), None)
    ref = rn.predict(ex, {"box_preds": box, "cls_preds": cls, "dir_cls_preds": dr}, d.nms_dict())[0]
    assert n[0] == len(ref["scores"])
    np.testing.assert_allclose(dets[0]["score"][:n[0]], ref["scores"], rtol=1e-6)
    np.testing.assert_allclose(dets[0]["box3d_lidar"][:n[0]], ref["box3d_lidar"], rtol=1e-6, atol=1e-6)
    a0 = (10 * d.head_w + 10) * 2
    assert dets[0]["anchor_index"][0] == a0 and dets[0]["dir_label"][0] == 1
    kept = set(int(v) for v in dets[0]["anchor_index"][:n[0]])
    assert (10 * d.head_w + 15) * 2 not in kept, "0.4 m apart must be suppressed by the +1 IoU"
    assert (40 * d.head_w + 50) * 2 + 1 in kept


@pytest.mark.parametrize("score_thr,pre_max,post_max", [(0.55, 100, 50), (0.0, 30, 7), (0.3, 1000, 5)])
def test_predict_thresholds_and_caps(pp, engines, score_thr, pre_max, post_max):
    """nms_score_threshold > 0, nms_pre_max_size < 100 and small post caps (config-driven branches
    of model/voxelnet.py:1193-1203 and libraries/eval_helper_functions.py:470-486)."""
    cfg = pp.config.pedestrian_d435i_config(1)
    s = cfg["model"]["second"]
    s["nms_score_threshold"], s["nms_pre_max_size"], s["nms_post_max_size"] = score_thr, pre_max, post_max
    eng = engines(f"net-A1-thr{score_thr}-{pre_max}-{post_max}", cfg, max_batch=1)
    d = eng.d
    rng = np.random.default_rng(int(score_thr * 100) + pre_max)
    box = (rng.standard_normal((1, d.head_h, d.head_w, 14)) * 0.3).astype(np.float32)
    cls = (rng.standard_normal((1, d.head_h, d.head_w, 2)) * 0.8).astype(np.float32)
    dr = rng.standard_normal((1, d.head_h, d.head_w, 4)).astype(np.float32)
    mask = (rng.random((1, d.num_anchors)) < 0.6).astype(np.uint8)
    rect, trv, _ = pp.synth.default_calib()
    dets, n = eng.predict(box, cls, dr, mask, rect[None], trv[None])
    ex = (None, None, None, rect[None], trv[None], None, eng.anchors[None], mask, np.array([0]), None)
    ref = rn.predict(ex, {"box_preds": box, "cls_preds": cls, "dir_cls_preds": dr}, d.nms_dict())[0]
    assert ref["scores"] is not None and n[0] == len(ref["scores"]) <= post_max
    np.testing.assert_allclose(dets[0]["score"][:n[0]], ref["scores"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dets[0]["box3d_lidar"][:n[0]], ref["box3d_lidar"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dets[0]["box3d_camera"][:n[0]], ref["box3d_camera"], rtol=1e-5, atol=1e-5)
    if score_thr > 0:
        assert (dets[0]["score"][:n[0]] >= score_thr).all()


def test_config0_variant_t100_12000(pp, engines):
    """BASELINE.json configs[0] as stated: 12000-pillar cap, 100 points per pillar."""
    cfg = pp.config.pedestrian_d435i_config(1, max_points=100, max_voxels=12000)
    eng = engines("net-A1-T100", cfg, max_batch=1, weights_seed=7)
    d = eng.d
    w = pp.weights.init_weights(d, seed=7)
    frames = [pp.synth.d435i_cloud(77)]
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    dets, n = eng.detect(frames, rect[None], trv[None])
    im = eng.intermediates()
    P = ref["frames"][0]["coordinates"].shape[0]
    assert im["n_pillars"][0] == P and np.array_equal(im["num_points"][0, :P], ref["frames"][0]["num_points"])
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(im[k], ref["preds"][k], rtol=0, atol=TOL)
    _assert_dets([pp.VoxelNet._to_dict(dets[0], int(n[0]), 0)], ref["dets"])


# ------------------------------------------------------------------ fused path, end to end
@pytest.mark.parametrize("name", ["tiny", "A"])
def test_fused_detect_matches_oracle_end_to_end(pp, engines, name):
    B = 3
    cfg = pp.config.tiny_config(B) if name == "tiny" else pp.config.pedestrian_d435i_config(B)
    eng = engines(f"net-{name}", cfg, max_batch=B, weights_seed=7)
    d = eng.d
    w = pp.weights.init_weights(d, seed=7)
    if name == "tiny":
        rng = np.random.default_rng(19)
        frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (700, 0, 333)]
    else:
        frames = [pp.synth.d435i_cloud(60, 16384), pp.synth.d435i_cloud(61, 9000), pp.synth.d435i_cloud(62, 16384)]
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    dets, n = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    im = eng.intermediates(canvas=True)
    for b in range(B):
        fr = ref["frames"][b]
        P = fr["coordinates"].shape[0]
        assert im["n_pillars"][b] == P
        assert np.array_equal(im["coors"][b, :P], fr["coordinates"])
        assert np.array_equal(im["num_points"][b, :P], fr["num_points"])
        assert np.array_equal(im["anchors_mask"][b].astype(bool), fr["anchors_mask"])
    np.testing.assert_allclose(im["canvas"], ref["canvas"], rtol=1e-5, atol=1e-5)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(im[k], ref["preds"][k], rtol=0, atol=TOL)
    from importlib import import_module
    to_dict = pp.VoxelNet._to_dict
    got = [to_dict(dets[b], int(n[b]), b) for b in range(B)]
    _assert_dets(got, ref["dets"])


def test_keras_h5_checkpoint_into_the_engine(pp, hip_lib):
    """f4 end to end (train.py:731-734, net.load_weights(model_weights_<epoch>.h5)): the checkpoint file written by the
    real HDF5 library in Keras's layout (tests/golden/keras_ckpt_tiny.h5, tools/gen_golden_h5.py) goes straight into
    VoxelNet.load_weights; the detections match the oracle run on the weights the file was made from."""
    import os
    B = 2
    cfg = pp.config.tiny_config(B)
    net = pp.VoxelNet(cfg, None, training=False, max_batch=B)
    try:
        net.load_weights(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keras_ckpt_tiny.h5"))
        w = pp.weights.init_weights(net.d, seed=29)
        rng = np.random.default_rng(23)
        frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (900, 450)]
        rect, trv, p2 = pp.synth.default_calib()
        ref = util_ref.oracle_detect(net.d, w, frames, rect, trv, p2)
        got = net.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
        _assert_dets(got, ref["dets"])
        assert any(r["scores"] is not None for r in ref["dets"])
    finally:
        net.engine.close()


# ------------------------------------------------------------------ size-independent properties at full batch
def test_full_batch_properties(pp, engines):
    B = 64
    eng = engines("net-A64", pp.config.pedestrian_d435i_config(B), max_batch=B, weights_seed=7)
    frames = [pp.synth.d435i_cloud(200 + i) for i in range(B)]
    dets1, n1 = eng.detect(frames)
    im1 = eng.intermediates()
    dets2, n2 = eng.detect(frames)
    assert np.array_equal(n1, n2) and dets1.tobytes() == dets2.tobytes(), "bit-reproducible"
    # a frame's result does not depend on its batch neighbours or its slot
    perm = np.random.default_rng(0).permutation(B)
    dets3, n3 = eng.detect([frames[i] for i in perm])
    for slot, i in enumerate(perm):
        assert n3[slot] == n1[i]
        assert dets3[slot, :n3[slot]].tobytes() == dets1[i, :n1[i]].tobytes()
    for b in range(B):
        k = int(n1[b])
        assert 0 <= k <= eng.d.nms_post_max_size
        s = dets1[b]["score"][:k]
        assert (np.diff(s) <= 0).all(), "keep order is descending score"
        assert ((s > 0) & (s < 1)).all()
        # NMS is idempotent: survivors do not suppress each other (stand-up AABB, `+1` IoU <= 0.5 for every kept pair;
        # oracle arithmetic: libraries/eval_helper_functions.py:553-564 on load_data.py:1525-1593 corners)
        if k > 1:
            bx = dets1[b]["box3d_lidar"][:k]
            aabb = rn.corner_to_standup(rn.center_to_corner_box2d(bx[:, :2], bx[:, 3:5], bx[:, 6]))
            for i in range(k):
                for j in range(i + 1, k):
                    assert rn.nms_iou(aabb[i], aabb[j]) <= 0.5 + 1e-6, (b, i, j)
        P = im1["n_pillars"][b]
        assert 0 < P <= eng.d.max_voxels
        assert im1["num_points"][b, :P].min() >= 1 and im1["num_points"][b, :P].max() <= eng.d.max_points
        flat = (im1["coors"][b, :P, 0] * eng.d.ny + im1["coors"][b, :P, 1]) * eng.d.nx + im1["coors"][b, :P, 2]
        assert len(np.unique(flat)) == P, "one pillar per cell"
    # pillar point budget: sum(min(count, T)) over pillars == points kept by the oracle for a sampled frame
    d = eng.d
    ve, ce, ne = c_oracle.points_to_voxel(frames[5], d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
    assert np.array_equal(im1["coors"][5, :len(ce)], ce) and np.array_equal(im1["num_points"][5, :len(ne)], ne)


def test_kitti_shaped_forward(pp, engines):
    """cfg-K (496x432 BEV, F=4, C=64, T=100): voxelise + mask bit-exact, head maps within 1e-4."""
    B = 1
    eng = engines("net-K", pp.config.kitti_shaped_config(B), max_batch=B, weights_seed=5)
    d = eng.d
    w = pp.weights.init_weights(d, seed=5)
    frames = [pp.synth.kitti_cloud(3)]
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    dets, n = eng.detect(frames, rect[None], trv[None])
    # (this grid runs the sparse canvas: the PFN writes occupied cells only, block1.0 consults the cell map,
    # and the debug tap zero-fills the cells that were never written)
    im = eng.intermediates(canvas=True)
    np.testing.assert_allclose(im["canvas"], ref["canvas"], rtol=1e-5, atol=1e-5)
    assert (np.abs(im["canvas"]).sum(axis=-1) == 0).mean() > 0.9, "mostly empty grid"
    fr = ref["frames"][0]
    P = fr["coordinates"].shape[0]
    assert im["n_pillars"][0] == P and np.array_equal(im["coors"][0, :P], fr["coordinates"])
    assert np.array_equal(im["anchors_mask"][0].astype(bool), fr["anchors_mask"])
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(im[k], ref["preds"][k], rtol=0, atol=TOL)
    got = [pp.VoxelNet._to_dict(dets[0], int(n[0]), 0)]
    _assert_dets(got, ref["dets"])
    # the compat entry (padded voxels in, head maps out) on the same sparse-canvas engine: its PFN writes the
    # whole pseudo-image, block1.0 still goes through the cell map built from the given coordinates
    ex = ref["example"]
    out = eng.forward_voxels(ex[0], ex[1], ex[2], B, want_canvas=True)
    np.testing.assert_allclose(out["canvas"], ref["canvas"], rtol=1e-5, atol=1e-5)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(out[k], ref["preds"][k], rtol=0, atol=TOL)


def test_error_behaviour(pp, engines):
    eng = engines("net-tiny-err", pp.config.tiny_config(1), max_batch=1)
    with pytest.raises(RuntimeError, match="weights"):
        eng.detect([np.zeros((4, 3), np.float32)])
    with pytest.raises(ValueError):
        eng.points_to_voxel(np.zeros((4, 4), np.float32))
    with pytest.raises(RuntimeError, match="batch"):
        eng.detect([np.zeros((4, 3), np.float32)] * 2)
    bad = np.array([[0, 0, 99, 0]], np.int32)
    with pytest.raises(RuntimeError, match="outside"):
        eng.anchor_mask(bad, 1)
    w = pp.weights.init_weights(eng.d, seed=1)
    w["rpn/conv_box/kernel"] = w["rpn/conv_box/kernel"][..., :7]
    with pytest.raises(ValueError):
        eng.load_weights(w)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"PP_GEMM_PREC": "f32"}, {"PP_GEMM_PREC": "f32", "PP_SEP_KERNEL": "ws"},
                                 {"PP_PFN_KERNEL": "1"}, {"PP_NO_HEAD_FUSION": "1"},
                                 {"PP_NO_GRAPH": "1", "PP_SEP_NT": "64"}, {"PP_SEP_K4": "0", "PP_DECONV_K4": "0"}, {"PP_DENSE_CANVAS": "1"}])
def test_fallback_kernel_generations_stay_in_parity(hip_lib, env):
    """The earlier kernel generations are selectable at process start (fp32-MFMA instantiations, the
    producer/consumer GEMM, the first PFN); one child process per selection runs the whole path on two
    frames against the oracle (`__graft_entry__.smoke`), so the fallbacks do not rot."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child_env = dict(os.environ)
    child_env.update(env)
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root, env=child_env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke OK" in r.stdout


@pytest.mark.gpu
def test_voxelise_size_boundaries_of_the_lds_path(pp, engines):
    """Frames around the boundaries of k_voxel_frame's register / LDS path (<= 16384 points, 16 per thread)
    and of its global-memory fallback, with a small max_voxels so the `break` falls inside the frame, as a
    ragged batch through the fused path (per-frame coors / num_points from the intermediates)."""
    rng = np.random.default_rng(99)
    sizes = [1, 63, 1023, 1024, 1025, 4097, 16383, 16384, 16385, 20000]
    cfg = pp.config.pedestrian_d435i_config(len(sizes))
    cfg["model"]["second"]["voxel_generator"].update(max_number_of_points_per_voxel=7, max_number_of_voxels=900)
    eng = engines("vox-bounds", cfg, max_batch=len(sizes), nmax=20480, weights_seed=7)
    d = eng.d
    frames = []
    for n in sizes:
        pts = np.stack([rng.uniform(-0.3, 6.8, n), rng.uniform(-2.8, 2.8, n), rng.uniform(-3.2, 3.2, n)], 1).astype(np.float32)
        pts[rng.integers(0, n, max(1, n // 7))] = pts[rng.integers(0, n, max(1, n // 7))]   # duplicates
        frames.append(pts)
    rect, trv, _ = pp.synth.default_calib()
    eng.detect(frames, np.stack([rect] * len(sizes)), np.stack([trv] * len(sizes)))
    im = eng.intermediates()
    for b, pts in enumerate(frames):
        ve, ce, ne = c_oracle.points_to_voxel(pts, d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
        P = ce.shape[0]
        assert im["n_pillars"][b] == P, (sizes[b], im["n_pillars"][b], P)
        assert np.array_equal(im["coors"][b, :P], ce), sizes[b]
        assert np.array_equal(im["num_points"][b, :P], ne), sizes[b]
    # and one at a time through the padded compat entry point (both paths again, different batch shape)
    for n in (1024, 16384, 16385):
        pts = frames[sizes.index(n)]
        v, c, k = eng.points_to_voxel(pts)
        ve, ce, ne = c_oracle.points_to_voxel(pts, d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
        assert np.array_equal(c, ce) and np.array_equal(k, ne) and np.array_equal(v, ve)


@pytest.mark.gpu
def test_graph_cache_across_changing_batches(pp, engines):
    """pp_detect_async replays cached hipGraphs keyed by (batch, point-count bucket): batches of changing
    size and point counts (more distinct keys than cache slots, with repeats) must keep matching the plain
    launches of a profiling run, which never uses a graph."""
    cfg = pp.config.pedestrian_d435i_config(4)
    eng = engines("graph-cache", cfg, max_batch=4, nmax=20480, weights_seed=7)
    rect, trv, _ = pp.synth.default_calib()
    shapes = [(4, 3000), (2, 3000), (4, 9000), (1, 17000), (3, 12000), (4, 20000), (4, 3000), (2, 3000), (1, 17000)]
    for k, (nb, npts) in enumerate(shapes):
        frames = [pp.synth.d435i_cloud(500 + 10 * k + i, npts - 37 * i) for i in range(nb)]
        r, t = np.stack([rect] * nb), np.stack([trv] * nb)
        dets, n = eng.detect(frames, r, t)                      # graph path
        eng.set_profiling(True)
        dets2, n2 = eng.detect(frames, r, t)                    # plain launches
        eng.set_profiling(False)
        assert np.array_equal(n, n2), (k, n, n2)
        for b in range(nb):
            kk = int(n[b])
            assert np.array_equal(dets[b]["anchor_index"][:kk], dets2[b]["anchor_index"][:kk])
            assert np.array_equal(dets[b]["box3d_lidar"][:kk], dets2[b]["box3d_lidar"][:kk])


@pytest.mark.gpu
def test_bench_two_rank_path_rehearsal(hip_lib):
    """bench.py's N > 1 code path (rank env, frame shards, max-over-ranks timing, one JSON line on rank 0)
    with two ranks sharing this box's GPU and `gloo` carrying the scalar all-reduces: the throughput it
    prints means nothing, the contract fields do."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update({"PP_BENCH_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--batch", "8", "--no-latency-b1"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 8 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    assert d["cpu_baseline"] is None and d["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_one_rank_under_launcher_with_rccl(hip_lib):
    """bench.py as the driver starts it for N > 1 -- under torch.distributed.run with the `nccl` backend (= RCCL) --
    with ONE rank on this box's one GPU: the RCCL communicator is created before any engine exists, the device
    barrier and the MAX / SUM all-reduces around the timed region run on the GPU, and the training leg's gradient
    all-reduce goes through RCCL too.  (World size 1 moves no bytes between GPUs; what it covers is that the nccl
    branch initialises and runs beside the engine's own non-blocking streams.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("PP_BENCH_DIST_BACKEND", None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    # the gradient all-reduce of the training leg is skipped at world size 1 unless forced: here it goes through RCCL,
    # enqueued -- as on a multi-GPU node -- on the engine's own stream (torch ExternalStream), behind the backward pass
    env["PP_FORCE_ALLREDUCE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29519", "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--batch", "8", "--no-latency-b1", "--no-cfgk", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = lines[0]
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["collective_backend"].startswith("nccl"), d["config"]["collective_backend"]
    tr = d["detail"]["train"]
    assert "error" not in tr and tr["ms_per_step"] > 0 and np.isfinite(tr["last_loss"]), tr


def _random_config(rng, B):
    """A small reference-schema config drawn at random: grid, first stride, z cells, channel widths, layer counts,
    point features, pillar capacity."""
    import copy
    cfg = copy.deepcopy(pp_mod().config.pedestrian_d435i_config(B))
    s1 = int(rng.choice([1, 2]))
    nx, ny = 4 * s1 * int(rng.integers(3, 7)), 4 * s1 * int(rng.integers(2, 6))
    v = 0.08
    nz2 = bool(rng.integers(0, 2))
    zr = (-3.0, 3.0) if nz2 else (-3.0, 1.0)
    x0, y0 = 0.0, -ny * v / 2
    F = int(rng.choice([3, 4]))
    C = int(rng.choice([32, 64, 128]))
    filters = [int(rng.choice([32, 64])), int(rng.choice([32, 64])), int(rng.choice([64, 128]))]
    up = int(rng.choice([32, 64, 128]))
    cfg["eval_input_reader"].update(batch_size=B, feature_map_size=[1, ny // s1, nx // s1], num_point_features=F)
    s = cfg["model"]["second"]
    s["num_point_features"] = F
    s["voxel_generator"].update(point_cloud_range=[x0, y0, zr[0], x0 + nx * v, y0 + ny * v, zr[1]],
                                max_number_of_points_per_voxel=int(rng.choice([5, 12, 50])),
                                max_number_of_voxels=int(rng.choice([150, 2000])))
    s["voxel_feature_extractor"]["num_filters"] = C
    s["rpn"].update(layer_nums=[int(rng.integers(1, 3)) for _ in range(3)], layer_strides=[s1, 2, 2],
                    num_filters=filters, upsample_strides=[1, 2, 4], num_upsample_filters=[up] * 3)
    s["target_assigner"]["anchor_generators"]["anchor_generator_stride"].update(
        strides=[v * s1, v * s1, 0.0], offsets=[x0 + v * s1, y0, -1.465])
    return cfg


def pp_mod():
    import pp_amd
    return pp_amd


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106, 107, 108, 109, 110])
def test_random_small_configs_end_to_end(seed):
    """Configurations the shipped YAML does not use (first stride 2, one z cell, 4 point features, narrow layers,
    small pillar caps that trigger the max_voxels break): whole path against the oracle."""
    pp = pp_mod()
    rng = np.random.default_rng(seed)
    B = 2
    cfg = _random_config(rng, B)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=8192)
    d = eng.d
    w = pp.weights.init_weights(d, seed=seed)
    eng.load_weights(w)
    lo, hi = d.pc_range[:3], d.pc_range[3:]
    frames = []
    for n in (int(rng.integers(800, 4000)), int(rng.integers(1, 600))):
        xyz = rng.uniform(lo - 0.2, hi + 0.2, (n, 3))
        extra = rng.uniform(0, 1, (n, d.num_point_features - 3))
        frames.append(np.concatenate([xyz, extra], axis=1).astype(np.float32))
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    dets, n = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    im = eng.intermediates()
    for b in range(B):
        fr = ref["frames"][b]
        P = fr["coordinates"].shape[0]
        assert im["n_pillars"][b] == P and np.array_equal(im["coors"][b, :P], fr["coordinates"])
        assert np.array_equal(im["num_points"][b, :P], fr["num_points"])
        assert np.array_equal(im["anchors_mask"][b].astype(bool), fr["anchors_mask"])
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(im[k], ref["preds"][k], rtol=0, atol=TOL)
    got = [pp.VoxelNet._to_dict(dets[b], int(n[b]), b) for b in range(B)]
    _assert_dets(got, ref["dets"])
    eng.close()


@pytest.mark.gpu
def test_small_and_large_batch_kernels_agree(pp, engines):
    """A frame alone (split-K kernels k_sep_k4 / k_deconv_k4) and the same frame inside a 64-frame batch
    (persistent k_sep_u / k_deconv_u): the head maps agree far inside the parity tolerance and the detections
    are the same boxes."""
    big = engines("net-A64", pp.config.pedestrian_d435i_config(64), max_batch=64, weights_seed=7)
    one = engines("net-A1-w7", pp.config.pedestrian_d435i_config(1), max_batch=1, weights_seed=7)
    assert any(t.startswith("k_sep_k4") for t in one.layer_tags()) and any(t.startswith("k_sep_u") for t in big.layer_tags())
    frames = [pp.synth.d435i_cloud(400 + i) for i in range(64)]
    dets64, n64 = big.detect(frames)
    im64 = big.intermediates()
    for i in (0, 17, 63):
        d1, n1 = one.detect([frames[i]])
        im1 = one.intermediates()
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(im1[k][0], im64[k][i], rtol=2e-5, atol=2e-5)
        assert n1[0] == n64[i]
        assert np.array_equal(d1[0]["anchor_index"][:n1[0]], dets64[i]["anchor_index"][:n64[i]])
        np.testing.assert_allclose(d1[0]["score"][:n1[0]], dets64[i]["score"][:n64[i]], rtol=1e-5, atol=1e-6)
