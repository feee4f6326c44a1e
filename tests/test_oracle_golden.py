"""The oracle (oracle/) against the reference's own outputs (tests/golden/ref_*.npz).

Pins the CPU restatement: every function here must reproduce, bit for bit, what
the reference's numpy code produced in tools/gen_golden.py.  CPU only.
"""
import hashlib

import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle, ref_numpy as rn

VOX_CASES = ["a2k", "a16k", "brk", "t100", "kitti5k", "tiny"]


def _derived(pp, case):
    c = pp.config
    if case == "kitti5k":
        return c.Derived(c.kitti_shaped_config())
    if case == "tiny":
        return c.Derived(c.tiny_config())
    return c.Derived(c.pedestrian_d435i_config())


@pytest.mark.parametrize("case", VOX_CASES)
def test_c_voxeliser_matches_reference(pp, case):
    g = load_golden("ref_voxel.npz")
    d = _derived(pp, case)
    T, MV = (int(v) for v in g[case + "_params"])
    v, c, n = c_oracle.points_to_voxel(g[case + "_points"], d.voxel_size, d.pc_range, T, MV)
    assert np.array_equal(c, g[case + "_coors"])
    assert np.array_equal(n, g[case + "_num"])
    assert np.array_equal(v, g[case + "_voxels"])


@pytest.mark.parametrize("case", ["a2k", "brk", "tiny"])
def test_numpy_voxeliser_matches_reference(pp, case):
    g = load_golden("ref_voxel.npz")
    d = _derived(pp, case)
    T, MV = (int(v) for v in g[case + "_params"])
    v, c, n = rn.points_to_voxel(g[case + "_points"], d.voxel_size, d.pc_range, T, True, MV)
    assert np.array_equal(c, g[case + "_coors"])
    assert np.array_equal(n, g[case + "_num"])
    assert np.array_equal(v, g[case + "_voxels"])


def test_voxeliser_z_cells_and_break_are_exercised():
    g = load_golden("ref_voxel.npz")
    assert (g["a2k_coors"][:, 0] == 1).any(), "fixture must populate the second z cell (SURVEY fact 10)"
    yx = g["a2k_coors"][:, 1] * 1000 + g["a2k_coors"][:, 2]
    assert len(np.unique(yx)) < len(yx), "fixture must contain two pillars sharing (y, x)"
    assert g["brk_coors"].shape[0] == 300 and g["a2k_coors"].shape[0] > 300, "break case must hit max_voxels"
    assert g["a2k_num"].max() == 50, "fixture must overflow max_points in a cell"


def test_anchors_match_reference(pp):
    g = load_golden("ref_anchors.npz")
    c = pp.config
    for name, cfg in (("A", c.pedestrian_d435i_config()), ("T", c.tiny_config())):
        d = c.Derived(cfg)
        a = rn.generate_anchors(d.feature_map_size, d.anchor_cfg).reshape(-1, 7)
        assert a.dtype == np.float32 and np.array_equal(a, g[name + "_anchors"])
        bv = rn.rbbox2d_to_near_bbox(a[:, [0, 1, 3, 4, 6]])
        assert np.array_equal(bv, g[name + "_bv"])
    d = c.Derived(c.kitti_shaped_config())
    a = rn.generate_anchors(d.feature_map_size, d.anchor_cfg).reshape(-1, 7)
    assert list(a.shape) == list(g["K_shape"])
    assert hashlib.sha256(a.tobytes()).digest() == g["K_sha256"].tobytes()
    assert np.array_equal(a[::997], g["K_rows"])
    bv = rn.rbbox2d_to_near_bbox(a[:, [0, 1, 3, 4, 6]])
    assert hashlib.sha256(np.ascontiguousarray(bv).tobytes()).digest() == g["K_bv_sha256"].tobytes()


@pytest.mark.parametrize("case", ["a2k", "a16k"])
def test_anchor_mask_matches_reference(pp, case):
    gv, gm, ga = load_golden("ref_voxel.npz"), load_golden("ref_mask.npz"), load_golden("ref_anchors.npz")
    d = pp.config.Derived(pp.config.pedestrian_d435i_config())
    coors = gv[case + "_coors"]
    m = rn.anchors_mask(coors, ga["A_anchors"], d.voxel_size, d.pc_range, d.anchor_area_threshold)
    assert np.array_equal(m, gm[case + "_mask"])
    cells = rn.anchor_cells(ga["A_bv"], d.voxel_size, d.pc_range, d.grid)
    m2 = c_oracle.anchor_mask(coors, d.ny, d.nx, cells, float(d.anchor_area_threshold))
    assert np.array_equal(m2, gm[case + "_mask"])


def test_decode_corners_standup_match_reference():
    g = load_golden("ref_decode.npz")
    dec = rn.second_box_decode(g["enc"], g["anchors"])
    assert dec.dtype == np.float32 and np.array_equal(dec, g["decoded"])
    bev = dec[..., [0, 1, 3, 4, 6]]
    corners = rn.center_to_corner_box2d(bev[:, :2], bev[:, 2:4], bev[:, 4])
    assert np.array_equal(corners, g["corners"])
    assert np.array_equal(rn.corner_to_standup(corners), g["standup"])


def test_nms_host_sweep_matches_reference():
    g = load_golden("ref_nms_post.npz")
    for n in (1, 37, 64, 100, 130):
        keep = rn.nms_postprocess(g[f"n{n}_mask"], n)
        assert np.array_equal(np.array(keep, dtype=np.int32), g[f"n{n}_keep"])


def test_lidar_to_camera_matches_reference():
    g = load_golden("ref_camera.npz")
    for s in ("", "2"):
        cam = rn.box_lidar_to_camera(g["boxes"], g["rect" + s], g["trv" + s])
        assert cam.dtype == np.float64 and np.array_equal(cam, g["cam" + s])


def test_nms_gpu_matches_the_reference_kernel():
    """ref_cuda_kernels.npz: the reference's nms_gpu -> nms_kernel -> nms_postprocess (libraries/eval_helper_functions.py:
    494-598) run as they are by the CUDA-model emulator of tools/ref_shim.py (round 4: the kernel's block / thread
    indexing, shared staging and bit masks are no longer unpinned): box counts on both sides of the 64-thread blocks,
    three thresholds, no pair within 1e-4 of a threshold."""
    g = load_golden("ref_cuda_kernels.npz")
    ncases = 0
    for thr in (0.5, 0.1, 0.7):
        for n in (1, 2, 63, 64, 65, 100, 129, 200):
            dets, keep = g[f"nms_t{thr}_n{n}_dets"], g[f"nms_t{thr}_n{n}_keep"]
            assert list(rn.nms_gpu(dets, thr)) == list(keep), (thr, n)
            order = np.argsort(-dets[:, 4], kind="stable")              # (scores are a permutation: no ties)
            k_c = c_oracle.nms_sorted(np.ascontiguousarray(dets[order]), thr)
            assert [int(order[i]) for i in k_c] == list(keep), (thr, n)
            ncases += 1
    assert ncases == 24 and len(g["nms_t0.1_n200_keep"]) < 40 < len(g["nms_t0.7_n200_keep"])


def test_nms_function_matches_the_reference():
    """nms() as the reference wrote it (libraries/eval_helper_functions.py:463-492: np.argpartition top pre_max_size,
    nms_gpu on the emulated kernel, post_max_size, None for nothing kept): the numpy-1.19 list-index idiom it uses runs
    through an ndarray subclass with the old meaning (tools/gen_golden_kernels.py)."""
    g = load_golden("ref_cuda_kernels.npz")
    for k in range(int(g["nmsfn_count"])):
        pre, post, thr = g[f"nmsfn_{k}_args"]
        got = rn.nms(g[f"nmsfn_{k}_boxes"], g[f"nmsfn_{k}_scores"], None if pre < 0 else int(pre), int(post), float(thr))
        if bool(g[f"nmsfn_{k}_none"]):
            assert got is None
        else:
            assert np.array_equal(np.asarray(got, dtype=np.int64), g[f"nmsfn_{k}_keep"]), k
    assert int(g["nmsfn_count"]) == 6


def test_nms_kernel_restatements_agree():
    """The numpy and C restatements of nms_kernel are written independently and must agree with each other (beside the
    fixture above)."""
    rng = np.random.default_rng(5)
    for n in (1, 5, 64, 100, 131):
        c = rng.uniform(0, 6, (n, 2)).astype(np.float32)
        wh = rng.uniform(0.3, 1.2, (n, 2)).astype(np.float32)
        boxes = np.concatenate([c - wh / 2, c + wh / 2, np.sort(rng.random((n, 1)).astype(np.float32), axis=0)[::-1]], axis=1)
        k_np = rn.nms_postprocess(rn.nms_mask(boxes, 0.5), n)
        k_c = c_oracle.nms_sorted(boxes, 0.5)
        assert list(k_c) == list(k_np)
    # the +1 convention: two 0.6 m boxes 0.5 m apart overlap "more than 0.5" 
    b = np.array([[0, 0, .6, .6, .9], [.5, 0, 1.1, .6, .8], [3, 3, 3.6, 3.6, .7]], dtype=np.float32)
    assert list(c_oracle.nms_sorted(b, 0.5)) == [0, 2]


def test_nms_iou_formula_matches_reference_device_function():
    """iou_device (libraries/eval_helper_functions.py:553-564) executed as plain Python by the generator:
    float32 there (numpy promotion) vs float64 under numba, hence a 1e-6 bar instead of equality."""
    g = load_golden("ref_iou_device.npz")
    b = g["boxes"]
    for ri, i in enumerate(g["rows"]):
        for ci, j in enumerate(g["cols"]):
            assert abs(rn.nms_iou(b[i], b[j]) - g["iou"][ri, ci]) < 1e-6
    assert (g["iou"] > 0.5).sum() > 20 and (g["iou"] == 0.0).sum() > 20


def test_oracle_voxelise_forward_index_matches_reference():
    """reverse_index=False (load_data.py:643-692): the restatement against the reference's own output."""
    g, gf = load_golden("ref_voxel.npz"), load_golden("ref_voxel_fwd.npz")
    import pp_amd
    d = pp_amd.config.Derived(pp_amd.config.pedestrian_d435i_config())
    for case in ("a2k", "brk"):
        T, MV = (int(v) for v in gf[case + "_params"])
        v, c, n = rn.points_to_voxel(g[case + "_points"], d.voxel_size, d.pc_range, T, False, MV)
        assert np.array_equal(c, gf[case + "_coors"]) and np.array_equal(n, gf[case + "_num"])
        assert np.array_equal(v, gf[case + "_voxels"])
