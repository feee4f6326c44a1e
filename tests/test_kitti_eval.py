"""Row f2: rotated-IoU kernels and the KITTI-style AP evaluator.

Fixtures (tools/gen_golden_eval.py, produced by running the reference's second/utils/eval.py and the
device functions of second/core/non_max_suppression/nms_gpu.py as plain Python):
  tests/golden/ref_rotate_iou.npz   [N,K] rotated overlaps for criterion -1, 0, 1, 2
  tests/golden/ref_kitti_eval.npz   57 synthetic frames of gt / dt annos, per-frame overlaps, mAP arrays, reports
CPU tests pin the C oracle and the host bookkeeping (overlaps injected from the oracle); GPU tests run
the HIP kernels through the C-ABI.
"""
import numpy as np
import pytest

from conftest import load_golden

from oracle import c_oracle

GT_KEYS = ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y")
DT_KEYS = GT_KEYS + ("score",)


def _annos(g):
    n = int(g["nframes"])
    gts = [{k: g[f"gt_{i}_{k}"] for k in GT_KEYS} for i in range(n)]
    dts = [{k: g[f"dt_{i}_{k}"] for k in DT_KEYS} for i in range(n)]
    return gts, dts


def _oracle_bev(boxes, qboxes, criterion=-1):
    return c_oracle.rotate_iou_eval(boxes, qboxes, criterion)


def _oracle_d3(boxes, qboxes, criterion=-1):
    # second/utils/eval.py:132-163 on top of the oracle's BEV intersection
    boxes, qboxes = np.asarray(boxes, dtype=np.float64), np.asarray(qboxes, dtype=np.float64)
    rinc = c_oracle.rotate_iou_eval(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2)
    for i in range(boxes.shape[0]):
        for j in range(qboxes.shape[0]):
            if rinc[i, j] > 0:
                iw = min(boxes[i, 1], qboxes[j, 1]) - max(boxes[i, 1] - boxes[i, 4], qboxes[j, 1] - qboxes[j, 4])
                if iw > 0:
                    a1, a2 = boxes[i, 3] * boxes[i, 4] * boxes[i, 5], qboxes[j, 3] * qboxes[j, 4] * qboxes[j, 5]
                    inc = iw * rinc[i, j]
                    ua = {-1: a1 + a2 - inc, 0: a1, 1: a2}.get(criterion, 1.0)
                    rinc[i, j] = inc / ua
                else:
                    rinc[i, j] = 0.0
    return rinc


ORACLE_FNS = {1: _oracle_bev, 2: _oracle_d3}


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_rotate_iou_matches_reference():
    g = load_golden("ref_rotate_iou.npz")
    for crit in (-1, 0, 1, 2):
        out = c_oracle.rotate_iou_eval(g["boxes"], g["qboxes"], crit)
        assert np.array_equal(out, g[f"iou_c{crit}"]), crit
    assert (g["iou_c-1"] > 0).sum() > 200 and g["iou_c-1"].max() <= 1.0


def test_oracle_rotate_iou_matches_the_reference_kernel():
    """ref_cuda_kernels.npz: rotate_iou_gpu_eval -> rotate_iou_kernel_eval itself (nms_gpu.py:493-527, :618-653), run by the
    CUDA-model emulator of tools/ref_shim.py on 70 x 130 boxes (block edges in both grid dimensions)."""
    g = load_golden("ref_cuda_kernels.npz")
    for crit in (-1, 1):
        out = c_oracle.rotate_iou_eval(g["riou_boxes"], g["riou_qboxes"], crit)
        assert np.array_equal(out, g[f"riou_c{crit}"]), crit
    assert (g["riou_c-1"] > 0).sum() > 1000


def test_oracle_overlaps_match_reference_per_frame(pp):
    g = load_golden("ref_kitti_eval.npz")
    gts, dts = _annos(g)
    for metric in (1, 2):
        ov, _, tg, td = pp.kitti_eval.calculate_iou_partly(dts, gts, metric, num_parts=5, overlap_fns=ORACLE_FNS)
        assert len(ov) == len(gts)
        for i, o in enumerate(ov):
            ref = g[f"ov_m{metric}_{i}"]
            assert o.shape == ref.shape
            assert np.array_equal(o, ref), (metric, i)


def test_host_evaluator_matches_reference_reports(pp):
    g = load_golden("ref_kitti_eval.npz")
    gts, dts = _annos(g)
    ke = pp.kitti_eval
    text, mbbox, mbev, m3d, maos = ke.get_official_eval_result(gts, dts, ["Pedestrian"], compute_bbox=False,
                                                               overlap_fns=ORACLE_FNS)
    assert mbbox is None
    np.testing.assert_allclose(mbev, g["official_bev"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(m3d, g["official_3d"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(maos, g["official_aos"], rtol=0, atol=1e-9)
    assert text == str(g["official_text"])
    text2, b2, bev2, d32, aos2 = ke.get_official_eval_result(gts, dts, ["Pedestrian", "Cyclist"], difficultys=[0, 1, 2],
                                                             compute_bbox=True, overlap_fns=ORACLE_FNS)
    np.testing.assert_allclose(b2, g["official2_bbox"], rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(bev2, g["official2_bev"], rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(d32, g["official2_3d"], rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(aos2, g["official2_aos"], rtol=0, atol=1e-9, equal_nan=True)
    assert text2 == str(g["official2_text"])
    assert ke.get_coco_eval_result(gts, dts, ["Pedestrian"], overlap_fns=ORACLE_FNS) == str(g["coco_text"])


def test_few_frames_and_empty_frames(pp):
    """Fewer frames than num_parts (the reference raises there) and frames without boxes."""
    g = load_golden("ref_kitti_eval.npz")
    gts, dts = _annos(g)
    text, _, mbev, m3d, _ = pp.kitti_eval.get_official_eval_result(gts[:7], dts[:7], ["Pedestrian"], compute_bbox=False,
                                                                  overlap_fns=ORACLE_FNS)
    assert mbev.shape == (1, 3, 6) and m3d.shape == (1, 3, 6) and text.count("\n") == 6 * 4
    empty = {k: gts[0][k][:0] for k in GT_KEYS}
    empty_dt = {k: dts[0][k][:0] for k in DT_KEYS}
    ov, _, _, _ = pp.kitti_eval.calculate_iou_partly([empty_dt, dts[1]], [empty, gts[1]], 1, overlap_fns=ORACLE_FNS)
    assert ov[0].shape == (0, 0)


def test_image_box_overlap(pp):
    b = np.array([[0., 0., 10., 10.], [5., 5., 15., 15.], [20., 20., 30., 30.]])
    q = np.array([[0., 0., 10., 10.], [8., 8., 12., 12.]])
    o = pp.kitti_eval.image_box_overlap(b, q)
    assert o[0, 0] == 1.0 and o[2, 0] == 0.0 and abs(o[0, 1] - 4.0 / (100 + 16 - 4)) < 1e-15
    assert abs(pp.kitti_eval.image_box_overlap(b, q, 0)[1, 1] - 16.0 / 100.0) < 1e-15
    assert abs(pp.kitti_eval.image_box_overlap(b, q, 1)[1, 1] - 1.0) < 1e-15


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_rotate_iou_matches_reference(pp, hip_lib):
    g = load_golden("ref_rotate_iou.npz")
    for crit in (-1, 0, 1, 2):
        out = pp.kitti_eval.rotate_iou_eval(g["boxes"], g["qboxes"], crit)
        assert out.dtype == np.float32 and out.shape == g[f"iou_c{crit}"].shape
        assert np.array_equal(out, g[f"iou_c{crit}"]), (crit, np.abs(out - g[f"iou_c{crit}"]).max())


@pytest.mark.gpu
def test_hip_rotate_iou_matches_the_reference_kernel(pp, hip_lib):
    g = load_golden("ref_cuda_kernels.npz")
    for crit in (-1, 1):
        out = pp.kitti_eval.rotate_iou_eval(g["riou_boxes"], g["riou_qboxes"], crit)
        assert np.array_equal(out, g[f"riou_c{crit}"]), (crit, np.abs(out - g[f"riou_c{crit}"]).max())


@pytest.mark.gpu
def test_hip_rotate_iou_large_random_vs_oracle(pp, hip_lib):
    rng = np.random.default_rng(5)

    def rb(n):
        return np.concatenate([rng.uniform(-6, 6, (n, 2)), rng.uniform(0.2, 3.0, (n, 2)), rng.uniform(-7, 7, (n, 1))],
                              axis=1).astype(np.float32)
    b, q = rb(777), rb(1301)
    q[:50] = b[:50]                     # identical pairs on the diagonal
    q[50:80, :2] = b[50:80, :2]         # concentric, different size / angle
    for crit in (-1, 2):
        out = pp.kitti_eval.rotate_iou_eval(b, q, crit)
        ref = c_oracle.rotate_iou_eval(b, q, crit)
        assert np.array_equal(np.isnan(out), np.isnan(ref))
        np.testing.assert_array_equal(np.nan_to_num(out), np.nan_to_num(ref))
    assert pp.kitti_eval.rotate_iou_eval(b[:0], q).shape == (0, 1301)
    with pytest.raises(RuntimeError):
        pp.kitti_eval.rotate_iou_eval(b, q, criterion=7)


@pytest.mark.gpu
def test_hip_evaluator_end_to_end(pp, hip_lib):
    g = load_golden("ref_kitti_eval.npz")
    gts, dts = _annos(g)
    for metric in (1, 2):
        ov, _, _, _ = pp.kitti_eval.calculate_iou_partly(dts, gts, metric, num_parts=5)
        for i, o in enumerate(ov):
            np.testing.assert_allclose(o, g[f"ov_m{metric}_{i}"], rtol=0, atol=1e-6)
    text, _, mbev, m3d, maos = pp.kitti_eval.get_official_eval_result(gts, dts, ["Pedestrian"], compute_bbox=False)
    assert text == str(g["official_text"])
    np.testing.assert_allclose(mbev, g["official_bev"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(m3d, g["official_3d"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(maos, g["official_aos"], rtol=0, atol=1e-9)
