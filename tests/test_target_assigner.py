"""Training-side target assignment (SURVEY 8f, f3 data half) against outputs of the reference's own
create_target_np / assign / nearest_iou_similarity (tools/gen_golden_targets.py -> ref_targets.npz)."""
import os

import numpy as np
import pytest

import pp_amd as pp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_targets.npz"))
ta = pp.target_assigner
CFG = {"sample_positive_fraction": "None", "rpn_batch_size": 512}


def test_near_bbox_and_similarity_match_reference():
    assert np.array_equal(ta.rbbox2d_to_near_bbox(G["rb1"]), G["near1"])
    sim = ta.nearest_iou_similarity(G["rb1"], G["rb2"])
    assert sim.dtype == G["sim"].dtype and np.array_equal(sim, G["sim"])
    assert (sim > 0).sum() > 5 and (sim == 0).sum() > 5


@pytest.mark.parametrize("name", ["three", "offgrid", "empty", "tie"])
@pytest.mark.parametrize("masked", [False, True])
def test_assign_matches_reference(name, masked):
    tag = f"{name}_{'mask' if masked else 'all'}"
    gt = G[tag + "_gt"]
    mask = G[tag + "_anchors_mask"] if masked else None
    r = ta.assign(G["anchors"], gt, mask, np.ones(len(gt), np.int32), 0.5, 0.35, CFG)
    for k in ("labels", "bbox_targets", "bbox_outside_weights", "positive_gt_id", "assigned_anchors_inds"):
        want = G[tag + "_" + k]
        assert r[k].dtype == want.dtype, k
        assert np.array_equal(r[k], want), k
    if bool(G[tag + "_overlap_is_none"]):
        assert r["assigned_anchors_overlap"] is None
    else:
        assert np.array_equal(r["assigned_anchors_overlap"], G[tag + "_assigned_anchors_overlap"])
    if name == "three" and not masked:
        assert (r["labels"] > 0).sum() == 103 and (r["labels"] == -1).sum() > 0 and (r["labels"] == 0).sum() > 9000
    if name == "empty":
        assert (r["labels"][mask] == 0).all() if masked else (r["labels"] == 0).all()
        assert not r["bbox_targets"].any()


def test_encode_is_the_inverse_of_the_reference_decode():
    """bbox_targets -> the reference's numpy second_box_decode (fixture) returns the assigned boxes."""
    pos = G["three_all_assigned_anchors_inds"]
    gt = G["three_all_gt"][G["three_all_positive_gt_id"]]
    np.testing.assert_allclose(G["three_all_decoded"], gt, rtol=2e-6, atol=2e-6)
    enc = ta.second_box_encode(gt, G["anchors"][pos])
    assert enc.dtype == np.float32 and np.array_equal(enc, G["three_all_bbox_targets"][pos])


def test_positive_fraction_sampling_bounds():
    """sample_positive_fraction set: at most fraction * rpn_batch_size positives stay, negatives are drawn
    from the background set (numpy.random global state, like the reference)."""
    gt = G["three_all_gt"]
    np.random.seed(3)
    r = ta.assign(G["anchors"], gt, None, np.ones(len(gt), np.int32), 0.5, 0.35,
                  {"sample_positive_fraction": 0.1, "rpn_batch_size": 512})
    assert (r["labels"] > 0).sum() <= 51
    assert 0 < (r["labels"] == 0).sum() <= 512 - (r["labels"] > 0).sum()
    assert set(np.where(r["labels"] > 0)[0]) <= set(G["three_all_assigned_anchors_inds"])
