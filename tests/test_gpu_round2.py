"""Round-2 parity cases (all `-m gpu`, through the C-ABI):

* the exact batch shape bench.py times (cfg-A, B=64, staged asynchronous upload) against the oracle, frame by frame;
* BASELINE.json configs[2] as stated: KITTI-shaped grid, batch 32, two classes;
* the configuration branches the reference implements but the shipped YAML does not use
  (num_class > 1 score/label rule, use_direction_classifier=False, with_distance, reverse_index=False);
* the asynchronous entry points and the handle's state rules (graph eviction under load, weight reloads,
  device-resident input, stage calls invalidating the fused state).
"""
import os
import numpy as np
import pytest

from oracle import c_oracle, ref_numpy as rn
import util_ref

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _assert_dets(pp_dicts, ref_dicts, labels=False):
    assert len(pp_dicts) == len(ref_dicts)
    for a, b in zip(pp_dicts, ref_dicts):
        if b["scores"] is None:
            assert a["scores"] is None and a["box3d_lidar"] is None
            continue
        assert a["scores"] is not None and a["scores"].shape == b["scores"].shape
        np.testing.assert_allclose(a["scores"], b["scores"], rtol=0, atol=TOL)
        np.testing.assert_allclose(a["box3d_lidar"], b["box3d_lidar"], rtol=0, atol=TOL)
        np.testing.assert_allclose(a["box3d_camera"], b["box3d_camera"], rtol=0, atol=TOL)
        assert np.array_equal(a["label_preds"], b["label_preds"])


# ------------------------------------------------------------------ the benchmarked shape, against the oracle
def test_bench_batch_b64_matches_oracle_per_frame(pp, hip_lib):
    """cfg-A at B=64 is the only shape that runs the kernel instantiations bench.py times
    (k_sep_u<128,1,2,1,0>, k_sep_u<64,1,3,1,0>, k_sep_p, k_deconv_u<128,3> for deconv1, k_deconv_r<128|256>, persistent grids): ALL 64 frames of the
    bench's own first batch -- same frame ids, same weights, uploaded from a page-locked staging buffer with
    pp_upload_points_async like the bench does -- against the oracle, head maps and detections."""
    B, N = 64, 16384
    eng = pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=N)
    d = eng.d
    w = pp.weights.init_weights(d, seed=7)                     # bench.py's weights
    eng.load_weights(w)
    tags = eng.layer_tags()
    for want in ("k_sep_u<128,1,2,1,0>", "k_sep_u<64,1,3,1,0>", "k_sep_p", "k_deconv_u<128,3>", "k_deconv_r<128>",
                 "k_deconv_r<256>"):
        assert any(t.startswith(want + ":") for t in tags), (want, tags)
    frames = [pp.synth.d435i_cloud(i, N, d.num_point_features) for i in pp.frame_shard.rank_frames(0, 1, B)]
    rect, trv, p2 = pp.synth.default_calib()
    st = eng.staging(frames)
    eng.set_calib(np.stack([rect] * B), np.stack([trv] * B), B)
    eng.upload_async(st)
    eng.detect_async()
    dets, n = eng.detections()                                  # waits
    im = eng.intermediates()
    for b in range(B):                                          # every frame: the oracle takes ~80 ms per frame
        ref = util_ref.oracle_detect(d, w, [frames[b]], rect, trv, p2)
        fr = ref["frames"][0]
        P = fr["coordinates"].shape[0]
        assert im["n_pillars"][b] == P and np.array_equal(im["coors"][b, :P], fr["coordinates"])
        assert np.array_equal(im["num_points"][b, :P], fr["num_points"])
        assert np.array_equal(im["anchors_mask"][b].astype(bool), fr["anchors_mask"])
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(im[k][b], ref["preds"][k][0], rtol=0, atol=TOL, err_msg=f"frame {b} {k}")
        _assert_dets([pp.VoxelNet._to_dict(dets[b], int(n[b]), 0)], ref["dets"])
    assert int(n.sum()) > B, "the synthetic frames must produce detections"
    st.close()
    eng.close()


# ------------------------------------------------------------------ configs[2]: KITTI-shaped, batch 32, two classes
def test_kitti_shaped_batch32_two_classes(pp, hip_lib):
    B, N = 32, 20000
    cfg = pp.config.kitti_shaped_config(B, num_class=2)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=N)
    d = eng.d
    assert (d.nx, d.ny, d.num_class, d.num_anchor_per_loc) == (432, 496, 2, 2)
    w = pp.weights.init_weights(d, seed=5)
    eng.load_weights(w)
    frames = [pp.synth.kitti_cloud(300 + i, N) for i in range(B)]
    rect, trv, p2 = pp.synth.default_calib()
    dets, n = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    im = eng.intermediates()
    # size-independent properties over the whole batch
    dets2, n2 = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    assert np.array_equal(n, n2) and dets.tobytes() == dets2.tobytes(), "bit-reproducible"
    labels_seen = set()
    for b in range(B):
        k = int(n[b])
        assert 0 < k <= d.nms_post_max_size
        s = dets[b]["score"][:k]
        assert (np.diff(s) <= 0).all() and ((s > 0) & (s < 1)).all()
        labels_seen |= set(int(v) for v in dets[b]["label"][:k])
        P = im["n_pillars"][b]
        assert 0 < P <= d.max_voxels
        flat = im["coors"][b, :P, 1] * d.nx + im["coors"][b, :P, 2]
        assert len(np.unique(flat)) == P and (im["coors"][b, :P, 0] == 0).all()
        assert im["num_points"][b, :P].min() >= 1 and im["num_points"][b, :P].max() <= d.max_points
    assert labels_seen == {0, 1}, labels_seen
    # frame sub-ranges (pp_set_cache_budget): with the default budget block1's layers run as four launches of 8 frames and
    # block2's as two of 16, walked sub-range by sub-range; without it one launch per layer -- and the same bits
    def launches_of(layer):
        eng.set_profiling(True)
        d_, n_ = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
        k = sum(1 for name, _ in eng.kernel_times() if name.endswith(":" + layer))
        eng.set_profiling(False)
        return k, d_, n_
    k1, d1, n1 = launches_of("block1.2")
    k2, _, _ = launches_of("block2.3")
    assert (k1, k2) == (4, 2), (k1, k2)
    assert np.array_equal(n1, n) and d1.tobytes() == dets.tobytes()
    eng.set_cache_budget(0)
    k0, d0, n0 = launches_of("block1.2")
    assert k0 == 1 and np.array_equal(n0, n) and d0.tobytes() == dets.tobytes(), "sub-ranges do not change a bit"
    eng.set_cache_budget(256)
    # the feed the bench uses: page-locked staging, voxelised at upload time on the copy stream into the handle's OTHER set
    # of voxeliser products (cell map, CSR, occupancy bitmap), three uploads so that both sets are used -- same bits
    st = eng.staging(frames)
    for _ in range(3):
        eng.upload_async(st)
        eng.detect_async()
        dets3, n3 = eng.detections()
        assert np.array_equal(n, n3) and dets.tobytes() == dets3.tobytes(), "upload-time voxeliser path"
    im3 = eng.intermediates()
    assert np.array_equal(im3["n_pillars"], im["n_pillars"]) and np.array_equal(im3["coors"], im["coors"])
    assert np.array_equal(im3["anchors_mask"], im["anchors_mask"])
    st.close()
    # eight frames against the oracle (pillar indices bit-exact, head maps and detections within 1e-4)
    for b in (0, 3, 7, 12, 17, 22, 30, 31):
        ref = util_ref.oracle_detect(d, w, [frames[b]], rect, trv, p2)
        fr = ref["frames"][0]
        P = fr["coordinates"].shape[0]
        assert im["n_pillars"][b] == P and np.array_equal(im["coors"][b, :P], fr["coordinates"])
        assert np.array_equal(im["anchors_mask"][b].astype(bool), fr["anchors_mask"])
        assert im["cls_preds"].shape[-1] == 4
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(im[k][b], ref["preds"][k][0], rtol=0, atol=TOL, err_msg=f"frame {b} {k}")
        _assert_dets([pp.VoxelNet._to_dict(dets[b], int(n[b]), 0)], ref["dets"])
    eng.close()


# ------------------------------------------------------------------ config branches
def test_multi_class_score_and_label_rule(pp, hip_lib):
    """num_class = 3 on hand-made head maps: score = largest class score, label = argmax (first maximum),
    threshold on that score (model/voxelnet.py:1183-1203)."""
    cfg = pp.config.pedestrian_d435i_config(1)
    cfg["model"]["second"]["num_class"] = 3
    cfg["model"]["second"]["nms_score_threshold"] = 0.3
    eng = pp.Engine(cfg, max_batch=1, max_points_per_frame=4096)
    d = eng.d
    rng = np.random.default_rng(5)
    box = (rng.standard_normal((1, d.head_h, d.head_w, 14)) * 0.3).astype(np.float32)
    cls = (rng.standard_normal((1, d.head_h, d.head_w, 6)) * 0.8).astype(np.float32)
    dr = rng.standard_normal((1, d.head_h, d.head_w, 4)).astype(np.float32)
    mask = (rng.random((1, d.num_anchors)) < 0.5).astype(np.uint8)
    rect, trv, _ = pp.synth.default_calib()
    dets, n = eng.predict(box, cls, dr, mask, rect[None], trv[None])
    ex = (None, None, None, rect[None], trv[None], None, eng.anchors[None], mask, np.array([0]), None)
    ref = rn.predict(ex, {"box_preds": box, "cls_preds": cls, "dir_cls_preds": dr}, d.nms_dict())[0]
    k = int(n[0])
    assert k == len(ref["scores"]) > 5
    np.testing.assert_allclose(dets[0]["score"][:k], ref["scores"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dets[0]["box3d_lidar"][:k], ref["box3d_lidar"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(dets[0]["label"][:k], ref["label_preds"])
    assert len(set(dets[0]["label"][:k].tolist())) == 3
    eng.close()


@pytest.mark.parametrize("use_dir,with_dist,F", [(False, False, 3), (True, True, 3), (False, True, 4)])
def test_direction_and_distance_branches(pp, hip_lib, use_dir, with_dist, F):
    """use_direction_classifier=False (no conv_dir_cls, no flip: model/voxelnet.py:690,714,1297) and
    with_distance=True (the point's Euclidean norm as an extra PFN feature: model/pointpillars.py:185-188),
    whole path against the oracle on a small grid."""
    B = 2
    cfg = pp.config.tiny_config(B)
    s = cfg["model"]["second"]
    s["use_direction_classifier"] = use_dir
    s["voxel_feature_extractor"]["with_distance"] = with_dist
    s["num_point_features"] = F
    cfg["eval_input_reader"]["num_point_features"] = F
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=4096)
    d = eng.d
    assert d.pfn_in == F + 5 + (1 if with_dist else 0)
    w = pp.weights.init_weights(d, seed=31)
    assert ("rpn/conv_dir_cls/kernel" in w) == use_dir
    eng.load_weights(w)
    rng = np.random.default_rng(77)
    frames = []
    for npts in (900, 350):
        xyz = rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (npts, 3))
        extra = rng.uniform(0, 1, (npts, F - 3))
        frames.append(np.concatenate([xyz, extra], axis=1).astype(np.float32))
    rect, trv, p2 = pp.synth.default_calib()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    dets, n = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    im = eng.intermediates(canvas=True)
    np.testing.assert_allclose(im["canvas"], ref["canvas"], rtol=1e-5, atol=1e-5)
    assert ("dir_cls_preds" in im) == use_dir and ("dir_cls_preds" in ref["preds"]) == use_dir
    for k in ref["preds"]:
        np.testing.assert_allclose(im[k], ref["preds"][k], rtol=0, atol=TOL)
    _assert_dets([pp.VoxelNet._to_dict(dets[b], int(n[b]), b) for b in range(B)], ref["dets"])
    if not use_dir:
        assert all((dets[b]["dir_label"][:n[b]] == 0).all() for b in range(B))
    # the padded-voxel entry point (VoxelNet.__call__) runs the same branches
    ex = ref["example"]
    out = eng.forward_voxels(ex[0], ex[1], ex[2], B)
    assert set(out) == set(ref["preds"])
    for k in ref["preds"]:
        np.testing.assert_allclose(out[k], ref["preds"][k], rtol=0, atol=TOL)
    eng.close()


def test_points_to_voxel_reverse_index_false(pp, hip_lib):
    """reverse_index=False (_points_to_voxel_kernel, load_data.py:643-692): same pillars in the same order, the
    coordinate columns are (x, y, z) instead of (z, y, x) -- against the reference's own output."""
    from conftest import load_golden
    g, gf = load_golden("ref_voxel.npz"), load_golden("ref_voxel_fwd.npz")
    d = pp.config.Derived(pp.config.pedestrian_d435i_config())
    for case in ("a2k", "brk"):
        T, MV = (int(v) for v in gf[case + "_params"])
        v, c, n = pp.points_to_voxel(g[case + "_points"], d.voxel_size, d.pc_range, T, False, MV)
        assert np.array_equal(c, gf[case + "_coors"]) and np.array_equal(n, gf[case + "_num"])
        assert np.array_equal(v, gf[case + "_voxels"])


# ------------------------------------------------------------------ asynchronous entry points, handle state
def test_graph_eviction_under_async_load(pp, hip_lib):
    """More (batch, point-bucket) keys than graph slots, enqueued back to back with NO sync in between: evicting
    a graph that may still be replaying must wait for it.  Every result is checked against a plain-
    launch pass of the same frames."""
    eng = pp.Engine(pp.config.pedestrian_d435i_config(4), max_batch=4, max_points_per_frame=24576)
    eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
    rect, trv, _ = pp.synth.default_calib()
    shapes = [(4, 3000), (3, 7000), (2, 11000), (4, 15000), (1, 19000), (2, 23000), (4, 3000), (3, 7000)]
    batches = [[pp.synth.d435i_cloud(900 + 10 * k + i, npts - 11 * i) for i in range(nb)] for k, (nb, npts) in enumerate(shapes)]
    stagings = [eng.staging(f) for f in batches]
    eng.set_calib(np.stack([rect] * 4), np.stack([trv] * 4), 4)
    results = []
    for rounds in range(2):
        for st in stagings:                       # no sync between the enqueues of different keys
            eng.upload_async(st)
            eng.detect_async()
        results.append([a.copy() for a in eng.detections()])      # last batch of the round
    eng.set_profiling(True)                       # plain launches, no graph
    for k, st in enumerate(stagings):
        eng.upload_async(st)
        eng.detect_async()
        dets, n = eng.detections()
        if k == len(stagings) - 1:
            for r in results:
                assert np.array_equal(r[1][:len(n)], n) and r[0][:len(n)].tobytes() == dets.tobytes()
    eng.set_profiling(False)
    # and every key once more, each checked (graphs were re-captured several times above)
    for k, st in enumerate(stagings):
        eng.upload_async(st)
        eng.detect_async()
        d1, n1 = [a.copy() for a in eng.detections()]
        eng.set_profiling(True)
        eng.upload_async(st)
        eng.detect_async()
        d2, n2 = eng.detections()
        eng.set_profiling(False)
        assert np.array_equal(n1, n2) and d1.tobytes() == d2.tobytes(), k
    for st in stagings:
        st.close()
    eng.close()


def test_reloading_weights_does_not_leak(pp, hip_lib):
    eng = pp.Engine(pp.config.pedestrian_d435i_config(1), max_batch=1, max_points_per_frame=4096)
    w = pp.weights.init_weights(eng.d, seed=3)
    eng.load_weights(w)
    eng.load_weights(w)
    free0 = eng.device_mem_free()
    for _ in range(20):
        eng.load_weights(w)
    free1 = eng.device_mem_free()
    assert free0 - free1 < 4 << 20, f"{(free0 - free1) / 2**20:.1f} MiB lost over 20 reloads"
    # and the reloaded weights are live: different weights, different result
    frames = [pp.synth.d435i_cloud(5, 4096)]
    eng.detect(frames)
    a = eng.intermediates()["cls_preds"].copy()
    eng.load_weights(pp.weights.init_weights(eng.d, seed=4))
    eng.detect(frames)
    assert np.abs(eng.intermediates()["cls_preds"] - a).max() > 1e-3
    eng.close()


def test_upload_from_device_memory_with_producer_stream(pp, hip_lib):
    """pp_upload_points_device: the engine's stream waits for the producer's stream (a side stream of the HIP
    runtime, still busy ahead of the copy that writes the points when the call is made)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")            # the process's one HIP runtime (pp_amd._lib._one_hip_runtime)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def ok(rc):
        assert rc == 0, f"hip error {rc}"
    B = 2
    eng = pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=16384)
    eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
    frames = [pp.synth.d435i_cloud(70 + i) for i in range(B)]
    want = [a.copy() for a in eng.detect(frames)]
    pts = np.ascontiguousarray(np.concatenate(frames, axis=0), np.float32)
    offs = np.array([0, frames[0].shape[0], pts.shape[0]], np.int32)
    side, dev, big = C.c_void_p(), C.c_void_p(), C.c_void_p()
    big_bytes = 1 << 30
    ok(hip.hipStreamCreate(C.byref(side)))
    ok(hip.hipMalloc(C.byref(dev), pts.nbytes))
    ok(hip.hipMalloc(C.byref(big), big_bytes))
    ok(hip.hipMemsetAsync(dev, 0xff, pts.nbytes, side))      # NaNs until the copy lands
    for k in range(20):                                       # keep the producer stream busy ahead of the copy
        ok(hip.hipMemsetAsync(big, k, big_bytes, side))
    ok(hip.hipMemcpyAsync(dev, pts.ctypes.data_as(C.c_void_p), pts.nbytes, 1, side))
    eng.upload_device(dev.value, offs, producer_stream=side.value)
    eng.detect_async()
    dets, n = eng.detections()
    assert np.array_equal(n, want[1]) and dets.tobytes() == want[0].tobytes()
    ok(hip.hipStreamSynchronize(side))
    eng.close()
    ok(hip.hipFree(dev))
    ok(hip.hipFree(big))
    ok(hip.hipStreamDestroy(side))


def test_stage_calls_invalidate_the_fused_state(pp, hip_lib):
    eng = pp.Engine(pp.config.pedestrian_d435i_config(2), max_batch=2, max_points_per_frame=16384)
    eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
    with pytest.raises(RuntimeError, match="PP_ERR_STATE"):
        eng.detections()                          # nothing has run yet
    frames = [pp.synth.d435i_cloud(80 + i) for i in range(2)]
    dets, n = eng.detect(frames)
    assert eng._batches() == (2, 2)
    eng.points_to_voxel(frames[0])                # reuses the point / cell-map buffers of the fused path
    assert eng._batches() == (0, 0)
    for call in (eng.detections, eng.intermediates, eng.detect_async):
        with pytest.raises(RuntimeError, match="PP_ERR_STATE"):
            call()
    dets2, n2 = eng.detect(frames)                # a fresh upload restores it
    assert np.array_equal(n, n2) and dets.tobytes() == dets2.tobytes()
    eng.close()


# ------------------------------------------------------------------ every voxeliser path, bit-exact
@pytest.mark.parametrize("sizes", [(33000, 17, 40000, 0, 16384), (32768, 32769), (20000, 5000, 20001, 16385, 1, 9, 12345, 777, 3)])
def test_voxeliser_paths_beyond_the_lds_capacity(pp, hip_lib, sizes):
    """Frames of more than 32 768 points take k_voxel_frame's global-memory path (A-D over the key / index
    buffers) and k_sort_points gathers from its index buffer; batches whose largest frame is <= 32 768 take the
    32-points-per-thread LDS path for every frame (ragged sizes, offsets not multiples of 4, a batch that does not
    fill a group of 8 XCDs).  Pillar order, coordinates, point counts AND the padded voxel contents against the C
    oracle (the reference's sequential loop), with the `break` inside the large frames."""
    rng = np.random.default_rng(sum(sizes))
    B = len(sizes)
    cfg = pp.config.pedestrian_d435i_config(B)
    cfg["model"]["second"]["voxel_generator"].update(max_number_of_points_per_voxel=9, max_number_of_voxels=2500)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=max(max(sizes), 4096))
    eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
    d = eng.d
    frames = []
    for n in sizes:
        pts = np.stack([rng.uniform(-0.3, 6.8, n), rng.uniform(-2.8, 2.8, n), rng.uniform(-3.2, 3.2, n)], 1).astype(np.float32)
        if n > 10:
            pts[rng.integers(0, n, n // 7)] = pts[rng.integers(0, n, n // 7)]   # duplicates
        frames.append(pts)
    rect, trv, _ = pp.synth.default_calib()
    eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    im = eng.intermediates()
    for b, pts in enumerate(frames):
        ve, ce, ne = c_oracle.points_to_voxel(pts, d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
        P = ce.shape[0]
        assert im["n_pillars"][b] == P, (sizes[b], im["n_pillars"][b], P)
        assert np.array_equal(im["coors"][b, :P], ce), sizes[b]
        assert np.array_equal(im["num_points"][b, :P], ne), sizes[b]
    for b in np.argsort(sizes)[-2:]:                          # the padded tensor too (contents = the sorted copy)
        v, c, k = eng.points_to_voxel(frames[b])
        ve, ce, ne = c_oracle.points_to_voxel(frames[b], d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
        assert np.array_equal(c, ce) and np.array_equal(k, ne) and np.array_equal(v, ve), sizes[b]
    eng.close()


@pytest.mark.parametrize("B", [1, 3, 4, 5])
def test_zero_copy_feed_of_small_batches(pp, hip_lib, B):
    """pp_upload_points_async with up to 4 frames feeds the first kernel straight from the page-locked buffer (no copy
    engine, no events); 5 frames go through the copy stream.  Same detections, bit for bit, as the synchronous
    upload, over many steps with the staging buffers recycled and the batch shape changing in between."""
    eng = pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=16384)
    eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
    rect, trv, _ = pp.synth.default_calib()
    eng.set_calib(np.stack([rect] * B), np.stack([trv] * B), B)
    sets = [[pp.synth.d435i_cloud(900 + 10 * k + i, 16384 - 211 * i - 17 * k) for i in range(B)] for k in range(5)]
    def same(d, n, w):           # rows past a frame's count are not defined
        return np.array_equal(n, w[1]) and all(d[b, :n[b]].tobytes() == w[0][b, :n[b]].tobytes() for b in range(len(n)))

    want = []
    for fr in sets:
        d, n = eng.detect(fr, np.stack([rect] * B), np.stack([trv] * B))
        want.append((d.copy(), n.copy()))
    stg = [eng.staging(fr) for fr in sets]
    for rep in range(3):
        for k in range(5):
            eng.upload_async(stg[k])
            eng.detect_async()
            d, n = eng.detections()
            assert same(d, n, want[k]), (rep, k)
        if B > 1:                                           # another batch shape in between (other graph, other key)
            eng.detect(sets[0][:1], rect[None], trv[None])
    # pipelined: upload k+1 while k runs
    eng.upload_async(stg[0]); eng.detect_async()
    for k in range(1, 5):
        eng.upload_async(stg[k])
        d, n = eng.detections()
        assert same(d, n, want[k - 1])
        eng.detect_async()
    d, n = eng.detections()
    assert same(d, n, want[4])
    for s_ in stg:
        s_.close()
    eng.close()


def test_async_upload_of_pageable_memory_falls_back_to_the_copy(pp, hip_lib):
    """pp_upload_points_async with ordinary (not page-locked) memory: the zero-copy feed cannot map it, the call
    falls back to the copy path and the results are the same."""
    import types
    eng = pp.Engine(pp.config.pedestrian_d435i_config(1), max_batch=1, max_points_per_frame=16384)
    eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
    fr = [pp.synth.d435i_cloud(77)]
    d0, n0 = eng.detect(fr)
    d0, n0 = d0.copy(), n0.copy()
    fake = types.SimpleNamespace(points=np.ascontiguousarray(fr[0], np.float32),
                                 offsets=np.array([0, fr[0].shape[0]], np.int32))
    for _ in range(3):
        eng.upload_async(fake)
        eng.detect_async()
        d, n = eng.detections()
        assert np.array_equal(n, n0) and d[0, :n[0]].tobytes() == d0[0, :n0[0]].tobytes()
    eng.close()


def test_zero_copy_feed_buffer_lifetimes(pp, hip_lib):
    """The zero-copy feed (batches of <= 4 frames: the pass's first kernel reads the caller's page-locked buffer)
    against the two lifetime holes of round 2:
    (1) a TEMPORARY Staging -- `eng.upload_async(eng.staging(frames))` -- must stay alive until the pass that reads
        it is through (the engine keeps the last two);
    (2) a staging buffer is freed and ordinary pageable memory appears at the SAME address: the library must not
        remember the old device mapping -- the call has to take the copy path and give the same result."""
    import ctypes
    import mmap
    import types
    eng = pp.Engine(pp.config.pedestrian_d435i_config(1), max_batch=1, max_points_per_frame=16384)
    eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
    fr = [pp.synth.d435i_cloud(78)]
    d0, n0 = eng.detect(fr)
    d0, n0 = d0.copy(), n0.copy()
    assert int(n0[0]) > 0

    def same():
        d, n = eng.detections()
        return np.array_equal(n, n0) and d[0, :n[0]].tobytes() == d0[0, :n0[0]].tobytes()

    # (1) temporaries: nothing but the engine references the buffers while the kernels read them
    for _ in range(3):
        eng.upload_async(eng.staging(fr))
        eng.detect_async()
        assert same()
    # (2) pinned -> freed -> pageable at the same address
    st = eng.staging(fr)
    addr, nbytes = st._p.value, st.points.nbytes
    eng.upload_async(st)
    eng.detect_async()
    assert same()
    st.close()                                           # waits for the engine, unregisters, hipHostFree
    length = (nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mmap.restype = ctypes.c_void_p
    libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    MAP_PRIVATE, MAP_ANONYMOUS, MAP_FIXED_NOREPLACE = 0x02, 0x20, 0x100000
    got = libc.mmap(ctypes.c_void_p(addr), length, mmap.PROT_READ | mmap.PROT_WRITE,
                    MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0)
    if got in (None, ctypes.c_void_p(-1).value) or got != addr:
        if got not in (None, ctypes.c_void_p(-1).value):
            libc.munmap(ctypes.c_void_p(got), length)
        eng.close()
        pytest.skip("the freed staging address could not be re-mapped as pageable memory on this box")
    try:
        view = np.ctypeslib.as_array(ctypes.cast(ctypes.c_void_p(addr), ctypes.POINTER(ctypes.c_float)),
                                     shape=(fr[0].size,)).reshape(fr[0].shape)
        view[...] = fr[0]
        fake = types.SimpleNamespace(points=view, offsets=np.array([0, fr[0].shape[0]], np.int32))
        for _ in range(2):
            eng.upload_async(fake)                       # same address as the freed pinned block
            eng.detect_async()
            assert same()
        eng.sync()
    finally:
        eng.close()
        libc.munmap(ctypes.c_void_p(addr), length)


def test_folded_weights_outside_float16_range_fall_back_to_f32(pp, hip_lib):
    """The split-precision GEMM kernels carry a float32 operand as two float16 pieces, which needs |w| < 65504 for the
    BN-folded weights.  A layer whose folded weights do not fit must run on the float32 matrix instruction instead of
    silently producing inf / NaN: deconv3 with BatchNorm gammas of 1e7 (its folded kernel reaches ~1e5-1e6), compensated
    in the head kernels' rows of that branch, so the outputs stay O(1) and comparable with the oracle."""
    B = 2
    cfg = pp.config.pedestrian_d435i_config(B)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=8192)
    d = eng.d
    w = dict(pp.weights.init_weights(d, seed=41))
    big = np.float32(1e7)
    w["rpn/deconv3/bn/gamma"] = w["rpn/deconv3/bn/gamma"] * big
    w["rpn/deconv3/bn/beta"] = w["rpn/deconv3/bn/beta"] * big
    c0 = 2 * 128                                                    # deconv3's slice of the 384 concat channels
    for k in ("rpn/conv_box/kernel", "rpn/conv_cls/kernel", "rpn/conv_dir_cls/kernel"):
        kk = w[k].copy()
        kk[:, :, c0:, :] /= big
        w[k] = kk
    eng.load_weights(w)
    tags = eng.layer_tags()
    assert not any(t.startswith(("k_deconv_r", "k_deconv_u", "k_deconv_k4")) and t.endswith(":deconv3") for t in tags), tags
    assert any(t.startswith("k_deconv") and t.endswith(":deconv2") for t in tags), tags   # the others keep their kernels
    frames = [pp.synth.d435i_cloud(500 + i, 6000) for i in range(B)]
    rect, trv, p2 = pp.synth.default_calib()
    eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    im = eng.intermediates()
    ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        assert np.isfinite(im[k]).all()
        np.testing.assert_allclose(im[k], ref["preds"][k], rtol=0, atol=TOL)
    eng.close()


def test_depthwise_once_kernel_ragged_tiles(pp, hip_lib):
    """k_sep_p (256-channel layers, 8-wave workgroups, chunk pairs) on a shape whose pixel count ends inside a
    128-pixel tile: a 40x24 grid (block3 map 10x6 = 60 pixels per frame), 110 frames -> 6 600 pixels = 51.6 tiles.
    Head maps of the big batch against the oracle for two frames, and against the same frames run alone (split-K
    kernels) for three."""
    import copy
    B = 110
    cfg = copy.deepcopy(pp.config.pedestrian_d435i_config(B))
    cfg["eval_input_reader"]["feature_map_size"] = [1, 24, 40]
    s = cfg["model"]["second"]
    s["voxel_generator"].update(point_cloud_range=[0, -0.96, -3.0, 3.2, 0.96, 3.0], max_number_of_voxels=960)
    s["target_assigner"]["anchor_generators"]["anchor_generator_stride"].update(offsets=[0.08, -0.96, -1.465])
    big = pp.Engine(cfg, max_batch=B, max_points_per_frame=4096)
    d = big.d
    assert (d.nx, d.ny) == (40, 24)
    w = pp.weights.init_weights(d, seed=13)
    big.load_weights(w)
    tags = big.layer_tags()
    assert sum(t.startswith("k_sep_p:") for t in tags) == 5, tags
    # the same shape ends inside a tile for k_deconv_r too (6 600 deconv3 pixels = 51.6 tiles, 832 units on 416 workgroups)
    assert any(t.startswith("k_deconv_r<256>:") for t in tags) and any(t.startswith("k_deconv_r<128>:") for t in tags), tags
    cfg1 = copy.deepcopy(cfg)
    one = pp.Engine(cfg1, max_batch=1, max_points_per_frame=4096)
    one.load_weights(w)
    assert any(t.startswith("k_sep_k4") for t in one.layer_tags())
    rng = np.random.default_rng(5)
    frames = [rng.uniform([0, -0.96, -3], [3.2, 0.96, 3], (int(n), 3)).astype(np.float32)
              for n in rng.integers(800, 3000, B)]
    rect, trv, p2 = pp.synth.default_calib()
    big.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    imb = big.intermediates()
    for i in (0, 57, 109):
        one.detect([frames[i]], rect[None], trv[None])
        im1 = one.intermediates()
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(im1[k][0], imb[k][i], rtol=2e-5, atol=2e-5)
    for i in (3, 109):
        ref = util_ref.oracle_detect(d, w, [frames[i]], rect, trv, p2)
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(imb[k][i], ref["preds"][k][0], rtol=0, atol=TOL)
    big.close()
    one.close()


@pytest.mark.parametrize("B", [5, 33])
def test_deconv_r_unit_runs_at_odd_batches(pp, hip_lib, B):
    """k_deconv_r cuts the (tile, tap) units of a layer into equal runs per workgroup: batches whose unit count does not
    divide (B = 5: deconv3 has 12.5 tiles, 208 units, one per workgroup; B = 33: 1 328 units, 3 per workgroup with a short
    last run) -- head maps against the same frames run alone, and against the oracle for one frame."""
    cfg = pp.config.pedestrian_d435i_config(B)
    big = pp.Engine(cfg, max_batch=B, max_points_per_frame=8192)
    d = big.d
    w = pp.weights.init_weights(d, seed=17)
    big.load_weights(w)
    tags = big.layer_tags()
    assert any(t.startswith("k_deconv_r<256>:") for t in tags) and any(t.startswith("k_deconv_r<128>:") for t in tags), tags
    one = pp.Engine(pp.config.pedestrian_d435i_config(1), max_batch=1, max_points_per_frame=8192)
    one.load_weights(w)
    frames = [pp.synth.d435i_cloud(700 + i, 5000 + 37 * i) for i in range(B)]
    rect, trv, p2 = pp.synth.default_calib()
    big.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    imb = big.intermediates()
    for i in (0, B // 2, B - 1):
        one.detect([frames[i]], rect[None], trv[None])
        im1 = one.intermediates()
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(im1[k][0], imb[k][i], rtol=2e-5, atol=2e-5, err_msg=f"frame {i} {k}")
    ref = util_ref.oracle_detect(d, w, [frames[B - 1]], rect, trv, p2)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(imb[k][B - 1], ref["preds"][k][0], rtol=0, atol=TOL, err_msg=k)
    big.close()
    one.close()


def test_results_written_by_the_post_process_itself(pp, hip_lib):
    """Round 3: k_postprocess stores a frame's kept detections straight into the page-locked result buffers (no copy
    nodes behind the fused path).  What the host reads must be the kept entries and zeros behind them -- also when the
    pass before left MORE detections in the same frame slot -- and a second handle running the same frames must agree."""
    cfg = pp.config.pedestrian_d435i_config(4)
    d = pp.config.Derived(cfg)
    w = pp.weights.init_weights(d, seed=7)
    eng = pp.Engine(cfg, max_batch=4, max_points_per_frame=20000)
    eng.load_weights(w)
    rich = [pp.synth.d435i_cloud(40 + i, 16384) for i in range(4)]
    poor = [pp.synth.d435i_cloud(60 + i, 16384)[:300] for i in range(4)]      # a few hundred points: fewer boxes kept
    d1, n1 = eng.detect(rich)
    d1, n1 = d1.copy(), n1.copy()
    d2, n2 = eng.detect(poor)
    assert (n2 <= n1).all() and (n2 < n1).any(), (n1, n2)
    zero = np.zeros((), dtype=d2.dtype)
    for b in range(4):
        assert (d2[b, n2[b]:] == zero).all(), "entries behind the kept detections must read as zeros"
    d3, n3 = eng.detect(rich)
    assert np.array_equal(n3, n1) and all(np.array_equal(d3[b, :n1[b]], d1[b, :n1[b]]) for b in range(4))
    other = pp.Engine(cfg, max_batch=4, max_points_per_frame=20000)
    other.load_weights(w)
    d4, n4 = other.detect(poor)
    assert np.array_equal(n4, n2) and all(np.array_equal(d4[b, :n2[b]], d2[b, :n2[b]]) for b in range(4))
    other.close()
    eng.close()



# ------------------------------------------------------------------ activation range of the float16 operand pieces
def _out_of_range_weights(pp, d, where):
    """Seeded weights changed so that ONE tensor of activations leaves the float16 range (|x| >= 65504) while every
    BN-folded weight stays inside it, compensated in the consumer so that the network's outputs stay O(1):
      sep     block2.3's BatchNorm scaled by 1e5 -> block2.4's depthwise output (the operand k_sep_u / k_sep_k4 split),
              block2.4's pointwise kernel divided by 1e5
      deconv  block3.5's BatchNorm scaled by 1e5 -> deconv3's input (the operand k_deconv_r / k_deconv_k4 split),
              deconv3's kernel divided by 1e5
      head    deconv2's BatchNorm shift + 1e5 -> the head GEMM's operand (split in the deconv epilogue); not compensated:
              the logits become O(1e4) and are compared relatively"""
    w = dict(pp.weights.init_weights(d, seed=23))
    s = np.float32(1e5)
    if where == "sep":
        w["rpn/block2/3/bn/gamma"] = w["rpn/block2/3/bn/gamma"] * s
        w["rpn/block2/3/bn/beta"] = w["rpn/block2/3/bn/beta"] * s
        w["rpn/block2/4/pointwise_kernel"] = w["rpn/block2/4/pointwise_kernel"] / s
    elif where == "deconv":
        w["rpn/block3/5/bn/gamma"] = w["rpn/block3/5/bn/gamma"] * s
        w["rpn/block3/5/bn/beta"] = w["rpn/block3/5/bn/beta"] * s
        w["rpn/deconv3/kernel"] = w["rpn/deconv3/kernel"] / s
    else:
        w["rpn/deconv2/bn/beta"] = w["rpn/deconv2/bn/beta"] + s
    return w


@pytest.mark.parametrize("where", ["sep", "deconv", "head"])
@pytest.mark.parametrize("B", [2, 40])
def test_activations_outside_float16_range_raise_or_fall_back(pp, hip_lib, where, B):
    """The split-precision kernels split ACTIVATIONS into two float16 pieces in registers; a value beyond 65504 becomes
    inf - inf.  That must never come back as boxes: the default arithmetic reports PP_ERR_NUMERIC (NumericError), and
    Engine.detect's fallback re-runs the resident frames on the float32 matrix instruction and matches the oracle.
    B = 2 runs the split-K small-map kernels, B = 40 the persistent ones (k_sep_u / k_sep_p / k_deconv_r)."""
    cfg = pp.config.pedestrian_d435i_config(B)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=8192)
    d = eng.d
    w = _out_of_range_weights(pp, d, where)
    eng.load_weights(w)
    tags = eng.layer_tags()
    assert all(t.split(":")[0].startswith(("k_sep_", "k_deconv_")) for t in tags), tags   # no weight-range fallback here
    frames = [pp.synth.d435i_cloud(800 + i, 6000) for i in range(B)]
    rect, trv, p2 = pp.synth.default_calib()
    R, T = np.stack([rect] * B), np.stack([trv] * B)
    assert eng.gemm_precision() == "split_f16"
    with pytest.raises(pp.NumericError, match="PP_ERR_NUMERIC"):
        eng.detect(frames, R, T, on_numeric="raise")
    with pytest.raises(pp.NumericError):                         # asking again does not hand the buffers out either
        eng.detections()
    dets, n = eng.detect(frames, R, T)                           # default: float32 fallback on the resident frames
    assert eng.gemm_precision() == "f32"
    assert any(t.startswith(("k_gemm_ws", "k_gemm_layer")) for t in eng.layer_tags())
    im = eng.intermediates()
    picks = range(B) if B <= 2 else (0, B // 2, B - 1)
    for b in picks:
        ref = util_ref.oracle_detect(d, w, [frames[b]], rect, trv, p2)
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            assert np.isfinite(im[k][b]).all()
            if where == "head":      # logits of O(1e4): float32 round-off of both sides scales with them
                np.testing.assert_allclose(im[k][b], ref["preds"][k][0], rtol=2e-5, atol=TOL, err_msg=f"frame {b} {k}")
            else:
                np.testing.assert_allclose(im[k][b], ref["preds"][k][0], rtol=0, atol=TOL, err_msg=f"frame {b} {k}")
    if where != "head":
        for b in picks:
            ref = util_ref.oracle_detect(d, w, [frames[b]], rect, trv, p2)
            _assert_dets([pp.VoxelNet._to_dict(dets[b], int(n[b]), 0)], ref["dets"])
    # back to the default arithmetic with in-range weights: the same engine serves them on the split path again
    eng.set_gemm_precision("split_f16")
    w_ok = pp.weights.init_weights(d, seed=23)
    eng.load_weights(w_ok)
    dets, n = eng.detect(frames, R, T, on_numeric="raise")
    assert eng.gemm_precision() == "split_f16" and int(n.sum()) > 0
    eng.close()


def test_predict_stage_call_reports_non_finite_head_maps(pp, hip_lib):
    """pp_predict on caller-supplied head maps with a NaN logit: PP_ERR_NUMERIC, not NaN boxes."""
    eng = pp.Engine(pp.config.pedestrian_d435i_config(1), max_batch=1, max_points_per_frame=4096)
    d = eng.d
    rng = np.random.default_rng(3)
    H, W, k = d.head_h, d.head_w, d.num_anchor_per_loc
    box = rng.normal(0, 0.1, (1, H, W, k * 7)).astype(np.float32)
    cls = rng.normal(-2, 1, (1, H, W, k)).astype(np.float32)
    dr = rng.normal(0, 1, (1, H, W, k * 2)).astype(np.float32)
    mask = np.ones((1, d.num_anchors), np.uint8)
    rect, trv, _ = pp.synth.default_calib()
    dets, n = eng.predict(box, cls, dr, mask, rect[None], trv[None])
    assert n[0] > 0
    cls[0, 3, 5, 1] = np.nan
    with pytest.raises(pp.NumericError):
        eng.predict(box, cls, dr, mask, rect[None], trv[None])
    cls[0, 3, 5, 1] = 0.0
    box[0, 0, 0, 2] = np.inf              # a box code of an anchor that may or may not be selected: only flagged if it is
    cls[0, 0, 0, 0] = 9.0                 # ... so make it the top candidate
    with pytest.raises(pp.NumericError):
        eng.predict(box, cls, dr, mask, rect[None], trv[None])
    eng.close()


@pytest.mark.parametrize("order", ["random", "ascending"])
def test_predict_more_candidates_than_the_lds_buffer(pp, hip_lib, order):
    """107 136 anchors per frame (KITTI-shaped grid), every one a candidate: nine times what k_postprocess keeps in LDS.
    "random": the bound from the first 12 288 candidates leaves a few hundred keys for the in-LDS select (round 4);
    "ascending": logits grow with the anchor index, the bound keeps nearly everything and the select falls back to
    re-reading the head map per pass.  Both against the numpy oracle (model/voxelnet.py:1105-1379)."""
    cfg = pp.config.kitti_shaped_config(1, num_class=2)
    eng = pp.Engine(cfg, max_batch=1, max_points_per_frame=4096)
    d = eng.d
    A, H, W, k = d.num_anchors, d.head_h, d.head_w, d.num_anchor_per_loc
    assert A > 8 * 12288
    rng = np.random.default_rng(11)
    box = (rng.standard_normal((1, H, W, k * 7)) * 0.2).astype(np.float32)
    dr = rng.standard_normal((1, H, W, k * 2)).astype(np.float32)
    if order == "random":
        cls = (rng.standard_normal((1, H, W, k * d.num_class)) * 1.5 - 1.0).astype(np.float32)
    else:   # anchor a's best class logit = -6 + 8 a / A (+ a little noise below the step between neighbours' classes)
        base = -6.0 + 8.0 * np.arange(A, dtype=np.float64) / A
        cls = np.stack([base, base - 1.0], axis=1).astype(np.float32).reshape(1, H, W, k * d.num_class)
    mask = np.ones((1, A), np.uint8)
    rect, trv, _ = pp.synth.default_calib()
    dets, n = eng.predict(box, cls, dr, mask, rect[None], trv[None])
    ex = (None, None, None, rect[None], trv[None], None, eng.anchors[None], mask, np.array([0]), None)
    ref = rn.predict(ex, {"box_preds": box, "cls_preds": cls, "dir_cls_preds": dr}, d.nms_dict())[0]
    assert ref["scores"] is not None and n[0] == len(ref["scores"]) > 0
    np.testing.assert_allclose(dets[0]["score"][:n[0]], ref["scores"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dets[0]["box3d_lidar"][:n[0]], ref["box3d_lidar"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(dets[0]["label"][:n[0]], ref["label_preds"])
    eng.close()


def test_float32_fallback_on_the_sparse_canvas_configuration(pp, hip_lib):
    """pp_set_gemm_precision(PP_PREC_F32) on the KITTI-shaped configuration: the sparse canvas (whose first layer only the
    split-precision kernels understand) is switched off with the float16 pieces, the dense PFN + float32 kernels run the
    same resident frames, and the results match the split-precision pass within the 1e-4 bar and the oracle."""
    B, N = 2, 20000
    cfg = pp.config.kitti_shaped_config(B, num_class=2)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=N)
    d = eng.d
    w = pp.weights.init_weights(d, seed=5)
    eng.load_weights(w)
    frames = [pp.synth.kitti_cloud(700 + i, N) for i in range(B)]
    rect, trv, p2 = pp.synth.default_calib()
    R, T = np.stack([rect] * B), np.stack([trv] * B)
    dets, n = eng.detect(frames, R, T)
    im = eng.intermediates()
    eng.set_gemm_precision("f32")
    assert eng.gemm_precision() == "f32"
    eng.detect_async()                       # the frames are still resident
    dets32, n32 = eng.detections()
    im32 = eng.intermediates()
    assert np.array_equal(im32["coors"], im["coors"]) and np.array_equal(im32["anchors_mask"], im["anchors_mask"])
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(im32[k], im[k], rtol=0, atol=TOL)
    ref = util_ref.oracle_detect(d, w, [frames[1]], rect, trv, p2)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(im32[k][1], ref["preds"][k][0], rtol=0, atol=TOL)
    _assert_dets([pp.VoxelNet._to_dict(dets32[1], int(n32[1]), 0)], ref["dets"])
    eng.set_gemm_precision("split_f16")
    dets2, n2 = eng.detect(frames, R, T)
    assert np.array_equal(n2, n) and dets2.tobytes() == dets.tobytes(), "back on the split path: the same bits as before"
    eng.close()


# ------------------------------------------------------------------ randomised soaks (tools/fuzz_parity.py, tools/fuzz_train.py)
def _tool(name):
    import importlib
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    return importlib.import_module(name)


@pytest.mark.parametrize("seed", [1081, 1304, 10012, 10026, 10758, 10900, 10901, 60484])
def test_fuzz_parity_seeds(pp, hip_lib, seed):
    """A few cases of the randomised whole-path soak (round 4: 1 128 + 1 703 valid cases on the GPU box, none outside the
    bars): random grids / strides / widths / classes / NMS settings / batch sizes 1..32.  Seeds 1081 and 1304 hold two
    boxes whose scores agree to 1e-7 (the reference's order of equal scores is implementation-defined: the soak
    accepts a swap among equal scores and nothing else), 10758 a 32 m box (sizes are exp(t) * anchor: compared to 1e-4
    + 1e-4 of their size), 60484 two anchors of equal score at the last place of the candidate selection."""
    fz = _tool("fuzz_parity")
    print(fz.one_case(pp, util_ref, seed))


@pytest.mark.parametrize("seed", [20000, 20001, 20002, 20003, 5212, 5120])
def test_fuzz_train_seeds(pp, hip_lib, seed):
    """A few cases of the randomised training-step soak (round 4: 1 026 cases in both forward modes, none unexplained):
    losses, every gradient against torch autograd, a bit-identical second pass.  Seeds 5212 and 5120 are the documented
    hard kind: a pre-ReLU value of 1e-7 (float64) in one layer, so the gradients agree with the plain float32 / float64
    graphs only if the kernels' round-off happens to put that element on the same side (with the round's first kernels
    neither did: 2.9e-2 / 8.1e-2 off; since the BatchNorm finalise adds its partial rows in another order 5212 does,
    5120 still does not).  Whatever the side, every case
    here must agree to 1e-4 with the float64 graph that takes the step's OWN ReLU / max decisions
    (pp_train_fetch_decisions) -- the soak's criterion for such cases, applied to all six."""
    from oracle import train_ref
    ft = _tool("fuzz_train")
    res = ft.one_case(pp, util_ref, train_ref, _tool("fuzz_parity"), seed, check_decisions=True)
    print(res)
