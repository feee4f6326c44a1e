"""Training step (BASELINE.json configs[4]) on the GPU: pp_train_step (training-mode forward, loss,
backward) against torch autograd over the CPU restatement (oracle/train_ref.py, parity unpinned: TF absent), the
BatchNorm moving statistics, the optimizer step on the flat buffers and the two-rank gradient exchange."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import train_ref
import util_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(pp, cfg, frames, seed, npos=40):
    d = pp.config.Derived(cfg)
    rng = np.random.default_rng(seed)
    B, A = len(frames), d.num_anchors
    labels = rng.choice([-1, 0, 0, 0, 0], size=(B, A)).astype(np.int32)
    reg = np.zeros((B, A, 7), np.float32)
    for b in range(B):
        pos = rng.choice(A, npos if b == 0 else npos // 3, replace=False)
        labels[b, pos] = 1
        reg[b, pos] = rng.normal(0, 0.4, (len(pos), 7)).astype(np.float32)
    return d, labels, reg


def _rel_errors(got, want):
    """Per tensor: max |got - want| / max |want| and ||got - want|| / ||want||; returns the worst of each."""
    worst_max, worst_l2 = ("", 0.0), ("", 0.0)
    for name, g in want.items():
        assert got[name].shape == g.shape, name
        dmax = float(np.abs(got[name] - g).max()) / max(float(np.abs(g).max()), 1e-12)
        dl2 = float(np.linalg.norm((got[name] - g).ravel())) / max(float(np.linalg.norm(g.ravel())), 1e-30)
        if dmax > worst_max[1]:
            worst_max = (name, dmax)
        if dl2 > worst_l2[1]:
            worst_l2 = (name, dl2)
    return worst_max, worst_l2


def _variant(pp, name, B):
    import copy
    cfg = pp.config.tiny_config(B)
    s = cfg["model"]["second"]
    if name == "deep":
        s["rpn"].update(layer_nums=[3, 5, 5])
    elif name == "wide":
        s["rpn"].update(num_filters=[64, 128, 256], num_upsample_filters=[128, 128, 128])
        s["voxel_feature_extractor"]["num_filters"] = 128
    elif name == "two-classes":          # the training branch is config-driven (model/voxelnet.py:74-155, :461-512)
        s["num_class"] = 2
    elif name == "three-classes-no-dir":
        s["num_class"] = 3
        s["use_direction_classifier"] = False
    elif name == "T50-F4-dist":
        s["voxel_generator"]["max_number_of_points_per_voxel"] = 50
        s["num_point_features"] = 4
        cfg["eval_input_reader"]["num_point_features"] = 4
        s["voxel_feature_extractor"]["with_distance"] = True
    return copy.deepcopy(cfg)


@pytest.mark.parametrize("name", ["tiny", "deep", "wide", "T50-F4-dist", "two-classes", "three-classes-no-dir"])
def test_gradients_match_autograd_small_grids(pp, hip_lib, name):
    """Every trainable tensor's gradient against torch autograd over the restated network (float32), on a 20x16
    grid where float32 round-off stays small: max |diff| <= 1e-4 of the tensor's largest gradient (measured ~5e-6).
    Variants: the reference's layer counts [3,5,5]; its channel widths 64/128/256 + 128-channel upsampling and
    PFN; 50 points per pillar with 4 point features and the distance feature."""
    B = 2
    cfg = _variant(pp, name, B)
    d = pp.config.Derived(cfg)
    rng = np.random.default_rng(4)
    F = d.num_point_features
    frames = []
    for n in (900, 400):
        xyz = rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3))
        frames.append(np.concatenate([xyz, rng.uniform(0, 1, (n, F - 3))], axis=1).astype(np.float32))
    d, labels, reg = _problem(pp, cfg, frames, 11)
    if d.num_class > 1:                                     # every class among the positives
        lab_rng = np.random.default_rng(12)
        labels[labels > 0] = lab_rng.integers(1, d.num_class + 1, int((labels > 0).sum()))
    w = pp.weights.init_weights(d, seed=21)
    tr = pp.Trainer(cfg, w, max_batch=B, max_points_per_frame=4096)
    out = tr.forward_backward(frames, labels, reg)
    g1 = tr.grads.cpu().numpy().copy()
    rect, trv, p2 = pp.synth.default_calib()
    ex, _ = util_ref.oracle_example(d, frames, rect, trv, p2)
    vals, grads, stats, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0])
    for k in ("loss", "loc_loss_reduced", "cls_loss_reduced", "dir_loss_reduced", "cls_pos_loss", "cls_neg_loss"):
        assert abs(out[k] - vals[k]) <= 1e-5 * max(1.0, abs(vals[k])), (k, out[k], vals[k])
    if not d.use_direction_classifier:
        assert out["dir_loss_reduced"] == 0.0 and "rpn/conv_dir_cls/kernel" not in tr.gradients()
    assert out["num_positives"] == vals["num_positives"]
    (wn, wmax), _ = _rel_errors(tr.gradients(), grads)
    print(f"{name}: {tr.params.numel()} trainable parameters, worst relative gradient error {wmax:.2e} ({wn})")
    assert wmax <= 1e-4, (wn, wmax)
    # the same against the float64 graph that takes the step's own ReLU / max decisions (pp_train_fetch_decisions)
    import torch
    _, g64f, _, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0], dtype=torch.float64,
                                            forced=util_ref.forced_decisions(tr, ex))
    (fn_, fmax), _ = _rel_errors(tr.gradients(), g64f)
    print(f"{name}: against float64 with the step's decisions {fmax:.2e} ({fn_})")
    assert fmax <= 1e-4, (fn_, fmax)
    # a second pass over the same batch: bit-identical gradients (every reduction adds in a fixed order)
    out2 = tr.forward_backward(frames, labels, reg)
    assert out2["loss"] == out["loss"] and np.array_equal(tr.grads.cpu().numpy(), g1)
    tr.close()


def test_large_batch_paths_match_the_two_frame_step(pp, hip_lib):
    """The kernels a full-chip batch selects -- 64 x 64 product tiles with 32-wide chunks, weight- and input-gradient
    products paired in one launch, a workgroup per output in the column sums and the BatchNorm finalise (> 1 000 partial
    rows), the persistent reductions capped at one resident round of workgroups -- against the two-frame step the
    autograd tests pin, through a size-independent property: a batch made of 256 copies of a 2-frame batch has the
    2-frame batch's BatchNorm statistics and (the losses being batch means) its gradients.  Bar: 1e-4 of a tensor's
    largest gradient (different summation orders over 256 x the rows; measured 3.4e-6)."""
    cfg2 = _variant(pp, "wide", 2)
    d = pp.config.Derived(cfg2)
    rng = np.random.default_rng(5)
    frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (900, 400)]
    d, labels, reg = _problem(pp, cfg2, frames, 13)
    w = pp.weights.init_weights(d, seed=23)
    tr = pp.Trainer(cfg2, w, max_batch=2, max_points_per_frame=4096)
    out2 = tr.forward_backward(frames, labels, reg)
    g2 = tr.gradients()
    tr.close()
    copies = 256
    B = 2 * copies
    cfgB = _variant(pp, "wide", B)
    trb = pp.Trainer(cfgB, w, max_batch=B, max_points_per_frame=4096)
    outb = trb.forward_backward(frames * copies, np.tile(labels, (copies, 1)), np.tile(reg, (copies, 1, 1)))
    for k in ("loss", "loc_loss_reduced", "cls_loss_reduced", "dir_loss_reduced"):
        assert abs(outb[k] - out2[k]) <= 2e-5 * max(1.0, abs(out2[k])), (k, outb[k], out2[k])
    (wn, wmax), (ln, l2) = _rel_errors(trb.gradients(), g2)
    print(f"B={B} (256 copies) vs B=2: worst relative gradient difference {wmax:.2e} ({wn}), L2 {l2:.2e} ({ln})")
    assert wmax <= 1e-4, (wn, wmax)
    # and bit-reproducible at this size too
    gb = trb.grads.cpu().numpy().copy()
    outb2 = trb.forward_backward(frames * copies, np.tile(labels, (copies, 1)), np.tile(reg, (copies, 1, 1)))
    assert outb2["loss"] == outb["loss"] and np.array_equal(trb.grads.cpu().numpy(), gb)
    trb.close()


def test_gradients_shipped_config_batch2(pp, hip_lib):
    """cfg-A at B=2 (the reference's training batch, configs/train.yaml:62; 1.1 M trainable parameters).  With the
    synthetic random-initialised weights this problem is ill-conditioned in float32: torch's OWN float32 autograd
    differs from its float64 autograd by ~1e-2 of a tensor's largest gradient (relative L2 ~5e-3), the BatchNorm
    backward's `g - mean(g) - zhat * mean(g * zhat)` cancelling most of g over 16 stacked layers.  The yardstick is
    therefore the float64 graph, and the bar "as accurate as float32 autograd": the kernels' worst error must stay
    within 3x the float32 restatement's worst error (+1e-3), in the max norm and in the L2 norm; losses to 1e-6."""
    import torch
    B = 2
    cfg = pp.config.pedestrian_d435i_config(B)
    frames = [pp.synth.d435i_cloud(30 + i, 16384) for i in range(B)]
    d, labels, reg = _problem(pp, cfg, frames, 11)
    w = pp.weights.init_weights(d, seed=21)
    tr = pp.Trainer(cfg, w, max_batch=B, max_points_per_frame=16384)
    assert 1.0e6 < tr.params.numel() < 1.2e6
    out = tr.forward_backward(frames, labels, reg)
    rect, trv, p2 = pp.synth.default_calib()
    ex, _ = util_ref.oracle_example(d, frames, rect, trv, p2)
    v64, g64, _, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0], dtype=torch.float64)
    v32, g32, _, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0])
    for k in ("loss", "loc_loss_reduced", "cls_loss_reduced", "dir_loss_reduced"):
        assert abs(out[k] - v64[k]) <= 2e-6 * max(1.0, abs(v64[k])), (k, out[k], v64[k])
    (hn, hmax), (hl, hl2) = _rel_errors(tr.gradients(), g64)
    (tn, tmax), (tl, tl2) = _rel_errors(g32, g64)
    print(f"cfg-A B=2 vs float64 autograd: kernels max-norm {hmax:.2e} ({hn}), L2 {hl2:.2e} ({hl}); "
          f"torch float32 max-norm {tmax:.2e} ({tn}), L2 {tl2:.2e} ({tl})")
    assert hmax <= 3.0 * tmax + 1e-3, (hn, hmax, tmax)
    assert hl2 <= 3.0 * tl2 + 1e-3, (hl, hl2, tl2)
    # Round 4: where that 1e-2 comes from, and the sharp comparison.  Of the 9.5 M pre-ReLU values of this step a handful
    # lie within 1e-6 of zero; ReLU (and the PFN's max) are not differentiable there, and whichever side an
    # implementation's round-off puts such an element, the gradients change by its whole contribution.  Given the SAME
    # decisions the gradient is a smooth function: the float64 graph that takes the step's own ReLU masks and PFN winners
    # (pp_train_fetch_decisions) must agree with the kernels to 1e-4 of every tensor's largest gradient.
    forced = util_ref.forced_decisions(tr, ex)
    _, g64f, _, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0], dtype=torch.float64, forced=forced)
    (fn_, fmax), (fl, fl2) = _rel_errors(tr.gradients(), g64f)
    print(f"cfg-A B=2 vs float64 autograd taking the step's own ReLU / max decisions: max-norm {fmax:.2e} ({fn_}), L2 {fl2:.2e} ({fl})")
    assert fmax <= 1e-4, (fn_, fmax)
    tr.close()


def test_batchnorm_moving_statistics_update(pp, hip_lib):
    """moving = moving * momentum + batch * (1 - momentum): momentum 0.01 and the biased variance for the PFN's
    BatchNorm (model/pointpillars.py:109, rank-3 input), 0.99 and the unbiased variance for the RPN's fused ones."""
    B = 2
    cfg = pp.config.tiny_config(B)
    rng = np.random.default_rng(8)
    frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (800, 500)]
    d, labels, reg = _problem(pp, cfg, frames, 3, npos=12)
    w = pp.weights.init_weights(d, seed=5)
    tr = pp.Trainer(cfg, w, max_batch=B, max_points_per_frame=4096)
    tr.forward_backward(frames, labels, reg)
    after = tr.weights()
    rect, trv, p2 = pp.synth.default_calib()
    ex, _ = util_ref.oracle_example(d, frames, rect, trv, p2)
    _, _, stats, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0])
    rows = {"rpn/block1": B * 16 * 20, "rpn/block2": B * 8 * 10, "rpn/block3": B * 4 * 5, "rpn/deconv": B * 16 * 20}
    for pre, (mean, var) in stats.items():
        mom = 0.01 if pre == "pfn/bn" else 0.99
        want_mean = w[pre + "/moving_mean"] * mom + mean * (1 - mom)
        np.testing.assert_allclose(after[pre + "/moving_mean"], want_mean, rtol=2e-4, atol=2e-5, err_msg=pre)
        got_var = (after[pre + "/moving_variance"] - w[pre + "/moving_variance"] * mom) / (1 - mom)   # the batch term
        if pre == "pfn/bn":
            np.testing.assert_allclose(got_var, var, rtol=2e-3, atol=1e-5, err_msg=pre)
        else:
            n = next(v for k, v in rows.items() if pre.startswith(k))
            np.testing.assert_allclose(got_var, var * (n / (n - 1.0)), rtol=2e-3, atol=1e-5, err_msg=pre)   # Bessel
    tr.close()


def test_optimizer_steps_reduce_the_loss_and_export_to_inference(pp, hip_lib):
    B = 2
    cfg = pp.config.tiny_config(B)
    rng = np.random.default_rng(17)
    frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (900, 600)]
    d, labels, reg = _problem(pp, cfg, frames, 9, npos=10)
    tr = pp.Trainer(cfg, pp.weights.init_weights(d, seed=2), max_batch=B, max_points_per_frame=4096, learning_rate=2e-3,
                    weight_decay=1e-4)
    losses = [tr.step(frames, labels, reg)["loss"] for _ in range(12)]
    assert losses[-1] < 0.7 * losses[0], losses
    assert tr.optimizer.iterations == 12
    # every upload flips the handle's input buffer: one captured hipGraph per buffer, every later step a replay
    # (round 2 kept ONE graph keyed on the buffer and re-captured ~270 nodes on every optimizer step)
    captures, replays = tr.engine.train_graph_stats()
    assert captures <= 2 and replays == 12, (captures, replays)
    # a batch staged in page-locked memory (three DMA transfers) gives the bits of the pageable path
    a = tr.forward_backward(frames, labels, reg)
    ga = tr.grads.cpu().numpy().copy()
    st = tr.stage(frames, labels, reg)
    b = tr.forward_backward(st)
    assert a["loss"] == b["loss"] and np.array_equal(tr.grads.cpu().numpy(), ga)
    # the loader's hand-over: the NEXT batch's points uploaded while this step runs (pp_train_step_async / _wait);
    # the prefetched batch is then consumed without a second upload, and nothing changes in the numbers
    st2 = tr.stage(frames[::-1], labels[::-1], reg[::-1])
    c0 = tr.forward_backward(st2)
    gc = tr.grads.cpu().numpy().copy()
    b1 = tr.forward_backward(st, prefetch=st2)
    assert b1["loss"] == b["loss"] and np.array_equal(tr.grads.cpu().numpy(), ga)
    assert tr._prefetched is st2
    c1 = tr.forward_backward(st2, prefetch=st)
    assert c1["loss"] == c0["loss"] and np.array_equal(tr.grads.cpu().numpy(), gc)
    b2 = tr.forward_backward(st)
    assert b2["loss"] == b["loss"] and np.array_equal(tr.grads.cpu().numpy(), ga)
    st.close()
    st2.close()
    # the trained tensors (and the updated moving statistics) load into an inference engine
    w = tr.weights()
    pp.weights.check_weights(d, w)
    eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=4096)
    eng.load_weights(w)
    dets, n = eng.detect(frames)
    assert n.shape == (B,)
    eng.close()
    tr.close()


_WORKER = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
import pp_amd as pp
dist.init_process_group(backend="gloo")
r, n = dist.get_rank(), dist.get_world_size()
B = 2
cfg = pp.config.tiny_config(B)
d = pp.config.Derived(cfg)
rng = np.random.default_rng(100 + r)                 # every rank its own frames and targets
frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (700, 3)).astype(np.float32) for _ in range(B)]
labels = rng.choice([-1, 0, 0, 0], size=(B, d.num_anchors)).astype(np.int32)
reg = np.zeros((B, d.num_anchors, 7), np.float32)
pos = rng.choice(d.num_anchors, 8, replace=False); labels[0, pos] = 1; reg[0, pos] = rng.normal(0, 0.4, (8, 7))
tr = pp.Trainer(cfg, pp.weights.init_weights(d, seed=2), max_batch=B, max_points_per_frame=4096, learning_rate=1e-3)
tr.forward_backward(frames, labels, reg)
own = tr.grads.clone()
gathered = [torch.zeros_like(own.cpu()) for _ in range(n)]
dist.all_gather(gathered, own.cpu())
tr.step(frames, labels, reg, dist)                   # all-reduce (mean) + AdamW
mean = sum(gathered) / n
assert torch.allclose(tr.grads.cpu(), mean, rtol=1e-6, atol=1e-9), "the flat buffer must hold the rank mean"
p = tr.params.cpu()
ps = [torch.zeros_like(p) for _ in range(n)]
dist.all_gather(ps, p)
assert all(torch.equal(ps[0], q) for q in ps), "replicas must stay identical after the step"
print("rank", r, "ok")
tr.close()
dist.destroy_process_group()
"""


def test_two_rank_data_parallel_step(hip_lib, tmp_path):
    """Two ranks (sharing this box's GPU; gloo carries the collective here, RCCL on a multi-GPU node): different
    batches per rank, ONE all-reduce of the flat gradient buffer, identical parameters afterwards."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_voxelnet_training_call_surface(pp, hip_lib):
    """VoxelNet(config, writer, training=True)(voxels, num_points, coors, anchors, labels, reg_targets): the
    reference's call with the dataloader's padded tensors gives the same loss and gradients as the raw-cloud path."""
    B = 2
    cfg = pp.config.tiny_config(B)
    rng = np.random.default_rng(23)
    frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (800, 350)]
    d, labels, reg = _problem(pp, cfg, frames, 5, npos=10)
    w = pp.weights.init_weights(d, seed=6)
    net = pp.VoxelNet(cfg, None, training=True, max_batch=B, max_points_per_frame=4096)
    with pytest.raises(RuntimeError):
        net.train_step(frames, labels, reg)
    net.load_weights(w)
    a = net.train_step(frames, labels, reg, apply=False)
    ga = net.trainer.grads.cpu().numpy().copy()
    rect, trv, p2 = pp.synth.default_calib()
    ex, _ = util_ref.oracle_example(d, frames, rect, trv, p2)
    b = net(ex[0], ex[1], ex[2], ex[6], labels, reg)
    assert a["loss"] == b["loss"] and np.array_equal(net.trainer.grads.cpu().numpy(), ga)
    before = net.trainer.params.cpu().numpy().copy()
    net.apply_gradients()
    assert np.abs(net.trainer.params.cpu().numpy() - before).max() > 0
    with pytest.raises(ValueError):
        net(ex[0], ex[1], ex[2], ex[6])
    # the two halves of pp_train_step: a second launch before the wait, or a wait without a launch, is an error
    tr = net.trainer
    eng = tr.engine
    eng.upload(frames)
    eng.train_step_async(tr.params.data_ptr(), tr.grads.data_ptr(), tr.state.data_ptr(), labels, reg)
    with pytest.raises(RuntimeError):
        eng.train_step_async(tr.params.data_ptr(), tr.grads.data_ptr(), tr.state.data_ptr(), labels, reg)
    assert eng.train_step_wait()["loss"] > 0
    with pytest.raises(RuntimeError):
        eng.train_step_wait()
    assert eng.stream_ptr() != 0
    net.trainer.close()


def test_fused_forward_kernels_on_small_grids(hip_lib):
    """The fused training-forward launches (k_sep_u<..., TR = 1>: depthwise + product + statistics of a separable layer;
    TR = 2: the transposed convolutions' and the heads' forward products) only run from 32 768 output rows on by default,
    so the autograd tests above (640-row maps) time the separate kernels.  One child process with PP_TRAIN_FUSED_MIN=0 --
    the switch is read once per process -- runs the same six autograd comparisons through the fused launches: ragged
    tiles (640 / 160 / 40 rows), every tile width (32 / 64 / 128 output channels), stride 1 and 2, the NaN padding header
    on 20 x 16 maps, head bias."""
    env = dict(os.environ)
    env["PP_TRAIN_FUSED_MIN"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_train.py"), "-k", "small_grids and not fused_forward"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "6 passed" in r.stdout, r.stdout[-1000:]


def test_decisions_tap_surface(pp, hip_lib):
    """pp_train_fetch_decisions: state and argument errors, shapes, and that the masks are the ones the step used --
    an element the mask lets through has a positive activation in the oracle's float64 run unless it lies within
    round-off of zero."""
    import ctypes
    import torch
    B = 2
    cfg = pp.config.tiny_config(B)
    d = pp.config.Derived(cfg)
    rng = np.random.default_rng(6)
    frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (700, 300)]
    d, labels, reg = _problem(pp, cfg, frames, 5, npos=12)
    w = pp.weights.init_weights(d, seed=9)
    tr = pp.Trainer(cfg, w, max_batch=B, max_points_per_frame=4096)
    eng = tr.engine
    n = ctypes.c_int64(0)
    assert eng._lib.pp_train_fetch_decisions(eng._h, 0, None, 0, ctypes.byref(n)) == 2      # PP_ERR_STATE: no step yet
    tr.forward_backward(frames, labels, reg)
    assert eng._lib.pp_train_fetch_decisions(eng._h, 0, None, 0, None) == 1                  # PP_ERR_ARG: count is NULL
    assert eng._lib.pp_train_fetch_decisions(eng._h, 99, None, 0, ctypes.byref(n)) == 1      # no such layer
    assert eng._lib.pp_train_fetch_decisions(eng._h, 0, None, 0, ctypes.byref(n)) == 0 and n.value == B * 16 * 20 * 32
    small = np.zeros(8, np.uint8)
    assert eng._lib.pp_train_fetch_decisions(eng._h, 0, small.ctypes.data_as(ctypes.c_void_p), 8, ctypes.byref(n)) == 1
    dec = tr.decisions()
    assert dec["pfn"].shape == (B, d.max_voxels, d.pfn_filters) and dec["pfn"].dtype == np.int32
    assert dec["rpn/block1/0/bn"].shape == (B, 16 * 20, 32) and dec["rpn/deconv3/bn"].shape == (B, 4 * 5, 4, 4, 32)
    rect, trv, p2 = pp.synth.default_calib()
    ex, _ = util_ref.oracle_example(d, frames, rect, trv, p2)
    rec, marg = {}, {}
    train_ref.training_step(d, w, ex, labels, reg, ex[6][0], dtype=torch.float64, record=rec, margins=marg)
    forced = util_ref.forced_decisions(tr, ex)
    differ = {k: int((np.asarray(forced[k]) != np.asarray(rec[k]).reshape(np.asarray(forced[k]).shape)).sum()) for k in rec}
    total = sum(np.asarray(v).size for v in rec.values())
    print(f"decisions that differ from the float64 run: {sum(differ.values())} of {total} "
          f"({ {k: v for k, v in differ.items() if v} }); smallest margin {min(v for k, v in marg.items() if not k.startswith('#')):.1e}")
    assert sum(differ.values()) <= 5          # (a handful of elements within round-off of a kink, typically none)
    tr.close()
