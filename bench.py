#!/usr/bin/env python3
"""Headline benchmark: PointPillars inference frames/s on synthetic 16k-point d435i clouds.

    python bench.py --gpus N --steps K --warmup W
(for N > 1 the driver launches this under torch.distributed.run, one rank per GPU.)

A "step" is one pass of the whole hot path over one batch of B = 64 frames
(BASELINE.json configs[1]; the shipped reference config, SURVEY "cfg-A"): the
host-to-device copy of the batch's raw points from a page-locked staging buffer
(a DIFFERENT batch every step: a pool of distinct synthetic batches is cycled,
SURVEY section 8d's protocol and train.py:748's per-frame hand-over) -> voxelise ->
PFN + scatter -> anchor mask -> backbone + heads -> top-k / decode / NMS ->
detections copied to pinned host memory and read by the host.  `--inflight`
engine handles (own stream + workspaces) take turns, so the copy of one batch
overlaps the kernels of the others; a handle is synchronised and its detections
are fetched before it is given its next batch.  Frames are independent, so with N
GPUs every rank processes its own frames (weak scaling, no collective on the data
path); the timed region is bracketed by a barrier + device sync and the MAX over
ranks is taken.  `detail.resident_fps` is the same loop without the per-step
upload (points resident in HBM).

The JSON line also carries
  roofline      the dominant kernel (largest share of GPU time with ONE batch in
                flight), its average launch duration measured in this process
                with a HIP start/stop event pair carried by each launch
                (hipExtLaunchKernelGGL: the kernel's own execution interval, what
                rocprofv3's kernel trace reports), and algorithmic flops (or
                bytes) per launch / that duration against the gfx950 peak
                (`frac`); `frac_overlapped` is the same with --inflight batches
                sharing the chip;
  cpu_baseline  the CPU oracle (a faithful restatement of the reference's
                numpy/TF path: C voxeliser + numpy PFN + torch-CPU backbone +
                numpy predict) timed on this host on a bounded sample, rank 0,
                N == 1 only.  The literal TF-2.2 binary cannot run here
                (DESIGN.md "CPU baseline").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense 16-bit MFMA (v_mfma_f32_32x32x16_f16 / _bf16, 32 cyc/SIMD)
# 16-bit products per float32 product in the split-precision kernels: 3 (two float16 pieces per operand: hi*hi,
# hi*mid, mid*hi; the library's default build) or 6 (PP_SPLIT_MODE=0 build: three bfloat16 pieces)
SPLIT_TERMS = 6 if "bf16x3" in os.environ.get("PP_HIP_LIB", "") else 3
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def sparse_canvas(d):
    """Does the engine run this grid with the sparse canvas (pp_finalize_weights' rule: a large, mostly empty BEV
    grid -- the PFN writes only the cells that hold a pillar and block1.0 reads only those)?"""
    cells = d.ny * d.nx
    return os.environ.get("PP_DENSE_CANVAS", "") != "1" and cells >= 32768 and 4 * d.max_voxels <= cells


def layer_bytes(d, batch, heads_fused=False, n_pillars=None):
    """Algorithmic HBM bytes per launch of every backbone layer: the input map read once, the output map
    written once (fp32 activations), the fused head map written (first branch) or read + written.  With the sparse
    canvas (n_pillars given) block1.0 reads the occupied cells' rows and the cell map, not the dense pseudo-image."""
    out = {}
    h, w, cin = d.ny, d.nx, d.pfn_filters
    for b in range(3):
        cout = d.num_filters[b]
        for j in range(d.layer_nums[b] + 1):
            s = d.layer_strides[b] if j == 0 else 1
            ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            in_floats = h * w * cin
            if b == 0 and j == 0 and n_pillars is not None and sparse_canvas(d):
                in_floats = min(n_pillars, h * w) * cin + d.nz * h * w          # occupied rows + the cell -> pillar map
            out[f"block{b + 1}.{j}"] = 4.0 * batch * (in_floats + ho * wo * cout)
            h, w, cin = ho, wo, cout
        k = d.upsample_strides[b]
        if heads_fused:   # the concat slice is never written: input map + the 32-column head map (write, or read + write)
            out[f"deconv{b + 1}"] = 4.0 * batch * (h * w * cin + d.head_h * d.head_w * 32 * (1 if b == 0 else 2))
        else:
            out[f"deconv{b + 1}"] = 4.0 * batch * (h * w * cin + h * k * w * k * d.num_upsample_filters[b])
    if not heads_fused:
        out["heads"] = 4.0 * batch * d.head_h * d.head_w * (d.concat_channels + 32)
    return out


def layer_flops(d, batch, heads_fused=False):
    """Algorithmic FLOPs per launch of every backbone layer:
    depthwise 2*9*Cin + pointwise 2*Cin*Cout per output pixel; deconv 2*Cin*k*k*Cout per
    input pixel; heads 2*384*(14+2+4) per pixel (credited to the deconv launches, branch by
    branch, when the heads are fused into their epilogues)."""
    out = {}
    h, w, cin = d.ny, d.nx, d.pfn_filters
    for b in range(3):
        cout = d.num_filters[b]
        for j in range(d.layer_nums[b] + 1):
            s = d.layer_strides[b] if j == 0 else 1
            h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            out[f"block{b + 1}.{j}"] = 2.0 * batch * h * w * (9 * cin + cin * cout)
            cin = cout
        k = d.upsample_strides[b]
        out[f"deconv{b + 1}"] = 2.0 * batch * h * w * cin * k * k * d.num_upsample_filters[b]
    n_head = d.num_anchor_per_loc * (7 + d.num_class + 2)
    if heads_fused:
        for b in range(3):
            out[f"deconv{b + 1}"] += 2.0 * batch * d.head_h * d.head_w * d.num_upsample_filters[b] * n_head
    else:
        out["heads"] = 2.0 * batch * d.head_h * d.head_w * d.concat_channels * n_head
    return out


def stage_bytes(d, batch, n_points, n_pillars):
    """Algorithmic HBM bytes per launch of the non-GEMM kernels."""
    F, C = d.num_point_features, d.pfn_filters
    return {
        "k_cell_first": batch * (4 * F * n_points + 4 * n_points),
        "k_voxel_frame": batch * (4 * n_points * 2 + 16 * n_pillars),
        # dense canvas: every cell is written (zeros included); sparse canvas: the occupied cells only
        "k_pfn_canvas": batch * (4 * F * n_points + 16 * n_pillars +
                                 4 * (min(n_pillars, d.ny * d.nx) if sparse_canvas(d) else d.ny * d.nx) * C),
        "k_occ_rowscan": batch * (4 * d.ny * d.nx * d.nz + 4 * d.ny * d.nx),
        "k_colscan": batch * (8 * d.ny * d.nx),
        "k_anchor_lookup": batch * (4 * d.ny * d.nx + d.num_anchors * 17),
        "k_anchor_mask_frame": batch * (4 * d.ny * d.nx * d.nz + d.num_anchors * 17),   # the three in one (few frames)
        "k_sort_points": batch * (4 * n_points + 2 * 4 * F * n_points),
        "k_postprocess": batch * (d.num_anchors * 5),
    }


def stage_key(sym):
    """stage_bytes key of a non-GEMM kernel symbol."""
    if sym.startswith("k_pfn_canvas"):
        return "k_pfn_canvas"
    return sym.split("(")[0].split("<")[0]


def cpu_baseline(pp, d, weights, frames, calib, budget_s=28.0):
    """The oracle end to end on the host cores (checker code used as the CPU baseline).  SURVEY section 8d asks for a
    single-thread number beside the many-thread one: the torch-CPU backbone is timed at 1, 8, 16 and 32 threads, a
    bounded share of the budget each; `value` is the best of the sweep, `single_thread` the 1-thread rate.  The rank's
    NUMA pin is lifted for this leg (the baseline may use the whole host).  More threads than the sweep's 32 are not
    timed: a GPU box hands a job a CPU share far below its `os.cpu_count()` (16 CPUs beside one GPU on this pool), and
    round 3's "all 128 threads" point (0.08 frames/s) measured 128 spinning OpenMP threads on that share, not the
    batch-1 convolutions."""
    import torch
    import util_ref
    rect, trv, p2 = calib
    restored = pp.frame_shard.restore_affinity()
    all_threads = torch.get_num_threads()
    usable = len(os.sched_getaffinity(0))
    counts = sorted({c for c in (1, 8, 16, 32) if c <= max(1, min(all_threads, usable))})
    sweep = {}
    share = budget_s / len(counts)
    for cores in counts:
        util_ref.oracle_detect(d, weights, frames[:1], rect, trv, p2, num_threads=cores)  # warm-up
        done, t0, lat = 0, time.perf_counter(), []
        while done < len(frames) and (done == 0 or time.perf_counter() - t0 < share):
            t1 = time.perf_counter()
            util_ref.oracle_detect(d, weights, frames[done:done + 1], rect, trv, p2, num_threads=cores)
            lat.append(time.perf_counter() - t1)
            done += 1
        el = time.perf_counter() - t0
        sweep[cores] = {"frames_per_s": done / el, "p50_ms_per_frame": float(np.median(lat) * 1e3), "frames": done}
    torch.set_num_threads(all_threads)
    best = max(sweep, key=lambda c: sweep[c]["frames_per_s"])
    return {"value": sweep[best]["frames_per_s"], "unit": "frames/s", "cores": best, "kind": "port",
            "p50_ms_per_frame": sweep[best]["p50_ms_per_frame"],
            "single_thread": sweep[1]["frames_per_s"], "host_threads": all_threads, "cpus_in_affinity_mask": usable,
            "numa_pin_lifted_for_this_leg": bool(restored),
            "thread_sweep": {str(c): round(v["frames_per_s"], 3) for c, v in sweep.items()},
            "sample": f"{sum(v['frames'] for v in sweep.values())} passes over the same synthetic 16k-point frames, batch 1 "
                      f"(the reference's eval batch), C voxeliser + numpy PFN + torch-CPU fp32 backbone + numpy predict, "
                      f"thread sweep {counts}: best at {best} threads"}


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources: what a committed counter pass must match to be quoted."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, "3d-object-detection-for-autonomous-navigation_amd", "csrc")
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()[:16]


def is_split_kernel(sym):
    """Does this GEMM kernel run on the bf16 matrix pipe with split operands?  k_sep_u<NT,S,WPS,PREC,OCC>: PREC."""
    if sym.startswith(("k_deconv_u", "k_deconv_r", "k_deconv_k4", "k_sep_k4", "k_sep_p")):
        return True
    if sym.startswith("k_sep_u<"):
        params = sym[sym.index("<") + 1:sym.rindex(">")].split(",")
        return len(params) > 3 and params[3].strip() == "1"
    return False


class Feeder:
    """The evaluate loop body of train.py:689-786 for `len(engines)` batches in flight: step i gives engine
    i % n the staged batch i % pool (host-to-device copy + the whole path, all asynchronous); before an engine
    gets its next batch the host waits for its previous one and fetches the detections."""

    def __init__(self, engines, stagings, upload=True):
        self.engines, self.stagings, self.upload = engines, stagings, upload
        # one engine in flight: its passes get the whole last-level cache (frame sub-ranges for maps that do not fit,
        # pp_set_cache_budget); several: off -- their working sets evict each other and the extra launches only cost
        for e in engines:
            e.set_cache_budget(256 if len(engines) == 1 else 0)
        self.busy = [False] * len(engines)
        self.i = 0
        self.n_det = 0          # detections the host has read (checksum of the consumed results)
        self.n_batches = 0
        e0 = engines[0]
        self.out = (np.zeros((e0.max_batch, e0.d.nms_post_max_size), dtype=type(e0).det_dtype()),
                    np.zeros((e0.max_batch,), dtype=np.int32))

    def _collect(self, k):
        if self.busy[k]:
            e = self.engines[k]
            e.sync()
            dets, n = e.detections(self.out)
            self.n_det += int(n[:self.stagings[0].offsets.shape[0] - 1].sum())
            self.n_batches += 1
            self.busy[k] = False

    def step(self):
        k = self.i % len(self.engines)
        e = self.engines[k]
        if self.upload:   # on the handle's copy stream, into its other input buffer: runs beside the pass in flight
            e.upload_async(self.stagings[self.i % len(self.stagings)])
        self._collect(k)
        e.detect_async()
        self.busy[k] = True
        self.i += 1

    def drain(self):
        for k in range(len(self.engines)):
            self._collect(k)


SETTLE_STEPS = 40   # untimed set-up before the W warm-up steps: graph capture for both input buffers, clocks, caches


def timed_run(feeder, steps, warmup, barrier, settle=0):
    for _ in range(settle + warmup):
        feeder.step()
    feeder.drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        feeder.step()
    feeder.drain()
    barrier()
    return time.perf_counter() - t0


def kernel_pass(engines, feeder_factory, steps):
    """Per-kernel execution times (ms) with the feeder's pattern: {tag: [samples]}."""
    for e in engines:
        e.set_profiling(True)
    samples = {}
    f = feeder_factory()
    orig = f._collect

    def collect(k):
        was = f.busy[k]
        orig(k)
        if was:
            for tag, ms in engines[k].kernel_times():
                samples.setdefault(tag, []).append(ms)
    f._collect = collect
    for _ in range(steps):
        f.step()
    f.drain()
    for e in engines:
        e.set_profiling(False)
    return samples


def summarise(samples, steps):
    """{tag: [ms]} -> per-symbol (ms per step, launches per step), per-layer mean ms, dropped-sample count.
    A launch now and then lands on a stall that is not the kernel's: samples beyond 4x the tag's median are
    left out of the averages and counted."""
    agg, per_layer, dropped = {}, {}, 0
    for tag, ms_list in samples.items():
        med = float(np.median(ms_list))
        kept = [m for m in ms_list if m <= 4.0 * med] or ms_list
        dropped += len(ms_list) - len(kept)
        mean = sum(kept) / len(kept)
        sym, _, layer = tag.partition(":")
        a = agg.setdefault(sym, [0.0, 0])
        a[0] += mean * len(ms_list)
        a[1] += len(ms_list)
        if layer:
            # (mean launch ms, symbol, launches of this layer per step: > 1 when the engine walks a layer over
            # sub-ranges of the batch's frames, pp_set_cache_budget)
            per_layer[layer] = (mean, sym, len(ms_list) / steps)
    kernel_ms = {k: v[0] / steps for k, v in agg.items()}
    launches = {k: v[1] / steps for k, v in agg.items()}
    return kernel_ms, launches, per_layer, dropped


def kernel_roofs(d, B, n_points, n_pillars, kernel_ms, launches, per_layer, heads_fused):
    """Every kernel symbol against its roofs: algorithmic bytes / flops per launch over the average launch time."""
    lf, lb = layer_flops(d, B, heads_fused), layer_bytes(d, B, heads_fused, n_pillars)
    sb = stage_bytes(d, B, n_points, n_pillars)
    out = {}
    for sym, ms in kernel_ms.items():
        t = ms / launches[sym] * 1e-3
        mine = [layer for layer, v in per_layer.items() if v[1] == sym and layer in lf]
        r = {"avg_launch_ms": ms / launches[sym], "launches_per_step": launches[sym]}
        if mine:
            # per launch: the layers' work over the launches they took (a layer walked in four sub-ranges is four launches)
            nl = sum(per_layer[n][2] for n in mine)
            fl = sum(lf[n] for n in mine) / nl
            by = sum(lb[n] for n in mine) / nl
            split = is_split_kernel(sym)
            peak = BF16_MFMA_PEAK_TFLOPS / SPLIT_TERMS if split else F32_MFMA_PEAK_TFLOPS
            r.update({"algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                      "fp32_equivalent_tflops": fl / t / 1e12, "GBps": by / t / 1e9,
                      "frac_of_mfma_roof": fl / t / 1e12 / peak, "frac_of_hbm_roof": by / t / 1e9 / HBM_PEAK_GBS,
                      "mfma_roof": (f"16-bit dense {BF16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s / {SPLIT_TERMS} products per fp32 product"
                                    if split else "f32 MFMA"), "mfma_peak": peak})
        else:
            by = sb.get(stage_key(sym))
            if by:
                r.update({"algorithmic_bytes_per_launch": by, "GBps": by / t / 1e9,
                          "frac_of_hbm_roof": by / t / 1e9 / HBM_PEAK_GBS})
        out[sym] = r
    return out


def roofline_of(sym, r):
    """The contract's roofline object for kernel `sym` from its kernel_roofs entry (binding roof = larger fraction)."""
    fh, fm = r.get("frac_of_hbm_roof", 0.0), r.get("frac_of_mfma_roof", 0.0)
    if fm > fh:
        base = {"bound": "mfma", "kernel": sym, "achieved": r["fp32_equivalent_tflops"], "peak": r["mfma_peak"],
                "unit": "TFLOP/s", "frac": fm, "traffic": None}
    else:
        base = {"bound": "hbm", "kernel": sym, "achieved": r.get("GBps", 0.0), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": fh, "traffic": None}
    base.update({k: v for k, v in r.items() if k != "mfma_peak"})
    return base


def cfgk_leg(pp, local_rank, steps=12, only_inflight=None):
    """BASELINE.json configs[2]: KITTI-shaped clouds (20k points, 0.16 m pillars, 496x432 BEV, Pedestrian+Cyclist
    as two classes), batch 32 on one GPU, same feeder (a different staged batch per step)."""
    B, N = 32, 20000
    cfg = pp.config.kitti_shaped_config(B, num_class=2)
    engines = [pp.Engine(cfg, max_batch=B, max_points_per_frame=N, device=local_rank) for _ in range(2)]
    d = engines[0].d
    w = pp.weights.init_weights(d, seed=5)
    calib = pp.synth.default_calib()
    pool = 3
    stagings = [engines[0].staging([pp.synth.kitti_cloud(1000 + j * B + i, N) for i in range(B)]) for j in range(pool)]
    for e in engines:
        e.load_weights(w)
        e.set_calib(np.stack([calib[0]] * B), np.stack([calib[1]] * B), B)
    sync = lambda: None
    out = {"workload": f"cfg-K (432x496 BEV, 0.16 m pillars, T=100, C=64, strides [2,2,2], 2 classes), B={B} x {N} pts, "
                       "upload inside the step"}
    for nfl in ((2, 1) if only_inflight is None else (only_inflight,)):
        f = Feeder(engines[:nfl], stagings)
        el = timed_run(f, steps, 3, sync)
        key = "" if nfl == 2 else "_inflight1"
        out["fps" + key] = B * steps / el
        out["ms_per_step" + key] = el / steps * 1e3
        out["mean_detections_per_frame"] = f.n_det / max(f.n_batches * B, 1)
    if only_inflight is not None:     # profiling run (rocprofv3 wraps it): one regime, no per-launch event pass
        for s in stagings:
            s.close()
        for e in engines:
            e.close()
        return out
    samples = kernel_pass(engines[:1], lambda: Feeder(engines[:1], stagings), 3)
    kernel_ms, launches, per_layer, dropped = summarise(samples, 3)
    n_pillars = float(engines[0].intermediates()["n_pillars"].mean())
    heads_fused = not any(t.endswith(":heads") for t in engines[0].layer_tags())
    roofs = kernel_roofs(d, B, N, n_pillars, kernel_ms, launches, per_layer, heads_fused)
    dom = max(kernel_ms, key=kernel_ms.get)
    out["mean_pillars_per_frame"] = n_pillars
    out["kernel_ms_per_step_inflight1"] = {k: round(v, 4) for k, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1])}
    out["roofline"] = roofline_of(dom, roofs[dom])
    out["roofline"]["traffic"] = None
    try:      # HBM traffic per launch from the committed cfg-K counter passes, quoted only for the sources it ran on
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic_cfgk.json")))
        if files:
            doc = json.load(open(files[-1]))
            ent = doc["kernels"].get(dom) or doc["kernels"].get(dom[:-1] + ",0>")   # (the symbol gained a sixth template parameter: TR = 0)
            src = {"file": os.path.basename(files[-1]), "git_head": doc.get("git_head"), "csrc_sha16": doc.get("csrc_sha16")}
            if ent is not None and doc.get("csrc_sha16") == csrc_sha16():
                out["roofline"]["traffic"] = ent.get("hbm_bytes_per_launch")
            else:
                src["stale"] = "kernel sources changed since the pass" if ent is not None else "kernel not in the pass"
            out["roofline"]["traffic_source"] = src
    except Exception as ex:
        out["roofline"]["traffic_source"] = {"error": repr(ex)}
    out["layer_ms"] = {k: round(v[0] * v[2], 4) for k, v in per_layer.items()}     # per step (all sub-range launches of the layer)
    for s in stagings:
        s.close()
    for e in engines:
        e.close()
    return out


def train_gemm_flops(d, batch):
    """Algorithmic FLOPs of the GEMM-shaped work of one training step: the forward products of the pointwise, the
    transposed-convolution and the head layers, times three (forward, input gradient, weight gradient)."""
    lf = layer_flops(d, batch, heads_fused=False)
    fwd = 0.0
    h, w, cin = d.ny, d.nx, d.pfn_filters
    for b in range(3):
        cout = d.num_filters[b]
        for j in range(d.layer_nums[b] + 1):
            s_ = d.layer_strides[b] if j == 0 else 1
            h, w = (h + 2 - 3) // s_ + 1, (w + 2 - 3) // s_ + 1
            fwd += 2.0 * batch * h * w * cin * cout
            cin = cout
        fwd += lf[f"deconv{b + 1}"]
    fwd += 2.0 * batch * d.head_h * d.head_w * d.concat_channels * 32     # the packed 32-column head GEMM
    return 3.0 * fwd


def train_leg(pp, local_rank, rank, n_gpus, dist, comm_dev, barrier, steps=20, batch=2):
    """BASELINE.json configs[4]: one optimizer step = upload + voxelise + training-mode forward + loss + backward
    (pp_train_step) + ONE all-reduce of the flat 4.4 MB gradient buffer over the ranks + AdamW, at the reference's
    training batch (2 frames per GPU, configs/train.yaml:62), cfg-A, synthetic clouds and targets."""
    import torch
    cfg = pp.config.pedestrian_d435i_config(batch)
    d = pp.config.Derived(cfg)
    tr = pp.Trainer(cfg, pp.weights.init_weights(d, seed=7), max_batch=batch, max_points_per_frame=16384, device=local_rank,
                    learning_rate=2e-4, weight_decay=1e-4)
    rng = np.random.default_rng(50 + rank)
    frames = [pp.synth.d435i_cloud(5000 + rank * batch + i, 16384) for i in range(batch)]
    labels = rng.choice([-1, 0, 0, 0, 0], size=(batch, d.num_anchors)).astype(np.int32)
    reg = np.zeros((batch, d.num_anchors, 7), np.float32)
    for b in range(batch):
        pos = rng.choice(d.num_anchors, 30, replace=False)
        labels[b, pos] = 1
        reg[b, pos] = rng.normal(0, 0.4, (30, 7)).astype(np.float32)
    # two batches staged in page-locked memory take turns (the loader's hand-over, train.py:265-304: a different batch
    # every step); the host-to-device copies of points, labels and regression targets are inside every timed step -- the
    # targets travel behind the forward pass, the NEXT step's points (prefetch=) beside this step's kernels
    staged = [tr.stage(frames, labels, reg), tr.stage(frames[::-1], labels[::-1], reg[::-1])]
    for i in range(4):
        out = tr.step(staged[i % 2], dist=dist, prefetch=staged[(i + 1) % 2])
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        out = tr.step(staged[i % 2], dist=dist, prefetch=staged[(i + 1) % 2])
    barrier()
    el = pp.frame_shard.max_over_ranks(time.perf_counter() - t0, dist, comm_dev)
    # the gradient exchange by itself: a few more steps with an event pair around the all-reduce on the engine's stream
    # (outside the timed region: the events are two more nodes on the stream)
    tr.time_allreduce = True
    for i in range(6):
        tr.step(staged[i % 2], dist=dist, prefetch=staged[(i + 1) % 2])
    ar = tr.allreduce_ms()
    tr.time_allreduce = False
    ar_ms = pp.frame_shard.max_over_ranks(float(np.median(ar)) if ar else 0.0, dist, comm_dev)
    res = {"workload": f"cfg-A training step, {batch} frames/GPU x 16384 pts, {tr.params.numel()} trainable parameters "
                       f"({tr.params.numel() * 4 / 1e6:.1f} MB gradient all-reduce per step)",
           "n_gpus": n_gpus, "steps": steps, "ms_per_step": el / steps * 1e3, "steps_per_s": steps / el,
           "samples_per_s": n_gpus * batch * steps / el, "last_loss": out["loss"],
           "allreduce": {"ms_median_max_over_ranks": ar_ms, "bytes": tr.params.numel() * 4, "world": n_gpus,
                         "moved_bytes_between_gpus": bool(dist is not None and n_gpus > 1),
                         "busbw_GBps": (2.0 * (n_gpus - 1) / n_gpus * tr.params.numel() * 4 / (ar_ms * 1e-3) / 1e9
                                        if ar_ms > 0 and n_gpus > 1 else None),
                         "timing": "event pair around optim.allreduce_gradients on the engine's stream, 6 extra steps"}}
    if rank == 0:
        tr.engine.set_profiling(True)
        tr.forward_backward(frames, labels, reg)
        agg = {}
        for name, ms in tr.engine.kernel_times():
            a = agg.setdefault(name.split(":")[0], [0.0, 0])
            a[0] += ms
            a[1] += 1
        tr.engine.set_profiling(False)
        res["kernel_ms_per_step"] = {k: round(v[0], 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
        res["kernel_launches_per_step"] = {k: v[1] for k, v in agg.items()}
        res["sum_kernel_ms_per_step"] = sum(v[0] for v in agg.values())
        gemm = "k_tr_gemm2" if "k_tr_gemm2" in agg else ("k_tr_gemm" if "k_tr_gemm" in agg else None)
        if gemm:
            # k_tr_gemm2: three bfloat16 pieces per operand, six 16-bit products per float32 product (train.hip);
            # k_tr_gemm (PP_TRAIN_GEMM=f32): the float32 matrix instruction.  Round 4: the separable layers' forward
            # products may run inside k_sep_u_tr (depthwise + product + statistics in one launch, two float16 pieces):
            # that launch time counts as product time here (its depthwise share included), against the SAME roof as
            # before (2500 / 6), so the fraction stays comparable with round 3's and cannot rise by moving work out of
            # the counted symbols
            split = gemm == "k_tr_gemm2"
            peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if split else F32_MFMA_PEAK_TFLOPS
            fl = train_gemm_flops(d, batch)
            counted = [k for k in ("k_tr_gemm2", "k_tr_gemm", "k_sep_u_tr") if k in agg]
            t = sum(agg[k][0] for k in counted) * 1e-3
            res["roofline"] = {"bound": "mfma", "kernel": gemm, "achieved": fl / t / 1e12, "peak": peak,
                               "unit": "TFLOP/s", "frac": fl / t / 1e12 / peak, "traffic": None,
                               "algorithmic_flops_per_step": fl, "launches_per_step": sum(agg[k][1] for k in counted),
                               "ms_per_step": t * 1e3, "symbols_counted": counted,
                               "mfma_roof": ("16-bit dense 2500 TFLOP/s / 6 products per fp32 product (bf16 x 3 pieces)"
                                             if split else "f32 MFMA (v_mfma_f32_32x32x2_f32)")}
    for st in staged:
        st.close()
    tr.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight per GPU: each has its own engine handle (stream + workspaces), steps "
                         "take turns so the upload and the latency-bound front of one step (voxelise, PFN, NMS) "
                         "overlap the backbone of the others")
    ap.add_argument("--pool", type=int, default=4, help="distinct staged batches cycled through (pinned host memory)")
    ap.add_argument("--plain", action="store_true",
                    help="warm-up + timed region only (no per-kernel event pass, no latency / cfg-K / CPU legs): the "
                         "run rocprofv3 wraps for profiles/*_kernel_stats.csv, one regime per file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfgk", action="store_true", help="skip the KITTI-shaped B=32 leg (detail.cfgK)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg (detail.train, configs[4])")
    ap.add_argument("--latency-b1", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-latency-b1", action="store_true", help="skip the batch-1 latency leg")
    ap.add_argument("--train-batch", type=int, default=0,
                    help="frames per GPU of the --only train leg (default: the reference's training batch, 2)")
    ap.add_argument("--only", choices=["cfgk", "train"], default=None,
                    help="run ONE secondary leg by itself and print its dict (profiling runs: one regime per trace)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = world if world > 1 else 1
    want_cpu = rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and not args.plain
    if want_cpu:
        # the checker library of the cpu_baseline leg is built (make, a child process) and loaded now, before
        # anything in this process has touched the GPU
        from oracle import c_oracle
        c_oracle.lib()
    import torch
    dist = None
    comm_dev = f"cuda:{local_rank}"
    backend = None
    # Under a launcher (torch.distributed.run sets RANK / MASTER_ADDR) the process group is created for every world
    # size, 1 included: a one-rank run on a one-GPU box then exercises the same RCCL initialisation, device barrier
    # and all-reduces as the 8-rank run does.
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:
        import torch.distributed as dist
        # PP_BENCH_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks
        # (ranks share devices round-robin, the scalar all-reduces run on the CPU); the numbers mean nothing
        backend = os.environ.get("PP_BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend)
            comm_dev = "cpu"
        # communicator set-up (and RCCL's device buffers) now, before any engine memory or graph exists
        warm = torch.ones(1, device=comm_dev)
        dist.all_reduce(warm)
        dist.barrier()
        torch.cuda.synchronize()

    import pp_amd as pp
    pp._lib.lib()  # fails loudly if the HIP library is missing
    # this rank's CPUs = the NUMA node its GPU hangs off, before the page-locked staging pool is allocated
    try:
        numa = pp.frame_shard.pin_to_gpu_numa_node(pp.frame_shard.device_pci_bus_id(local_rank)) \
            if os.environ.get("PP_NO_NUMA_PIN", "") != "1" else {"pinned": False, "reason": "PP_NO_NUMA_PIN=1"}
    except Exception as ex:
        numa = {"pinned": False, "reason": repr(ex)}
    if args.only is not None:
        def _sync():
            torch.cuda.synchronize()
        leg = cfgk_leg(pp, local_rank, steps=max(4, min(args.steps, 60)),
                       only_inflight=(args.inflight if "--inflight" in sys.argv else None)) if args.only == "cfgk" else \
            train_leg(pp, local_rank, rank, n_gpus, dist, comm_dev, _sync, steps=max(20, min(args.steps, 100)),
                      batch=args.train_batch or 2)
        if rank == 0:
            print(json.dumps({"leg": args.only, **leg}))
        return
    B, N = args.batch, args.points
    cfg = pp.config.pedestrian_d435i_config(B)
    engines = [pp.Engine(cfg, max_batch=B, max_points_per_frame=max(N, 4096), device=local_rank)
               for _ in range(max(1, args.inflight))]
    eng = engines[0]
    d = eng.d
    weights = pp.weights.init_weights(d, seed=7)
    calib = pp.synth.default_calib()
    pool = max(1, args.pool)
    frame_ids = pp.frame_shard.rank_frames(rank, n_gpus, B * pool)     # this rank's frames (weak scaling)
    frames = [pp.synth.d435i_cloud(i, N, d.num_point_features) for i in frame_ids]
    stagings = [eng.staging(frames[j * B:(j + 1) * B]) for j in range(pool)]
    for e in engines:
        e.load_weights(weights)
        e.set_calib(np.stack([calib[0]] * B), np.stack([calib[1]] * B), B)

    torch_sync = [True]

    def barrier():
        if dist is not None:
            dist.barrier()
        for e in engines:          # the streams the work is on (the later legs run after these are closed)
            if e._h:
                e.sync()
        if torch_sync[0]:
            try:
                torch.cuda.synchronize()
            except Exception:      # torch's bundled runtime found no device (seen on some boxes); N = 1 only
                if dist is not None:
                    raise
                torch_sync[0] = False

    feeder = Feeder(engines, stagings)
    elapsed = timed_run(feeder, args.steps, args.warmup, barrier, settle=SETTLE_STEPS)
    elapsed = pp.frame_shard.max_over_ranks(elapsed, dist, comm_dev)
    counts = pp.frame_shard.gather_counts(B * args.steps, dist, comm_dev)
    ms_per_step = elapsed / args.steps * 1e3
    fps = sum(counts) / elapsed
    mean_det = feeder.n_det / max(feeder.n_batches * B, 1)
    im_np = eng.intermediates()["n_pillars"]
    info = eng.device_info()
    line = {
        "metric": "frames/sec (whole node) + p50 per-frame ms, 16k-pt pillars",
        "value": fps, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32" if os.environ.get("PP_GEMM_PREC", "")[:1] == "f"
                  else "f32 (GEMMs on the 16-bit matrix pipe as split operands -- two float16 pieces per value, 3 products, "
                       "fp32 accumulate: fp32-equivalent results, measured 1.5e-6 max from the fp32 oracle at this shape)"),
        "data": "synthetic",
        "config": {"workload": f"cfg-A (shipped d435i pedestrian config, 80x64 BEV, T=50, C=128), "
                               f"B={B} frames/GPU x {N} pts per step, host-to-device copy of a different staged batch "
                               f"every step -> raw points -> detections read by the host (BASELINE.json configs[1])",
                   "batch_per_gpu": B, "points_per_frame": N, "parallelism": f"frame-parallel x{n_gpus}, no collective",
                   "batches_in_flight_per_gpu": len(engines), "distinct_batches_cycled": pool,
                   "upload_in_timed_region": True,
                   "collective_backend": (("nccl (RCCL)" if backend == "nccl" else backend) + f", world {world}: barrier + "
                                          "MAX / SUM of scalars around the timed region, gradient all-reduce in detail.train")
                   if dist is not None else "none (single process)",
                   "numa": numa,
                   "mean_pillars_per_frame": float(im_np.mean()), "mean_detections_per_frame": mean_det,
                   "device": info["name"], "compute_units": info["compute_units"]},
    }
    if args.plain:
        if rank == 0:
            print(json.dumps(line))
        for s_ in stagings:
            s_.close()
        for e in engines:
            e.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    extras = {}
    # ---- the same loop with the points resident in HBM (no per-step upload): what round 1 reported ----
    res_steps = min(args.steps, 100)
    f_res = Feeder(engines, stagings, upload=False)
    el_res = timed_run(f_res, res_steps, min(args.warmup, 6), barrier)
    extras["resident_fps"] = n_gpus * B * res_steps / pp.frame_shard.max_over_ranks(el_res, dist, comm_dev)

    # ---- per-kernel execution times: --inflight batches sharing the chip, then ONE batch in flight ----
    prof_steps = 4 * len(engines)
    heads_fused = not any(t.endswith(":heads") for t in eng.layer_tags())
    npil = float(im_np.mean())
    s_ovl = kernel_pass(engines, lambda: Feeder(engines, stagings), prof_steps)
    k_ovl, l_ovl, layer_ovl, drop_ovl = summarise(s_ovl, prof_steps)
    s_iso = kernel_pass(engines[:1], lambda: Feeder(engines[:1], stagings), 8)
    k_iso, l_iso, layer_iso, drop_iso = summarise(s_iso, 8)
    roofs_iso = kernel_roofs(d, B, N, npil, k_iso, l_iso, layer_iso, heads_fused)
    roofs_ovl = kernel_roofs(d, B, N, npil, k_ovl, l_ovl, layer_ovl, heads_fused)
    dominant = max(k_iso, key=k_iso.get)
    roofline = roofline_of(dominant, roofs_iso[dominant])
    if dominant in roofs_ovl:
        ro = roofline_of(dominant, roofs_ovl[dominant])
        roofline["frac_overlapped"] = (ro["frac_of_mfma_roof"] if roofline["bound"] == "mfma" else ro["frac_of_hbm_roof"]) \
            if "frac_of_hbm_roof" in ro else ro["frac"]
        roofline["avg_launch_ms_overlapped"] = ro["avg_launch_ms"]
    roofline["timing"] = ("start/stop HIP events carried by each launch (hipExtLaunchKernelGGL), one batch in flight; "
                          "compare with AverageNs of profiles/r04_inflight1_kernel_stats.csv")
    # HBM traffic and matrix-pipe occupancy of the dominant kernel come from the committed counter passes
    # (rocprofv3 cannot run inside this process), per launch like `achieved`.  A pass is only quoted while the kernel
    # sources are the ones it ran on (sha of csrc/ recorded by the pass): otherwise null + "stale".
    csrc_sha = csrc_sha16()
    for key, pattern, field in (("traffic", "*pmc_traffic.json", "hbm_bytes_per_launch"),
                                ("mfma_busy_frac", "*pmc_sq.json", "mfma_busy_frac")):
        roofline[key] = None
        try:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
            if files:
                doc = json.load(open(files[-1]))
                ent = doc["kernels"].get(dominant) or doc["kernels"].get(dominant[:-1] + ",0>")   # (sixth template parameter: TR = 0)
                src = {"file": os.path.basename(files[-1]), "git_head": doc.get("git_head"), "csrc_sha16": doc.get("csrc_sha16")}
                if ent is not None and doc.get("csrc_sha16") == csrc_sha:
                    roofline[key] = ent.get(field)
                else:
                    src["stale"] = "kernel sources changed since the pass" if ent is not None else "kernel not in the pass"
                roofline[key + "_source"] = src
        except Exception as ex:
            roofline[key + "_source"] = {"error": repr(ex)}
    # the whole backbone against the HBM roof: sum of algorithmic bytes over sum of isolated launch time (the dominant
    # kernel's `frac` alone flatters the step: the other GEMM layers sit lower)
    lb_all = layer_bytes(d, B, heads_fused, npil)
    bb_bytes = sum(lb_all[layer] for layer in layer_iso if layer in lb_all)
    bb_ms = sum(layer_iso[layer][0] * layer_iso[layer][2] for layer in layer_iso if layer in lb_all)
    lf_all = layer_flops(d, B, heads_fused)
    bb_flops = sum(lf_all[layer] for layer in layer_iso if layer in lf_all)
    lf = layer_flops(d, B, heads_fused)
    extras.update({
        "kernel_ms_per_step_overlapped": {k: round(v, 4) for k, v in sorted(k_ovl.items(), key=lambda kv: -kv[1])},
        "kernel_ms_per_step_inflight1": {k: round(v, 4) for k, v in sorted(k_iso.items(), key=lambda kv: -kv[1])},
        "layer_ms_inflight1": {k: round(v[0] * v[2], 4) for k, v in layer_iso.items()},
        "sum_kernel_ms_per_step_inflight1": sum(k_iso.values()),
        "sum_kernel_ms_per_step_overlapped": sum(k_ovl.values()),
        "kernel_time_over_wall": sum(k_ovl.values()) / ms_per_step,   # > 1: kernels of the in-flight batches overlap
        "dropped_stall_samples": {"overlapped": drop_ovl, "inflight1": drop_iso},
        "backbone_tflops_end_to_end": sum(lf.values()) / (ms_per_step * 1e-3) / 1e12,
        "backbone_weighted_frac": bb_bytes / (bb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bb_ms > 0 else None,
        "backbone_weighted": {"algorithmic_bytes_per_step": bb_bytes, "isolated_ms_per_step": bb_ms,
                              "GBps": bb_bytes / (bb_ms * 1e-3) / 1e9 if bb_ms > 0 else None,
                              "fp32_equivalent_tflops": bb_flops / (bb_ms * 1e-3) / 1e12 if bb_ms > 0 else None,
                              "frac_of_split_mfma_roof": (bb_flops / (bb_ms * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS / SPLIT_TERMS)
                                                          if bb_ms > 0 else None)},
        "roofs_inflight1": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                                if kk not in ("mfma_roof", "mfma_peak")} for k, v in roofs_iso.items()},
    })

    # ---- one batch in flight: throughput and per-step latency distribution (upload included) ----
    f1 = Feeder(engines[:1], stagings)
    el1 = timed_run(f1, min(args.steps, 100), 4, barrier)
    extras["inflight1_fps"] = B * min(args.steps, 100) / el1
    step_ms = []
    for i in range(max(5, min(100, args.steps))):
        t1 = time.perf_counter()
        eng.upload_async(stagings[i % pool])
        eng.detect_async()
        eng.sync()
        step_ms.append((time.perf_counter() - t1) * 1e3)
    p50_step = float(np.median(step_ms))

    # ---- batch-1 latency (the reference's eval batch size): p50 per frame, upload included ----
    lat = lat95 = None
    if rank == 0 and not args.no_latency_b1:
        e1 = pp.Engine(pp.config.pedestrian_d435i_config(1), max_batch=1, max_points_per_frame=max(N, 4096),
                       device=local_rank, weights=weights)
        e1.set_calib(calib[0][None], calib[1][None], 1)
        st1 = [e1.staging(frames[i:i + 1]) for i in range(16)]
        ts = []
        for i in range(108):
            t1 = time.perf_counter()
            e1.upload_async(st1[i % 16])
            e1.detect_async()
            e1.sync()
            ts.append((time.perf_counter() - t1) * 1e3)
        lat = float(np.median(ts[8:]))
        lat95 = float(np.percentile(ts[8:], 95))
        for s_ in st1:
            s_.close()
        e1.close()

    try:    # the HBM figure measured in this run (1 GiB device copy, read + write) beside the 8 TB/s spec constant
        extras["hbm_copy_GBps"] = eng.device_copy_GBps(1 << 30, 5)
    except Exception as ex:
        extras["hbm_copy_GBps"] = None
        extras["hbm_copy_error"] = repr(ex)
    extras["hbm_peak_GBps_used"] = HBM_PEAK_GBS

    for s_ in stagings:
        s_.close()
    for e in engines:
        e.close()

    if rank == 0 and n_gpus == 1 and not args.no_cfgk:
        try:
            extras["cfgK"] = cfgk_leg(pp, local_rank)
        except Exception as ex:   # the leg must not take the headline line down with it
            extras["cfgK"] = {"error": repr(ex)}

    if not args.no_train:      # every rank takes part: the step ends in an all-reduce over the ranks
        try:
            extras["train"] = train_leg(pp, local_rank, rank, n_gpus, dist, comm_dev, barrier)
            # the same step at an MI355X-sized per-GPU batch (the reference trains with 2 frames per step on one GPU;
            # 288 GB of HBM hold the kept activations of far more)
            extras["train_b32"] = train_leg(pp, local_rank, rank, n_gpus, dist, comm_dev, barrier, steps=10, batch=32)
            # ... and at 64 frames per GPU: the small-map layers and the PFN are latency-bound at 32 (one round of
            # workgroups, chip half empty), so samples/s still rises with the batch
            extras["train_b64"] = train_leg(pp, local_rank, rank, n_gpus, dist, comm_dev, barrier, steps=8, batch=64)
            extras["train_b128"] = train_leg(pp, local_rank, rank, n_gpus, dist, comm_dev, barrier, steps=6, batch=128)
        except Exception as ex:
            if dist is not None:
                raise
            extras["train"] = {"error": repr(ex)}

    cpu = None
    if want_cpu:
        cpu = cpu_baseline(pp, d, weights, frames[:32], calib)

    if rank == 0:
        line.update({
            "p50_ms_per_step": p50_step,
            "p95_ms_per_step": float(np.percentile(step_ms, 95)),
            "schema": 4,
            "p50_ms_per_frame": p50_step / B,             # amortised, as in rounds 1-2 (round 3 put the step time here)
            "p50_ms_frame_latency_in_batch": p50_step,    # synchronous case: a frame waits for its batch
            "p50_ms_per_frame_batch1": lat,     # upload + detect + sync, wall clock, 100 frames
            "p95_ms_per_frame_batch1": lat95,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "detail": extras,
        })
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
