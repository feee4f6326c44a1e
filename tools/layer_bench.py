"""Kernel-tuning aid: per-layer launch time of the RPN GEMM kernels with phase ablation.

    python tools/layer_bench.py [--batch 64] [--layers 5,7] [--ablate 0,1,2,4,8]
ablate bits: 16 / 32 force the uniform-wave / producer-consumer separable kernel, 2048 / 4096 force the
split-precision bf16 / fp32 MFMA instantiation; in a -DPP_KERNEL_STAMPS build of libpp_hip.so also 1 no MFMA,
2 no depthwise FMAs, 4 no epilogue stores, 8 no global activation loads, 256 outer window columns not loaded
(half the window traffic of a stride-1 layer), 64 phase stamps (PP_STAMPS_OUT).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pp_amd as pp  # noqa: E402
from bench import layer_flops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--layers", default="")
ap.add_argument("--ablate", default="0")
ap.add_argument("--config", default="A")
args = ap.parse_args()
B = args.batch
cfg = pp.config.pedestrian_d435i_config(B) if args.config == "A" else pp.config.kitti_shaped_config(B)
eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=20000)
eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
frames = [pp.synth.d435i_cloud(i) if args.config == "A" else pp.synth.kitti_cloud(i) for i in range(B)]
eng.detect(frames)   # fill the activation buffers with real data
tags = eng.layer_tags()
lf = layer_flops(eng.d, B, heads_fused=not any(t.endswith(':heads') for t in tags))
sel = [int(v) for v in args.layers.split(",")] if args.layers else range(len(tags))
abl = [int(v) for v in args.ablate.split(",")]
tot = {a: 0.0 for a in abl}
for i in sel:
    name = tags[i].split(":")[1]
    row = f"{i:2d} {tags[i]:34s}"
    for a in abl:
        ms = eng.bench_layer(i, B, reps=20, ablate=a)
        tot[a] += ms
        row += f"  a{a}: {ms * 1e3:7.1f} us"
        if a == 0:
            row += f" ({lf[name] / (ms * 1e-3) / 1e12:5.1f} TF)"
    print(row)
print("total", {a: round(v, 4) for a, v in tot.items()})
