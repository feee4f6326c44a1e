"""KITTI-shaped configuration (BASELINE.json configs[2]): B=32 x 20k-point clouds, 432x496 BEV, end to end.

    python tools/bench_cfgk.py [--batch 32] [--steps 10]
Prints per-kernel times and frames/s (one batch in flight).  Not the headline bench (bench.py is cfg-A)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pp_amd as pp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
B = args.batch
cfg = pp.config.kitti_shaped_config(B)
eng = pp.Engine(cfg, max_batch=B, max_points_per_frame=24000)
eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
frames = [pp.synth.kitti_cloud(i) for i in range(B)]
calib = pp.synth.default_calib()
eng.upload(frames, np.repeat(calib[0][None], B, 0), np.repeat(calib[1][None], B, 0))
for _ in range(2):
    eng.detect_async()
    eng.sync()
t0 = time.perf_counter()
for _ in range(args.steps):
    eng.detect_async()
eng.sync()
el = time.perf_counter() - t0
print(f"cfg-K B={B}: {B * args.steps / el:.0f} frames/s, {el / args.steps * 1e3:.2f} ms per step")
eng.set_profiling(True)
eng.detect_async()
eng.sync()
tot = {}
for tag, ms in eng.kernel_times():
    k = tag.split(":")[0]
    tot[k] = tot.get(k, 0.0) + ms
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v * 1e3:9.1f} us")
dets, n = eng.detections()
print("mean detections per frame", float(np.mean(n)), "pillars", float(eng.intermediates()["n_pillars"].mean()))
