"""PCIe-inclusive rate of the default workload: every step hands the engine a packed HOST buffer of 64 x 16 384 points
(pp_upload_points: H2D on the engine's stream) before pp_detect_async; detections come back to pinned host memory.

    python tools/pcie_rate.py [--steps 200] [--inflight 3]
This is NOT bench.py's headline `value` (inputs resident in HBM); DESIGN.md section 5 quotes it beside it."""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pp_amd as pp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--inflight", type=int, default=3)
args = ap.parse_args()
B, N = 64, 16384
cfg = pp.config.pedestrian_d435i_config(B)
engines = [pp.Engine(cfg, max_batch=B, max_points_per_frame=N) for _ in range(args.inflight)]
w = pp.weights.init_weights(engines[0].d, seed=7)
frames = [pp.synth.d435i_cloud(i, N) for i in range(B)]
pts, offs = pp.Engine._pack(frames, 3)
for e in engines:
    e.load_weights(w)
    e.upload(frames)
lib = engines[0]._lib
pp_ptr = pts.ctypes.data_as(ctypes.c_void_p)
of_ptr = offs.ctypes.data_as(ctypes.c_void_p)


def step(e):
    st = lib.pp_upload_points(e._h, pp_ptr, of_ptr, B)
    assert st == 0
    e.detect_async()


for i in range(10):
    step(engines[i % len(engines)])
for e in engines:
    e.sync()
t0 = time.perf_counter()
for i in range(args.steps):
    step(engines[i % len(engines)])
for e in engines:
    e.sync()
el = time.perf_counter() - t0
print(f"pageable host buffer -> detections, {args.inflight} in flight: "
      f"{B * args.steps / el:.0f} frames/s, {el / args.steps * 1e3:.3f} ms per step "
      f"({pts.nbytes / 1e6:.1f} MB H2D per step = {pts.nbytes * args.steps / el / 1e9:.1f} GB/s)")
