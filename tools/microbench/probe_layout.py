"""Robustness probe: the engine next to large foreign device allocations made before, between and after its own
(what an RCCL / torch process looks like).  Runs detection on 3 engines in flight and compares the results of
every phase with the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import pp_amd as pp

hold = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")]
B = 16
cfg = pp.config.pedestrian_d435i_config(B)
w = None
engs = []
for k in range(3):
    e = pp.Engine(cfg, max_batch=B, max_points_per_frame=16384)
    if w is None:
        w = pp.weights.init_weights(e.d, seed=7)
    e.load_weights(w)
    engs.append(e)
    hold.append(torch.empty((1 << 28) * (k + 1), dtype=torch.uint8, device="cuda:0"))
frames = [pp.synth.d435i_cloud(i) for i in range(B)]
rect, trv, _ = pp.synth.default_calib()
r, t = np.stack([rect] * B), np.stack([trv] * B)
for e in engs:
    e.upload(frames, r, t)
ref = None
for phase in range(4):
    for _ in range(3):
        for e in engs:
            e.detect_async()
    outs = []
    for e in engs:
        e.sync()
        d, n = e.detections()
        outs.append((n.copy(), d["box3d_lidar"].copy()))
    if ref is None:
        ref = outs[0]
    for n, bx in outs:
        assert np.array_equal(n, ref[0]) and np.array_equal(bx, ref[1])
    hold.append(torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"))   # more foreign memory, then again
    hold[-1].fill_(phase)
    torch.cuda.synchronize()
    if phase == 2:
        hold.pop(1)   # free one foreign block: the allocator may unmap / reuse it
        torch.cuda.empty_cache()
print("probe OK", [int(x) for x in ref[0][:4]])
