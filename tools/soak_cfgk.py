"""Soak of the KITTI-shaped configuration (B = 32 x 20 000 points): the upload-time voxeliser with its two product sets, the
pillar-centric PFN / occupancy bitmap, and the frame sub-ranges (one engine, cache budget 256) or two engines in flight
(budget 0) -- every result must equal the first pass over the same staged batch.    python tools/soak_cfgk.py [steps=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pp_amd as pp

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, N, pool = 32, 20000, 3
cfg = pp.config.kitti_shaped_config(B, num_class=2)
calib = pp.synth.default_calib()
for n_eng, budget in ((1, 256), (2, 0)):
    engs = [pp.Engine(cfg, max_batch=B, max_points_per_frame=N) for _ in range(n_eng)]
    w = pp.weights.init_weights(engs[0].d, seed=5)
    for e in engs:
        e.load_weights(w)
        e.set_calib(np.stack([calib[0]] * B), np.stack([calib[1]] * B), B)
        e.set_cache_budget(budget)
    stg = [engs[0].staging([pp.synth.kitti_cloud(2000 + j * B + i, N) for i in range(B)]) for j in range(pool)]
    ref, busy, bad = {}, [None] * n_eng, 0
    t0 = time.perf_counter()
    for s in range(steps + n_eng):
        k = s % n_eng
        if s < steps:
            engs[k].upload_async(stg[s % pool])          # (before the previous pass of this engine is collected: the bench's order)
        if busy[k] is not None:
            d, n = engs[k].detections()
            sig = (n.tobytes(), b"".join(d[b, :n[b]].tobytes() for b in range(B)))
            if ref.setdefault(busy[k], sig) != sig:
                bad += 1
            busy[k] = None
        if s < steps:
            engs[k].detect_async()
            busy[k] = s % pool
    el = time.perf_counter() - t0
    print(f"cfg-K, {n_eng} engine(s), cache budget {budget}: {steps} steps, {B * steps / el:.0f} frames/s, mismatching results: {bad}")
    for x in stg:
        x.close()
    for e in engs:
        e.close()
