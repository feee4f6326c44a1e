"""Soak: many steps of the two feeding modes with a determinism check (every result must equal the first pass's).

    python tools/soak.py [--b64-steps 5000] [--b1-steps 20000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pp_amd as pp

ap = argparse.ArgumentParser()
ap.add_argument("--b64-steps", type=int, default=5000)
ap.add_argument("--b1-steps", type=int, default=20000)
args = ap.parse_args()
calib = pp.synth.default_calib()


def run(B, steps, pool):
    engs = [pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=16384) for _ in range(2)]
    w = pp.weights.init_weights(engs[0].d, seed=7)
    for e in engs:
        e.load_weights(w)
        e.set_calib(np.stack([calib[0]] * B), np.stack([calib[1]] * B), B)
    stg = [engs[0].staging([pp.synth.d435i_cloud(j * B + i) for i in range(B)]) for j in range(pool)]
    ref = {}
    busy = [None, None]
    bad = 0
    t0 = time.perf_counter()
    for s in range(steps + 2):
        k = s % 2
        if busy[k] is not None:
            d, n = engs[k].detections()
            key = busy[k]
            sig = (n.tobytes(), b"".join(d[b, :n[b]].tobytes() for b in range(B)))
            if key not in ref:
                ref[key] = sig
            elif ref[key] != sig:
                bad += 1
            busy[k] = None
        if s < steps:
            engs[k].upload_async(stg[s % pool])
            engs[k].detect_async()
            busy[k] = s % pool
    el = time.perf_counter() - t0
    for s_ in stg:
        s_.close()
    for e in engs:
        e.close()
    print(f"B={B}: {steps} steps, {B * steps / el:.0f} frames/s, mismatching results: {bad}", flush=True)
    return bad


bad = run(64, args.b64_steps, 4) + run(1, args.b1_steps, 16)
sys.exit(1 if bad else 0)
