"""Soak of the optimizer step: many steps of the two feeding modes the bench uses (staged batches, the next batch's
points prefetched beside the running step, update enqueued on the engine's stream), with the checks a race would trip:
finite losses, the loss of a repeated (batch, parameters) pair bit-identical across two trainers fed the same way.

    python tools/train_soak.py [--b2-steps 2000] [--b32-steps 200]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pp_amd as pp

ap = argparse.ArgumentParser()
ap.add_argument("--b2-steps", type=int, default=2000)
ap.add_argument("--b32-steps", type=int, default=200)
args = ap.parse_args()


def batches(d, B, n):
    rng = np.random.default_rng(3)
    out = []
    for j in range(n):
        frames = [pp.synth.d435i_cloud(100 * j + i, 16384) for i in range(B)]
        labels = rng.choice([-1, 0, 0, 0, 0], size=(B, d.num_anchors)).astype(np.int32)
        reg = np.zeros((B, d.num_anchors, 7), np.float32)
        for b in range(B):
            pos = rng.choice(d.num_anchors, 30, replace=False)
            labels[b, pos] = 1
            reg[b, pos] = rng.normal(0, 0.4, (30, 7)).astype(np.float32)
        out.append((frames, labels, reg))
    return out


def run(B, steps):
    cfg = pp.config.pedestrian_d435i_config(B)
    d = pp.config.Derived(cfg)
    w = pp.weights.init_weights(d, seed=21)
    data = batches(d, B, 3)
    losses = []
    t0 = time.perf_counter()
    for rep in range(2):                       # two trainers, the same feed: every loss must repeat bit for bit
        tr = pp.Trainer(cfg, w, max_batch=B, max_points_per_frame=16384, learning_rate=1e-4)
        staged = [tr.stage(*x) for x in data]
        seq = []
        for s in range(steps):
            out = tr.step(staged[s % 3], prefetch=staged[(s + 1) % 3])
            if not np.isfinite(out["loss"]):
                raise SystemExit(f"B={B}: non-finite loss at step {s}")
            seq.append(out["loss"])
        losses.append(seq)
        for st in staged:
            st.close()
        tr.close()
    bad = sum(1 for a, b in zip(*losses) if a != b)
    print(f"B={B}: 2 x {steps} optimizer steps in {time.perf_counter() - t0:.1f} s, loss {losses[0][0]:.4f} -> "
          f"{losses[0][-1]:.4f}, steps whose loss differs between the two runs: {bad}")
    return bad


bad = run(2, args.b2_steps) + run(32, args.b32_steps)
sys.exit(1 if bad else 0)
