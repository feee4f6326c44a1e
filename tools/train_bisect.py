"""Debug aid: worst relative gradient error of pp_train_step vs the autograd oracle over config variants."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pp_amd as pp
import util_ref
from oracle import train_ref

def run(tag, cfg, frames, balanced=False):
    d = pp.config.Derived(cfg)
    rng = np.random.default_rng(11)
    B, A = len(frames), d.num_anchors
    labels = rng.choice([-1, 0, 0, 0, 0], size=(B, A)).astype(np.int32)
    reg = np.zeros((B, A, 7), np.float32)
    pos = rng.choice(A, min(40, A // 4), replace=False); labels[0, pos] = 1; reg[0, pos] = rng.normal(0, 0.4, (len(pos), 7))
    if balanced:
        labels = rng.choice([1, 0], size=(B, A)).astype(np.int32)
        reg = (rng.normal(0, 0.4, (B, A, 7)) * (labels > 0)[..., None]).astype(np.float32)
    w = pp.weights.init_weights(d, seed=21)
    tr = pp.Trainer(cfg, w, max_batch=B, max_points_per_frame=16384)
    out = tr.forward_backward(frames, labels, reg)
    rect, trv, p2 = pp.synth.default_calib()
    ex, _ = util_ref.oracle_example(d, frames, rect, trv, p2)
    vals, grads, stats, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0])
    got = tr.gradients()
    import torch
    v64, g64, _, _ = train_ref.training_step(d, w, ex, labels, reg, ex[6][0], dtype=torch.float64)
    rows, rows32 = [], []
    for k, g in g64.items():
        sc = max(float(np.abs(g).max()), 1e-12)
        rows.append((float(np.abs(got[k] - g).max()) / sc, k))
        rows32.append((float(np.abs(grads[k] - g).max()) / sc, k))
    l2 = sorted(((float(np.linalg.norm((got[k] - g).ravel()) / max(np.linalg.norm(g.ravel()), 1e-30)), k) for k, g in g64.items()), reverse=True)
    l232 = sorted(((float(np.linalg.norm((grads[k] - g).ravel()) / max(np.linalg.norm(g.ravel()), 1e-30)), k) for k, g in g64.items()), reverse=True)
    print("   HIP  L2 vs f64:", [(f"{r:.1e}", k) for r, k in l2[:3]], flush=True)
    print("   t32  L2 vs f64:", [(f"{r:.1e}", k) for r, k in l232[:3]], flush=True)
    rows.sort(reverse=True); rows32.sort(reverse=True)
    print(tag, "loss", out["loss"], vals["loss"], v64["loss"], flush=True)
    print("   HIP  vs f64:", [(f"{r:.1e}", k) for r, k in rows[:3]], flush=True)
    print("   t32  vs f64:", [(f"{r:.1e}", k) for r, k in rows32[:3]], flush=True)
    tr.close()

rng = np.random.default_rng(4)
tiny_frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (900, 400)]
base = pp.config.tiny_config(2)
run("tiny", base, tiny_frames)
A = pp.config.pedestrian_d435i_config(2)
framesA = [pp.synth.d435i_cloud(30 + i, 16384) for i in range(2)]
c = copy.deepcopy(A); c["model"]["second"]["rpn"].update(layer_nums=[1, 1, 1])
run("A", A, framesA)
A8 = pp.config.pedestrian_d435i_config(8)
run("A B=8", A8, [pp.synth.d435i_cloud(30 + i, 16384) for i in range(8)])
