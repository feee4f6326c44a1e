"""Per-tensor gradient differences between a batch of `copies` x a 2-frame batch and the 2-frame batch itself (the
size-independent property of tests/test_gpu_train.py::test_large_batch_paths_match_the_two_frame_step), for bisecting
the training kernels a full-chip batch selects.    python tools/train_dup_check.py [copies=64]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pp_amd as pp
import test_gpu_train as T

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg2 = T._variant(pp, "wide", 2)
rng = np.random.default_rng(5)
frames = [rng.uniform([0, -0.64, -3], [1.6, 0.64, 3], (n, 3)).astype(np.float32) for n in (900, 400)]
d, labels, reg = T._problem(pp, cfg2, frames, 13)
w = pp.weights.init_weights(d, seed=23)
tr = pp.Trainer(cfg2, w, max_batch=2, max_points_per_frame=4096)
out2 = tr.forward_backward(frames, labels, reg)
g2 = tr.gradients()
tr.close()
B = 2 * copies
trb = pp.Trainer(T._variant(pp, "wide", B), w, max_batch=B, max_points_per_frame=4096)
outb = trb.forward_backward(frames * copies, np.tile(labels, (copies, 1)), np.tile(reg, (copies, 1, 1)))
gb = trb.gradients()
print("loss", out2["loss"], outb["loss"])
for name, g in g2.items():
    dmax = float(np.abs(gb[name] - g).max()) / max(float(np.abs(g).max()), 1e-12)
    print(f"{dmax:10.3e}  {name}")
trb.close()
