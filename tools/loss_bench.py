"""Head-loss kernel (pp_head_loss) at B=64 on cfg-A: kernel time and achieved HBM bandwidth.

    python tools/loss_bench.py [--batch 64]
Algorithmic bytes per frame: 256 B per head pixel (head row read + gradient row written) + 32 B per anchor
(label + 7 targets)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pp_amd as pp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
B = args.batch
eng = pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=20000)
d = eng.d
eng.load_weights(pp.weights.init_weights(d, seed=7))
eng.detect([pp.synth.d435i_cloud(i) for i in range(B)])
rng = np.random.default_rng(0)
labels = rng.choice([-1, 0, 0, 0, 0, 0, 0, 1], size=(B, d.num_anchors)).astype(np.int32)
reg = rng.normal(0, 0.3, (B, d.num_anchors, 7)).astype(np.float32)
for _ in range(3):
    eng.head_loss(labels, reg)
eng.set_profiling(True)
ts = []
for _ in range(10):
    out = eng.head_loss(labels, reg)
    ts += [ms for tag, ms in eng.kernel_times() if tag.startswith("k_loss")]
eng.set_profiling(False)
t = float(np.median(ts)) * 1e-3
nbytes = B * (d.head_h * d.head_w * 256 + d.num_anchors * 32)
print(f"B={B}: k_loss_count + k_loss_pixels + k_loss_finish {t * 1e6:.1f} us, {nbytes / 1e6:.1f} MB algorithmic, "
      f"{nbytes / t / 1e9:.0f} GB/s ({nbytes / t / 8e12:.3f} of 8 TB/s); loss {out['loss']:.6f}")
