"""How far the HIP path is from the fp32 oracle on the shipped configuration (2 frames x 16k points): max |diff|
of the head outputs and of the final boxes / scores, for the split-precision and the fp32-MFMA builds of the
GEMMs (PP_GEMM_PREC=f32).  Uses the oracle as the checker (test infrastructure)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pp_amd  # noqa: E402
import util_ref  # noqa: E402

cfg = pp_amd.config.pedestrian_d435i_config(2)
eng = pp_amd.Engine(cfg, max_batch=2, max_points_per_frame=16384)
d = eng.d
w = pp_amd.weights.init_weights(d, seed=7)
eng.load_weights(w)
frames = [pp_amd.synth.d435i_cloud(40 + i) for i in range(2)]
rect, trv, p2 = pp_amd.synth.default_calib()
dets, n = eng.detect(frames, np.stack([rect] * 2), np.stack([trv] * 2))
im = eng.intermediates()
ref = util_ref.oracle_detect(d, w, frames, rect, trv, p2)
print("GEMM arithmetic:", os.environ.get("PP_GEMM_PREC", "bf16x3 split"))
for k in ("box_preds", "cls_preds", "dir_cls_preds"):
    diff = np.abs(im[k] - ref["preds"][k])
    print(f"  {k:14s} max |diff| {diff.max():.3e}   rms {np.sqrt((diff ** 2).mean()):.3e}   max |ref| {np.abs(ref['preds'][k]).max():.2f}")
for b in range(2):
    r = ref["dets"][b]
    k = len(r["scores"])
    assert int(n[b]) == k
    print(f"  frame {b}: {k} detections, max |score diff| {np.abs(dets[b]['score'][:k] - r['scores']).max():.3e}, "
          f"max |box diff| {np.abs(dets[b]['box3d_lidar'][:k] - r['box3d_lidar']).max():.3e}")
eng.close()
