"""Distance of the HIP path from the fp32 oracle at the benchmarked shape (cfg-A, B=64, the persistent k_sep_u /
k_deconv_u kernels) and on one KITTI-shaped frame, for the library PP_HIP_LIB selects (A/B of the GEMM operand
splits: libpp_hip.so = PP_SPLIT_MODE 0, bf16 x 3 pieces / 6 products; libpp_hip_f16.so = mode 1, f16 x 2 pieces /
3 products; libpp_hip_f16x4.so = mode 2, 4 products).  The oracle is the checker (test infrastructure).
    PP_HIP_LIB=libpp_hip_f16.so python tools/parity_margin2.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pp_amd  # noqa: E402
import util_ref  # noqa: E402

out = {"lib": os.environ.get("PP_HIP_LIB", "libpp_hip.so"), "gemm_prec_env": os.environ.get("PP_GEMM_PREC", "")}
rect, trv, p2 = pp_amd.synth.default_calib()


def margins(eng, w, frames, picks):
    B = len(frames)
    dets, n = eng.detect(frames, np.stack([rect] * B), np.stack([trv] * B))
    im = eng.intermediates()
    worst = {"head_max": 0.0, "head_rms": 0.0, "score_max": 0.0, "box_max": 0.0, "same_detections": True}
    for b in picks:
        ref = util_ref.oracle_detect(eng.d, w, [frames[b]], rect, trv, p2)
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            diff = np.abs(im[k][b] - ref["preds"][k][0])
            worst["head_max"] = max(worst["head_max"], float(diff.max()))
            worst["head_rms"] = max(worst["head_rms"], float(np.sqrt((diff ** 2).mean())))
        r = ref["dets"][0]
        k = 0 if r["scores"] is None else len(r["scores"])
        if int(n[b]) != k:
            worst["same_detections"] = False
            continue
        if k:
            worst["score_max"] = max(worst["score_max"], float(np.abs(dets[b]["score"][:k] - r["scores"]).max()))
            worst["box_max"] = max(worst["box_max"], float(np.abs(dets[b]["box3d_lidar"][:k] - r["box3d_lidar"]).max()))
    return worst


B = 64
eng = pp_amd.Engine(pp_amd.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=16384)
w = pp_amd.weights.init_weights(eng.d, seed=7)
eng.load_weights(w)
frames = [pp_amd.synth.d435i_cloud(i) for i in range(B)]
out["cfgA_B64"] = margins(eng, w, frames, [0, 21, 42, 63])
out["cfgA_B64"]["kernels"] = sorted(set(t.split(":")[0] for t in eng.layer_tags()))
eng.close()
eng = pp_amd.Engine(pp_amd.config.kitti_shaped_config(32, num_class=2), max_batch=32, max_points_per_frame=20000)
w = pp_amd.weights.init_weights(eng.d, seed=5)
eng.load_weights(w)
frames = [pp_amd.synth.kitti_cloud(300 + i) for i in range(32)]
out["cfgK_B32"] = margins(eng, w, frames, [3])
eng.close()
print(json.dumps(out))
