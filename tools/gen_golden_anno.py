"""Golden vectors for the caller-side formatting row, produced by RUNNING the
reference's own predict_kitti_to_anno / remove_low_score / box_camera_to_lidar (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_anno.py   ->  tests/golden/ref_anno.npz
The predictions fed in are synthetic (seeded), shaped like VoxelNet.predict's output dicts
(model/voxelnet.py:1362-1379), including one frame with no detections.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ld, ehf = ref_shim.load_reference()

rng = np.random.default_rng(77)
frames = []
for b, n in enumerate((7, 0, 3)):
    if n == 0:
        frames.append({"bbox": None, "box3d_camera": None, "box3d_lidar": None, "scores": None,
                       "label_preds": None, "batch_idx": b})
        continue
    lidar = np.concatenate([rng.uniform([0.3, -2.4, -1.5], [7.0, 2.4, 0.0], (n, 3)),
                            rng.uniform(0.4, 1.8, (n, 3)), rng.uniform(-3.1, 6.2, (n, 1))], axis=1).astype(np.float32)
    cam = np.concatenate([np.stack([-lidar[:, 1], -lidar[:, 2], lidar[:, 0]], 1), lidar[:, [4, 5, 3]], lidar[:, 6:7]],
                         axis=1).astype(np.float64)
    frames.append({"bbox": np.tile(np.array([[400., 200., 500., 400.]]), (n, 1)), "box3d_camera": cam,
                   "box3d_lidar": lidar, "scores": np.sort(rng.uniform(0.2, 0.95, n).astype(np.float32))[::-1],
                   "label_preds": np.zeros(n, dtype=np.int64), "batch_idx": b})
example = [None] * 10
example[9] = np.zeros((3, 2), np.int32)
out = {}
for tag, limit in (("nolimit", None), ("limit", [0, -2.56, -3.0, 6.40, 2.56, 3.0])):
    annos = ehf.predict_kitti_to_anno(example, ["Pedestrian"], frames, limit, False)
    for b, a in enumerate(annos):
        for k, v in a.items():
            out[f"{tag}_{b}_{k}"] = np.asarray(v)
    if tag == "nolimit":
        f = ehf.remove_low_score(annos[0], 0.45)
        for k, v in f.items():
            out[f"filtered_{k}"] = np.asarray(v)
        boxes_camera = np.concatenate([f["location"], f["dimensions"], f["rotation_y"][..., np.newaxis]], axis=1)
        R0 = np.array([1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]).reshape(3, 3)
        Tr = np.array([0.0, -1.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0]).reshape(3, 4)
        out["prod_boxes_lidar"] = ehf.box_camera_to_lidar(boxes_camera, R0, Tr)
        out["R0"], out["Tr"] = R0, Tr
for b, fr in enumerate(frames):
    for k in ("bbox", "box3d_camera", "box3d_lidar", "scores", "label_preds"):
        if fr[k] is not None:
            out[f"in_{b}_{k}"] = fr[k]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_anno.npz"), **out)
print(sorted(out.keys())[:12], len(out))
