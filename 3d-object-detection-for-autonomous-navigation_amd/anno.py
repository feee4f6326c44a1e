"""Caller-side formatting that follows the hot path.

Mirrors, on the host, what `train.py evaluate` does with `net.predict`'s output:
  predict_kitti_to_anno   libraries/eval_helper_functions.py:150-273  (KITTI-style annos per frame)
  remove_low_score        libraries/eval_helper_functions.py:60-68
  production post-filter  train.py:810-828  (score >= 0.45, camera -> lidar, +0.9 m lift for RViz)
Pure numpy on <= 50 boxes per frame; nothing here touches the GPU.
"""
import os
import pickle

import numpy as np

_ANNO_KEYS = ("bbox", "name", "truncated", "occluded", "alpha", "dimensions", "location", "rotation_y", "score")


def empty_result_anno():
    """second/data/kitti_common.py:707-720."""
    return {
        "name": np.array([]), "truncated": np.array([]), "occluded": np.array([]), "alpha": np.array([]),
        "bbox": np.zeros([0, 4]), "dimensions": np.zeros([0, 3]), "location": np.zeros([0, 3]),
        "rotation_y": np.array([]), "score": np.array([]),
    }


def predict_kitti_to_anno(example, class_names, predictions_dicts, center_limit_range=None, lidar_input=False,
                          global_set=None):
    """One anno dict per frame; coordinates in camera space (KITTI style).  `example[9]` (image
    shapes) is read like the reference does but not used by this branch."""
    annos = []
    for preds in predictions_dicts:
        batch_idx = preds["batch_idx"]
        anno = None
        if preds["box3d_camera"] is not None:
            rows = {k: [] for k in _ANNO_KEYS}
            limit = None if center_limit_range is None else np.array(center_limit_range)
            for box_2d, box, box_lidar, score, label in zip(preds["bbox"], preds["box3d_camera"], preds["box3d_lidar"],
                                                             preds["scores"], preds["label_preds"]):
                if limit is not None and (np.any(box_lidar[:3] < limit[:3]) or np.any(box_lidar[:3] > limit[3:])):
                    continue
                rows["name"].append(class_names[int(label)])
                rows["bbox"].append(box_2d)
                rows["truncated"].append(0.0)
                rows["occluded"].append(0)
                rows["alpha"].append(-np.arctan2(-box_lidar[1], box_lidar[0]) + box[6])
                rows["dimensions"].append(box[3:6])
                rows["location"].append(box[:3])
                rows["rotation_y"].append(box[6])
                if global_set is not None:
                    for _ in range(100000):
                        if score in global_set:
                            score -= 1 / 100000
                        else:
                            global_set.add(score)
                            break
                rows["score"].append(score)
            if rows["name"]:
                anno = {k: np.stack(v) for k, v in rows.items()}
        if anno is None:
            anno = empty_result_anno()
        anno["batch_idx"] = np.array([batch_idx] * anno["name"].shape[0], dtype=np.int64)
        annos.append(anno)
    return annos


def remove_low_score(image_anno, thresh):
    keep = [i for i, s in enumerate(image_anno["score"]) if s >= thresh]
    return {k: v[keep] for k, v in image_anno.items()}


def camera_to_lidar(points, r_rect, velo2cam):
    """libraries/eval_helper_functions.py:42-56: r_rect [3,3] and velo2cam [3,4] (the calib dict of
    train.py:681-682) are made homogeneous, the product is inverted and applied."""
    shape = list(points.shape[0:-1])
    if points.shape[-1] == 3:
        points = np.concatenate([points, np.ones(shape + [1])], axis=-1)
    r4 = np.eye(4)
    r4[0:3, 0:3] = np.asarray(r_rect)[0:3, 0:3]
    v4 = np.eye(4)
    v4[0:3, :] = np.asarray(velo2cam)[0:3, :]
    lidar = np.dot(points, np.linalg.inv(np.dot(r4, v4).T))
    return lidar[..., :3]


def box_camera_to_lidar(data, r_rect, velo2cam):
    """libraries/eval_helper_functions.py:33-38."""
    xyz = camera_to_lidar(data[:, 0:3], r_rect, velo2cam)
    l, h, w, r = data[:, 3:4], data[:, 4:5], data[:, 5:6], data[:, 6:7]
    return np.concatenate([xyz, w, l, h, r], axis=1)


def production_boxes(dt_anno, r_rect, velo2cam, min_score=0.45, lift=0.9):
    """train.py:810-828: what production mode publishes -- boxes above `prediction_min_score`,
    back in lidar axes, lifted by `lift` metres.  Returns (centers, dims, angles, scores)."""
    a = remove_low_score(dt_anno, float(min_score))
    boxes_camera = np.concatenate([a["location"], a["dimensions"], a["rotation_y"][..., np.newaxis]], axis=1)
    boxes_lidar = box_camera_to_lidar(boxes_camera, r_rect, velo2cam)
    return boxes_lidar[:, :3] + [0.0, 0.0, lift], boxes_lidar[:, 3:6], boxes_lidar[:, 6], a["score"]


def save_results(dt_annos, out_dir, epoch_idx=None):
    """train.py:867-873: the detections of a whole evaluation run as one pickle (protocol 2):
    `result_epoch_<n>.pkl` during training-time evaluation, `result.pkl` otherwise.  Returns the path."""
    name = "result.pkl" if epoch_idx is None else "result_epoch_{}.pkl".format(str(epoch_idx))
    path = os.path.join(str(out_dir), name)
    with open(path, "wb") as f:
        pickle.dump(dt_annos, f, 2)
    return path


def load_results(path):
    with open(str(path), "rb") as f:
        return pickle.load(f)
