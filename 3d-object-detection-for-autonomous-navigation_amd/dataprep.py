"""Eval-side data preparation with the reference's products, computed on the GPU.

Mirror of the eval branch of dataLoader.prep_pointcloud (load_data.py:2543,
:2966-3072) and merge_second_batch (:2164-2224): voxelise, attach the static
anchors, compute anchors_mask, collate frames.  The example is returned both
as the reference's dict and as the positional 10-tuple `VoxelNet.predict`
indexes (train.py:689: voxels, num_points, coordinates, rect, Trv2c, P2,
anchors, anchors_mask, image_idx, image_shape).
"""
import numpy as np


def prep_example(engine, points, rect, trv2c, p2, image_idx=0, image_shape=(375, 1242)):
    """One frame -> reference-shaped example dict (eval mode)."""
    voxels, coors, num = engine.points_to_voxel(points)
    ex = {
        "voxels": voxels, "num_points": num, "coordinates": coors,
        "num_voxels": np.array([voxels.shape[0]], dtype=np.int64),
        "rect": rect, "Trv2c": trv2c, "P2": p2,
        "anchors": engine.anchors,
        "image_idx": image_idx, "image_shape": np.asarray(image_shape, dtype=np.int32),
    }
    if engine.d.anchor_area_threshold is not None and engine.d.anchor_area_threshold >= 0:
        c4 = np.concatenate([np.zeros((coors.shape[0], 1), np.int32), coors], axis=1)
        ex["anchors_mask"] = engine.anchor_mask(c4, 1)[0].astype(bool)
    return ex


def merge_batch(examples):
    """merge_second_batch (load_data.py:2164-2224) -> the positional 10-tuple."""
    voxels = np.concatenate([e["voxels"] for e in examples], axis=0)
    num_points = np.concatenate([e["num_points"] for e in examples], axis=0)
    coors = np.concatenate([np.pad(e["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=i)
                            for i, e in enumerate(examples)], axis=0)

    def stack(k):
        return np.stack([e[k] for e in examples], axis=0)

    mask = stack("anchors_mask").astype(np.uint8)  # load_data.py:2514-2515
    return (voxels, num_points, coors, stack("rect"), stack("Trv2c"), stack("P2"), stack("anchors"), mask,
            np.asarray([e["image_idx"] for e in examples]), stack("image_shape"))
