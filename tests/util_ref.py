"""Oracle end-to-end chain used by the GPU parity tests and smoke() (checker only)."""
import numpy as np

from oracle import c_oracle, nn_ref, ref_numpy as rn


def oracle_frames(d, frames):
    """voxelise + anchors mask per frame with the oracle; returns list of dicts."""
    anchors = rn.generate_anchors(d.feature_map_size, d.anchor_cfg).reshape(-1, 7)
    bv = rn.rbbox2d_to_near_bbox(anchors[:, [0, 1, 3, 4, 6]])
    cells = rn.anchor_cells(bv, d.voxel_size, d.pc_range, d.grid)
    out = []
    for pts in frames:
        v, c, n = c_oracle.points_to_voxel(pts, d.voxel_size, d.pc_range, d.max_points, d.max_voxels)
        m = c_oracle.anchor_mask(c, d.ny, d.nx, cells, float(d.anchor_area_threshold))
        out.append({"voxels": v, "coordinates": c, "num_points": n, "anchors_mask": m, "anchors": anchors})
    return out


def oracle_example(d, frames, rect, trv2c, p2):
    fr = oracle_frames(d, frames)
    voxels, num, coors = rn.merge_batch(fr)
    B = len(frames)
    ex = (voxels, num, coors, np.stack([rect] * B), np.stack([trv2c] * B), np.stack([p2] * B),
          np.stack([f["anchors"] for f in fr]), np.stack([f["anchors_mask"] for f in fr]).astype(np.uint8),
          np.arange(B), np.zeros((B, 2), np.int32))
    return ex, fr


def oracle_forward(d, w, ex, num_threads=None):
    B = ex[6].shape[0]
    return nn_ref.voxelnet_forward(ex[0], ex[1], ex[2], B, w, d.model_dict(), num_threads=num_threads)


def oracle_detect(d, w, frames, rect, trv2c, p2, num_threads=None):
    ex, fr = oracle_example(d, frames, rect, trv2c, p2)
    preds, canvas, feats = oracle_forward(d, w, ex, num_threads)
    dets = rn.predict(ex, preds, d.nms_dict())
    return {"example": ex, "frames": fr, "preds": preds, "canvas": canvas, "features": feats, "dets": dets}


def forced_decisions(trainer, example):
    """Trainer.decisions() (the GPU step's ReLU masks and PFN winners) in the form oracle/train_ref.py takes as `forced`:
    the PFN rows of the frames' real pillars, in the oracle example's pillar order (frame-major)."""
    dec = trainer.decisions()
    coors = example[2]
    B = dec["pfn"].shape[0]
    counts = [int((coors[:, 0] == b).sum()) for b in range(B)]
    out = {k: v for k, v in dec.items() if k != "pfn"}
    out["pfn"] = np.concatenate([dec["pfn"][b, :counts[b]] for b in range(B)], axis=0)
    return out
