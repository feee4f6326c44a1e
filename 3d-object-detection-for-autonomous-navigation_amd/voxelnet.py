"""VoxelNet with the reference's call surface, backed by the HIP engine.

Mirror of model/voxelnet.py::VoxelNet for the eval path (train.py:575-771):
    net = VoxelNet(config, writer, training=False)
    net.load_weights(path_or_dict)
    preds_dict = net(voxels, num_points, coors, batch_anchors)      # NHWC head maps
    predictions_dicts = net.predict(example, preds_dict)            # list of dicts
`example` is the positional 10-tuple (voxels, num_points, coordinates, rect,
Trv2c, P2, anchors, anchors_mask, image_idx, image_shape); elements may be
numpy arrays or anything with `.numpy()` (the reference passes TF tensors).
`detect(frames, ...)` is the fused raw-points path the reference does not have.
Training (training=True, labels/reg_targets) is out of scope and raises.
"""
import numpy as np

from . import weights as _weights
from .config import Derived
from .engine import Engine


def _np(x):
    return x.numpy() if hasattr(x, "numpy") else np.asarray(x)


class VoxelNet:
    def __init__(self, config, writer=None, training=False, max_batch=None, max_points_per_frame=32768, device=0):
        if training:
            raise NotImplementedError("only the inference path (training=False) is built (SURVEY section 8 scope)")
        self.config = config
        self.training = False
        self.d = Derived(config)
        self.batch_size = self.d.batch_size
        self.engine = Engine(self.d, max_batch=max_batch or self.batch_size,
                             max_points_per_frame=max_points_per_frame, device=device)
        self.box_code_size = 7

    # net.load_weights (train.py:731-734).  Accepts a dict name -> array (Keras
    # layouts, weights.py) or an .npz written by weights.save_npz.
    def load_weights(self, src):
        w = _weights.load_npz(src) if isinstance(src, str) else src
        self.engine.load_weights(w)

    def __call__(self, voxels, num_points, coors, batch_anchors, labels=None, reg_targets=None):
        if labels is not None or reg_targets is not None:
            raise NotImplementedError("training branch (labels / reg_targets) is out of scope")
        if not self.engine.weights_loaded:
            raise RuntimeError("VoxelNet: load_weights() has not been called")
        batch = int(_np(batch_anchors).shape[0])
        return self.engine.forward_voxels(_np(voxels), _np(num_points), _np(coors), batch)

    call = __call__

    def predict(self, example, preds_dict):
        rect, trv2c = _np(example[3]), _np(example[4])
        mask, img_idx = _np(example[7]), _np(example[8])
        batch = int(_np(example[6]).shape[0])
        dirp = _np(preds_dict["dir_cls_preds"]) if self.d.use_direction_classifier else None
        dets, n = self.engine.predict(_np(preds_dict["box_preds"]), _np(preds_dict["cls_preds"]), dirp, mask, rect, trv2c)
        return [self._to_dict(dets[b], int(n[b]), img_idx[b]) for b in range(batch)]

    def detect(self, frames, rect=None, trv2c=None, image_idx=None):
        """Fused path: list of raw clouds -> list of prediction dicts."""
        dets, n = self.engine.detect(frames, rect, trv2c)
        idx = image_idx if image_idx is not None else list(range(len(frames)))
        return [self._to_dict(dets[b], int(n[b]), idx[b]) for b in range(len(frames))]

    @staticmethod
    def _to_dict(dets, n, img_idx):
        # model/voxelnet.py:1362-1379: all-None dict (except batch_idx) when nothing survives
        if n == 0:
            return {"bbox": None, "box3d_camera": None, "box3d_lidar": None, "scores": None,
                    "label_preds": None, "batch_idx": img_idx}
        d = dets[:n]
        return {
            "bbox": np.tile(np.array([[400., 200., 500., 400.]]), (n, 1)),  # model/voxelnet.py:1357-1360
            "box3d_camera": d["box3d_camera"].copy(),
            "box3d_lidar": d["box3d_lidar"].copy(),
            "scores": d["score"].copy(),
            "label_preds": d["label"].astype(np.int64),
            "batch_idx": img_idx,
        }
