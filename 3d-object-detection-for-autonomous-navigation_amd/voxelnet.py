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

Training mode (model/voxelnet.py:922-1049 + train.py:265-304), `VoxelNet(config, writer, training=True)`:
    ret = net(voxels, num_points, coors, batch_anchors, labels, reg_targets)   # the reference's loss dict (scalars)
    net.apply_gradients(dist=None)        # optimizer.apply_gradients: one all-reduce over the ranks + AdamW
or, from raw clouds, `net.train_step(frames, labels, reg_targets, dist)`.  The forward pass, the loss and the
gradients of a call come from one `pp_train_step` (csrc/train.hip); the padded voxel tensor is unpadded on the
host into the pillar-ordered point list it was built from (the voxeliser then reproduces the same pillars).
"""
import os

import numpy as np

from . import weights as _weights
from .config import Derived
from .engine import Engine
from .trainer import Trainer


def _np(x):
    return x.numpy() if hasattr(x, "numpy") else np.asarray(x)


class VoxelNet:
    def __init__(self, config, writer=None, training=False, max_batch=None, max_points_per_frame=32768, device=0):
        self.config = config
        self.training = bool(training)
        self.d = Derived(config)
        self.batch_size = self.d.batch_size
        self.box_code_size = 7
        self.trainer = None
        self._ctor = dict(max_batch=max_batch or self.batch_size, max_points_per_frame=max_points_per_frame, device=device)
        if self.training:
            self.engine = None      # the Trainer (created by load_weights: it needs initial values) owns the engine
        else:
            self.engine = Engine(self.d, **self._ctor)

    # net.load_weights (train.py:731-734).  Accepts a dict name -> array (Keras layouts, weights.py), an .npz written by
    # weights.save_npz, or the reference's own checkpoint file (`model_weights_<epoch>.h5`, Keras save_weights,
    # train.py:407,436: weights.load_keras_h5, with h5py when it is installed and the built-in reader otherwise).
    def load_weights(self, src):
        w = _weights.load_any(src, self.d) if isinstance(src, (str, os.PathLike)) else src
        if self.training:
            if self.trainer is None:
                self.trainer = Trainer(self.config, w, **self._ctor)
                self.engine = self.trainer.engine
            else:
                self.trainer.set_weights(w)
            return
        self.engine.load_weights(w)

    def __call__(self, voxels, num_points, coors, batch_anchors, labels=None, reg_targets=None):
        if self.training:
            if labels is None or reg_targets is None:
                raise ValueError("training mode needs labels and reg_targets (model/voxelnet.py:850)")
            return self.train_step(self._unpad(_np(voxels), _np(num_points), _np(coors), int(_np(batch_anchors).shape[0])),
                                   _np(labels), _np(reg_targets), apply=False)
        if labels is not None or reg_targets is not None:
            raise ValueError("labels / reg_targets are training inputs: build the net with training=True")
        if not self.engine.weights_loaded:
            raise RuntimeError("VoxelNet: load_weights() has not been called")
        batch = int(_np(batch_anchors).shape[0])
        return self.engine.forward_voxels(_np(voxels), _np(num_points), _np(coors), batch)

    call = __call__

    # ---- training mode ----
    @staticmethod
    def _unpad(voxels, num_points, coors, batch):
        """Padded [P,T,F] + num_points + coors[b,z,y,x] -> per-frame point lists in pillar order."""
        frames = []
        for b in range(batch):
            rows = np.nonzero(coors[:, 0] == b)[0]
            frames.append(np.concatenate([voxels[p, :num_points[p]] for p in rows], axis=0).astype(np.float32)
                          if len(rows) else np.zeros((0, voxels.shape[2]), np.float32))
        return frames

    def train_step(self, frames, labels, reg_targets, dist=None, apply=True):
        """Forward (training mode) + loss + backward on raw clouds; apply=True also runs the optimizer step.
        Returns the reference's loss scalars (model/voxelnet.py:1032-1043)."""
        if self.trainer is None:
            raise RuntimeError("VoxelNet(training=True): load_weights() with the initial values first")
        out = self.trainer.forward_backward(frames, labels, reg_targets)
        if apply:
            self.apply_gradients(dist)
        return out

    def apply_gradients(self, dist=None):
        """optimizer.apply_gradients (train.py:301) on the flat buffers, after the data-parallel all-reduce."""
        self.trainer.apply_gradients(dist)      # ends with a stream synchronisation (see Trainer.apply_gradients)

    def get_weights(self):
        return self.trainer.weights() if self.training else None

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
