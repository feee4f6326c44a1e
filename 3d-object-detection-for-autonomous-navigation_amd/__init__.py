"""MI355X-native PointPillars inference path (drop-in for the hot path of
krullgit/3D-Object-Detection-for-autonomous-navigation's `train.py evaluate`).

The directory name is not a Python identifier; import it with
`importlib.import_module("3d-object-detection-for-autonomous-navigation_amd")`
or through the root-level alias module `pp_amd`.
"""
from . import config, anchors, weights, synth, frame_shard, anno, kitti_eval, ingest, target_assigner, optim, h5lite  # noqa: F401  (host-side modules)
from . import _lib  # noqa: F401  (ctypes binding of the C-ABI; loads lazily)
from .voxel_generator import points_to_voxel  # noqa: F401
from .engine import Engine, NumericError  # noqa: F401
from .voxelnet import VoxelNet  # noqa: F401
from .dataprep import prep_example, merge_batch  # noqa: F401
from .trainer import Trainer  # noqa: F401

__all__ = ["config", "anchors", "weights", "synth", "frame_shard", "anno", "kitti_eval", "ingest", "target_assigner", "optim", "points_to_voxel", "Engine", "VoxelNet", "Trainer",
           "prep_example", "merge_batch"]
