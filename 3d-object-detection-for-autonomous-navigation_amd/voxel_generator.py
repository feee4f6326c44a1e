"""points_to_voxel with the reference's signature, running on the GPU.

Mirror of load_data.py:695-771: `points_to_voxel(points, voxel_size,
coors_range, max_points, reverse_index, max_voxels) -> (voxels, coordinates,
num_points_per_voxel)`.  The HIP voxeliser (csrc/voxelize.hip) reproduces the
sequential first-appearance / arrival-order / break semantics bit for bit.
`reverse_index=True` is what the reference passes (load_data.py:2966); with False
(_points_to_voxel_kernel, load_data.py:643-692) the same pillars come back with (x, y, z) columns.
"""
import numpy as np

from . import config as _config
from .engine import Engine

_engines = {}


def _voxel_only_config(voxel_size, coors_range, max_points, max_voxels, num_features):
    """A reference-schema config whose network part is as small as the schema allows:
    only the voxeliser of the engine is used by points_to_voxel()."""
    cfg = _config.pedestrian_d435i_config()
    s = cfg["model"]["second"]
    v, r = [float(x) for x in voxel_size], [float(x) for x in coors_range]
    s["voxel_generator"].update(point_cloud_range=r, voxel_size=v,
                                max_number_of_points_per_voxel=int(max_points),
                                max_number_of_voxels=int(max_voxels))
    s["num_point_features"] = int(num_features)
    s["voxel_feature_extractor"]["num_filters"] = 32
    s["rpn"].update(layer_nums=[0, 0, 0], layer_strides=[1, 1, 1], num_filters=[32, 32, 32],
                    upsample_strides=[1, 1, 1], num_upsample_filters=[32, 32, 32])
    s["target_assigner"]["anchor_generators"]["anchor_generator_stride"].update(
        sizes=[v[0] / 2, v[1] / 2, 1.0], strides=[v[0], v[1], 0.0],
        offsets=[r[0] + v[0] / 2, r[1] + v[1] / 2, 0.0], rotations=[0])
    return cfg


def _engine_for(voxel_size, coors_range, max_points, max_voxels, num_features, n_points):
    key = (tuple(float(v) for v in voxel_size), tuple(float(v) for v in coors_range), int(max_points),
           int(max_voxels), int(num_features))
    cap = 1 << max(12, int(np.ceil(np.log2(max(n_points, 1)))))
    eng = _engines.get(key)
    if eng is None or eng.max_points_per_frame < n_points:
        if eng is not None:
            eng.close()
        eng = Engine(_voxel_only_config(key[0], key[1], max_points, max_voxels, num_features),
                     max_batch=1, max_points_per_frame=cap)
        _engines[key] = eng
    return eng


def points_to_voxel(points, voxel_size, coors_range, max_points, reverse_index, max_voxels):
    points = np.ascontiguousarray(points, dtype=np.float32)
    if points.ndim != 2 or points.shape[1] < 3:
        raise ValueError("points must be [N, >=3]")
    # load_data.py:726-729: non-ndarray sizes / ranges are cast to the points' dtype (float32)
    if not isinstance(voxel_size, np.ndarray):
        voxel_size = np.array(voxel_size, dtype=points.dtype)
    if not isinstance(coors_range, np.ndarray):
        coors_range = np.array(coors_range, dtype=points.dtype)
    eng = _engine_for(np.asarray(voxel_size, dtype=np.float64), np.asarray(coors_range, dtype=np.float64),
                      max_points, max_voxels, points.shape[1], points.shape[0])
    voxels, coors, num = eng.points_to_voxel(points)
    if not reverse_index:
        # _points_to_voxel_kernel (load_data.py:643-692) runs the same scan -- same cells, same first-appearance
        # pillar order, same break -- and only stores the cell as (x, y, z) instead of (z, y, x)
        coors = np.ascontiguousarray(coors[:, ::-1])
    return voxels, coors, num
