"""ctypes wrapper over oracle/pp_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpp_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pp_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_points_to_voxel.restype = ctypes.c_int
        _lib.oracle_nms_sorted.restype = ctypes.c_int
        _lib.oracle_anchor_mask.restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def points_to_voxel(points, voxel_size, coors_range, max_points, max_voxels):
    """Same contract as ref_numpy.points_to_voxel(..., reverse_index=True)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    rng = np.ascontiguousarray(coors_range, dtype=np.float64)
    vs = np.ascontiguousarray(voxel_size, dtype=np.float64)
    n, F = points.shape
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
    voxels = np.empty((max_voxels, max_points, F), dtype=np.float32)
    coors = np.empty((max_voxels, 3), dtype=np.int32)
    num = np.empty((max_voxels,), dtype=np.int32)
    scratch = np.empty((int(np.prod(grid)),), dtype=np.int32)
    P = lib().oracle_points_to_voxel(_p(points), ctypes.c_int(n), ctypes.c_int(F), _p(rng), _p(vs),
                                     ctypes.c_int(max_points), ctypes.c_int(max_voxels),
                                     _p(voxels), _p(coors), _p(num), _p(scratch))
    return voxels[:P].copy(), coors[:P].copy(), num[:P].copy()


def anchor_mask(coors, ny, nx, cells, threshold):
    coors = np.ascontiguousarray(coors, dtype=np.int32)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    dense = np.empty((ny * nx,), dtype=np.float32)
    mask = np.empty((cells.shape[0],), dtype=np.uint8)
    lib().oracle_anchor_mask(_p(coors), ctypes.c_int(coors.shape[0]), ctypes.c_int(ny), ctypes.c_int(nx),
                             _p(cells), ctypes.c_int(cells.shape[0]), ctypes.c_float(threshold),
                             _p(dense), _p(mask))
    return mask.astype(bool)


def nms_sorted(boxes_sorted, thresh):
    boxes_sorted = np.ascontiguousarray(boxes_sorted, dtype=np.float32)
    n = boxes_sorted.shape[0]
    keep = np.empty((max(n, 1),), dtype=np.int32)
    k = lib().oracle_nms_sorted(_p(boxes_sorted), ctypes.c_int(n), ctypes.c_float(thresh), _p(keep))
    return keep[:k].copy()


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """second/core/non_max_suppression/nms_gpu.py:618-653 (rotate_iou_gpu_eval): [N,5] x [K,5] -> [N,K]."""
    dtype = np.asarray(boxes).dtype
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    q = np.ascontiguousarray(query_boxes, dtype=np.float32)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    if b.shape[0] and q.shape[0]:
        L = lib()
        L.oracle_rotate_iou_eval.restype = None
        L.oracle_rotate_iou_eval(_p(b), ctypes.c_int64(b.shape[0]), _p(q), ctypes.c_int64(q.shape[0]),
                                 ctypes.c_int(criterion), _p(out))
    return out.astype(dtype)
