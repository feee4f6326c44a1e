"""Generates tests/golden/ref_voxel_fwd.npz by RUNNING THE REFERENCE's own numpy code
(load_data.points_to_voxel with reverse_index=False -> _points_to_voxel_kernel, load_data.py:643-692)
on the points of two existing fixture cases.  Build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_fwd.py
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
import pp_amd  # noqa: E402  (config only)

g = np.load(os.path.join(ROOT, "tests", "golden", "ref_voxel.npz"))
dA = pp_amd.config.Derived(pp_amd.config.pedestrian_d435i_config())
out = {}
for case, (T, MV) in (("a2k", (50, 12000)), ("brk", (5, 300))):
    pts = g[case + "_points"]
    v, c, n = ld.points_to_voxel(pts, dA.voxel_size, dA.pc_range, T, False, MV)
    out[case + "_params"] = np.array([T, MV], dtype=np.int64)
    out[case + "_voxels"], out[case + "_coors"], out[case + "_num"] = v, c, n
    # same pillars, coordinate columns reversed
    assert np.array_equal(c[:, ::-1], g[case + "_coors"]) and np.array_equal(n, g[case + "_num"])
    print(case, v.shape, c[:3].tolist())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_voxel_fwd.npz"), **out)
