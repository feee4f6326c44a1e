"""CPU oracle for the PointPillars inference hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement of the
reference's algorithm (krullgit/3D-Object-Detection-for-autonomous-navigation)
for the path SURVEY.md section 8 names.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it, and only as the checker
or the timed CPU baseline -- never from the product package
(`3d-object-detection-for-autonomous-navigation_amd/`), which must fail loudly
when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * voxelise, anchors, anchor mask, box decode, corner/stand-up, NMS host sweep,
    lidar->camera: PINNED against outputs of the reference's own numpy code run
    in the build container (tools/gen_golden.py -> tests/golden/ref_*.npz).
  * PFN / scatter / RPN (TensorFlow Keras layers), the numba-CUDA NMS kernel
    and `VoxelNet.predict` glue: PARITY UNPINNED -- TensorFlow 2.2 and numba
    0.51 are not installable here and the reference ships no tests or golden
    vectors; these parts are restated from the source (file:line cited per
    function) and cross-checked between two independent restatements
    (torch-CPU vs plain numpy).
"""
