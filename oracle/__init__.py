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
  * rotated-rectangle IoU of the AP evaluator (`oracle_rotate_iou_eval`, pp_oracle.c): PINNED bit
    for bit against the reference's numba device functions executed as plain Python
    (tools/gen_golden_eval.py -> tests/golden/ref_rotate_iou.npz, ref_kitti_eval.npz).
  * the predict path's `+1` IoU (`iou_device`): formula PINNED to 1e-6 against the same kind of
    run (ref_iou_device.npz); the last bits follow numba's float32+int64 -> float64 typing rule,
    which a plain-Python run does not reproduce.
  * training loss at the head maps (`loss_ref.py`, torch-CPU restatement of model/voxelnet.py:922-1049):
    PARITY UNPINNED (TensorFlow graph in the reference); cross-checked against float64 numpy formulas and
    finite differences.
  * PFN / scatter / RPN (TensorFlow Keras layers), the numba-CUDA `nms_kernel` indexing
    and `VoxelNet.predict` glue: PARITY UNPINNED -- TensorFlow 2.2 and numba
    0.51 are not installable here and the reference ships no tests or golden
    vectors; these parts are restated from the source (file:line cited per
    function) and cross-checked between two independent restatements
    (torch-CPU vs plain numpy).
"""
