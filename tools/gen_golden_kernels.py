"""Golden vectors from the reference's numba-CUDA KERNELS themselves, run by the CUDA-model emulator of
tools/ref_shim.py (one Python thread per CUDA thread, a barrier for syncthreads, per-block shared arrays):

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_kernels.py      -> tests/golden/ref_cuda_kernels.npz

  nms_gpu -> nms_kernel -> nms_postprocess       libraries/eval_helper_functions.py:494-598 (the predict path's NMS:
            score sort, 64 x 64 block / thread indexing, shared-memory staging, bit masks, the host sweep)
  nms (pre / post caps around nms_gpu)            libraries/eval_helper_functions.py:463-492
  rotate_iou_gpu_eval -> rotate_iou_kernel_eval  second/core/non_max_suppression/nms_gpu.py:493-527, :618-653 (the AP
            evaluator's overlaps: its own block indexing -- rows on blockIdx.x -- and two shared tiles)

Box counts straddle the 64-thread blocks (1, 2, 63, 64, 65, 100, 129, 200).  Plain Python keeps `float32 + 1` in float32
where numba types it float64 (ref_shim docstring), so every NMS case is drawn until no pair's IoU lies within 1e-4 of the
threshold: the fixture pins indexing and decisions.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

_, ehf = ref_shim.load_reference()
ev, ng = ref_shim.load_reference_eval()
assert type(ehf.nms_kernel).__name__ == "_CudaKernel" and type(ng.rotate_iou_kernel_eval).__name__ == "_CudaKernel"


def iou64(a, b):
    """iou_device's formula in float64 on float32 differences (what numba evaluates)."""
    w = max(np.float64(np.float32(min(a[2], b[2]) - max(a[0], b[0]))) + 1.0, 0.0)
    h = max(np.float64(np.float32(min(a[3], b[3]) - max(a[1], b[1]))) + 1.0, 0.0)
    sa = (np.float64(np.float32(a[2] - a[0])) + 1.0) * (np.float64(np.float32(a[3] - a[1])) + 1.0)
    sb = (np.float64(np.float32(b[2] - b[0])) + 1.0) * (np.float64(np.float32(b[3] - b[1])) + 1.0)
    return w * h / (sa + sb - w * h)


def main():
    rng = np.random.default_rng(77)
    out = {}
    for thr in (0.5, 0.1, 0.7):
        for n in (1, 2, 63, 64, 65, 100, 129, 200):
            while True:
                # the `+1` convention makes metre-sized boxes overlap heavily: spread them so that all three thresholds decide
                c = rng.uniform(0, 14, (n, 2)).astype(np.float32)
                wh = rng.uniform(0.3, 4.0, (n, 2)).astype(np.float32)
                dets = np.concatenate([c - wh / 2, c + wh / 2, rng.permutation(n).astype(np.float32)[:, None] / n], axis=1)
                ious = np.array([[iou64(dets[i], dets[j]) for j in range(n)] for i in range(n)])
                if np.min(np.abs(ious - thr)) > 1e-4:
                    break
            keep = np.array(ehf.nms_gpu(dets.copy(), np.float32(thr)), dtype=np.int32)
            out[f"nms_t{thr}_n{n}_dets"] = dets
            out[f"nms_t{thr}_n{n}_keep"] = keep
            print(f"nms_gpu thr {thr} n {n}: {len(keep)} kept")
    # nms() itself (libraries/eval_helper_functions.py:463-492): top pre_max_size by np.argpartition, nms_gpu, post_max_size.
    # It indexes with the numpy-1.19 idiom `scores[[indices]]` (a list holding ONE array used to mean that array; numpy >= 1.23
    # reads it as a 2-D index): the inputs are handed over as an ndarray subclass that keeps the old meaning, the
    # reference's lines run unchanged.
    class Legacy(np.ndarray):
        def __getitem__(self, key):
            if isinstance(key, list) and len(key) == 1 and isinstance(key[0], np.ndarray):
                key = key[0]
            return super().__getitem__(key)
    k = 0
    for n, pre, post, thr in ((300, 100, 50, 0.5), (300, 1000, 100, 0.5), (90, 100, 300, 0.1), (150, 64, 5, 0.7), (40, None, 10, 0.5),
                              (0, 100, 50, 0.5)):
        while True:
            c = rng.uniform(0, 14, (n, 2)).astype(np.float32)
            wh = rng.uniform(0.3, 4.0, (n, 2)).astype(np.float32)
            boxes = np.concatenate([c - wh / 2, c + wh / 2], axis=1)
            scores = (rng.permutation(n).astype(np.float32) + 0.5) / max(n, 1)
            ious = np.array([[iou64(boxes[i], boxes[j]) for j in range(n)] for i in range(n)]) if n else np.ones((1, 1))
            if np.min(np.abs(ious - thr)) > 1e-4:
                break
        got = ehf.nms(boxes.view(Legacy), scores.view(Legacy), pre_max_size=pre, post_max_size=post, iou_threshold=np.float32(thr))
        out[f"nmsfn_{k}_boxes"], out[f"nmsfn_{k}_scores"] = boxes, scores
        out[f"nmsfn_{k}_args"] = np.array([-1 if pre is None else pre, post, thr], dtype=np.float64)
        out[f"nmsfn_{k}_keep"] = np.zeros((0,), np.int64) if got is None else np.asarray(got, dtype=np.int64)
        out[f"nmsfn_{k}_none"] = np.array(got is None)
        print(f"nms() n {n} pre {pre} post {post} thr {thr}: {'None' if got is None else len(got)}")
        k += 1
    out["nmsfn_count"] = np.array(k)
    b = np.concatenate([rng.uniform(-3, 3, (70, 2)), rng.uniform(0.3, 2.5, (70, 2)), rng.uniform(-3.5, 3.5, (70, 1))], axis=1).astype(np.float32)
    q = np.concatenate([rng.uniform(-3, 3, (130, 2)), rng.uniform(0.3, 2.5, (130, 2)), rng.uniform(-3.5, 3.5, (130, 1))], axis=1).astype(np.float32)
    out["riou_boxes"], out["riou_qboxes"] = b, q
    for crit in (-1, 1):
        got = ng.rotate_iou_gpu_eval(b, q, crit)
        # the kernel's indexing against the plain double loop over the same device function
        loop = np.array([[ng.devRotateIoUEval(q[k], b[i], crit) for k in range(q.shape[0])] for i in range(b.shape[0])], dtype=np.float32)
        assert np.array_equal(got, loop)
        out[f"riou_c{crit}"] = got
        print(f"rotate_iou_gpu_eval criterion {crit}: {got.shape}, {(got > 0).sum()} overlapping pairs")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_cuda_kernels.npz"), **out)


if __name__ == "__main__":
    main()
