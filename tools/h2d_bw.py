"""Host-to-device copy rate of this box for the bench's 12.6 MB point batches: torch pinned tensors on 1-3 streams,
the library's own staging path (pp_host_alloc + pp_upload_points_async), and raw HIP allocations with different
hipHostMalloc flags (through ctypes on libamdhip64).  Prints one JSON line."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

out = {}
NB = 64 * 16384 * 12
n = NB // 4


def rate(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return NB * reps / (time.perf_counter() - t0) / 1e9


hs = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(3)]
ds = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
for k in (1, 2, 3):
    def go():
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                ds[i].copy_(hs[i], non_blocking=True)
    out[f"torch_pinned_{k}streams_GBps"] = rate(go) * k
pg = torch.empty(n, dtype=torch.float32)
out["torch_pageable_GBps"] = rate(lambda: ds[0].copy_(pg), reps=5)
# device-side read of pinned host memory (zero-copy): a kernel pulls the bytes over the link itself
mapped = hs[0]
try:
    hip = ctypes.CDLL("libamdhip64.so")
    dptr = ctypes.c_void_p()
    st = hip.hipHostGetDevicePointer(ctypes.byref(dptr), ctypes.c_void_p(mapped.data_ptr()), 0)
    out["hipHostGetDevicePointer_status"] = st
    flags = {"default": 0, "portable": 1, "mapped": 2, "writecombined": 4, "numa_user": 0x20000000,
             "coherent": 0x40000000, "noncoherent": 0x80000000}
    hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    for name, fl in flags.items():
        p = ctypes.c_void_p()
        if hip.hipHostMalloc(ctypes.byref(p), NB, fl) != 0:
            out[f"hip_{name}_GBps"] = None
            continue
        ctypes.memset(p, 1, NB)
        dst = ds[0].data_ptr()
        s = streams[0].cuda_stream

        def go2():
            hip.hipMemcpyAsync(ctypes.c_void_p(dst), p, NB, 1, ctypes.c_void_p(s))
        out[f"hip_{name}_GBps"] = rate(go2)
        hip.hipHostFree(p)
except Exception as ex:  # noqa: BLE001
    out["hip_error"] = repr(ex)
# kernel reading pinned memory directly (torch: copy through a mapped view is not exposed; use the engine's path)
import pp_amd as pp  # noqa: E402
eng = pp.Engine(pp.config.pedestrian_d435i_config(64), max_batch=64, max_points_per_frame=16384)
frames = [pp.synth.d435i_cloud(i) for i in range(64)]
st = eng.staging(frames)


def go3():
    eng.upload_async(st)
lib = eng._lib
eng.upload_async(st)
eng._check(lib.pp_sync(eng._h), "sync")
t0 = time.perf_counter()
for _ in range(20):
    eng.upload_async(st)
torch.cuda.synchronize()
out["pp_upload_points_async_GBps"] = NB * 20 / (time.perf_counter() - t0) / 1e9
d2h = torch.empty(n, dtype=torch.float32).pin_memory()
out["torch_d2h_pinned_GBps"] = rate(lambda: d2h.copy_(ds[0], non_blocking=True))
print(json.dumps(out))
