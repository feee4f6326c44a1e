"""ctypes binding of the C-ABI in include/pp_hip.h (libpp_hip.so, built in-tree by hipcc).

There is no CPU fallback: if the shared library is missing or cannot be
loaded, `lib()` raises, and every compute entry point needs a gfx950 device
(`pp_create` fails without one).  `build()` only needs hipcc (it cross-compiles
for gfx950 on a machine without a GPU).
"""
import ctypes
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_PKG), "include", "pp_hip.h")
# PP_HIP_LIB: load (and build into) another library file -- A/B builds of compile-time variants, e.g.
#   PP_HIP_LIB=libpp_hip_f16.so PP_HIPCC_EXTRA="-DPP_SPLIT_MODE=1" python -c "import pp_amd; pp_amd._lib.build()"
SO_PATH = os.path.join(_PKG, os.environ.get("PP_HIP_LIB", "libpp_hip.so"))
_VARIANT = os.path.splitext(os.path.basename(SO_PATH))[0]
SOURCES = ["pp_api.hip", "voxelize.hip", "pfn.hip", "anchor_mask.hip", "backbone.hip", "postprocess.hip",
           "rotate_iou.hip", "loss.hip", "optim.hip", "train.hip"]
# -fno-slp-vectorize: keeps f32 FMAs as v_fma_f32; the SLP vectoriser's v_pk_fma_f32 is slow on a SIMD
# that is also issuing MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
               "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]

EXPORTS = [
    "pp_abi_version", "pp_create", "pp_destroy", "pp_last_error", "pp_set_weight", "pp_finalize_weights",
    "pp_set_anchors", "pp_points_to_voxel", "pp_anchor_mask", "pp_forward_voxels", "pp_predict",
    "pp_upload_points", "pp_upload_points_async", "pp_host_alloc", "pp_host_free", "pp_upload_points_device",
    "pp_current_batch", "pp_set_calib", "pp_detect_async", "pp_sync",
    "pp_get_detections", "pp_set_gemm_precision", "pp_get_gemm_precision", "pp_set_cache_budget", "pp_detect", "pp_fetch_intermediates", "pp_set_profiling", "pp_get_kernel_times",
    "pp_timer_start", "pp_timer_stop", "pp_device_info", "pp_device_copy_bench", "pp_device_mem_free", "pp_bench_layer", "pp_layer_count", "pp_layer_tag",
    "pp_rotate_iou_eval", "pp_d3_box_overlap", "pp_head_loss", "pp_adamw_step_device",
    "pp_train_layout", "pp_train_layout_entry", "pp_train_step", "pp_train_step_async", "pp_train_step_wait",
    "pp_train_graph_stats", "pp_stream", "pp_train_fetch_decisions",
]


class PPConfig(ctypes.Structure):
    _fields_ = [
        ("pc_range", ctypes.c_double * 6),
        ("voxel_size", ctypes.c_double * 3),
        ("max_points", ctypes.c_int32),
        ("max_voxels", ctypes.c_int32),
        ("num_point_features", ctypes.c_int32),
        ("pfn_filters", ctypes.c_int32),
        ("layer_nums", ctypes.c_int32 * 3),
        ("layer_strides", ctypes.c_int32 * 3),
        ("num_filters", ctypes.c_int32 * 3),
        ("upsample_strides", ctypes.c_int32 * 3),
        ("num_upsample_filters", ctypes.c_int32 * 3),
        ("num_anchor_per_loc", ctypes.c_int32),
        ("num_class", ctypes.c_int32),
        ("nms_pre_max_size", ctypes.c_int32),
        ("nms_post_max_size", ctypes.c_int32),
        ("nms_score_threshold", ctypes.c_float),
        ("nms_iou_threshold", ctypes.c_float),
        ("anchor_area_threshold", ctypes.c_float),
        ("max_batch", ctypes.c_int32),
        ("max_points_per_frame", ctypes.c_int32),
        ("use_direction_classifier", ctypes.c_int32),
        ("with_distance", ctypes.c_int32),
    ]


class PPLossConfig(ctypes.Structure):
    _fields_ = [
        ("alpha", ctypes.c_float),
        ("gamma", ctypes.c_float),
        ("sigma", ctypes.c_float),
        ("code_weight", ctypes.c_float * 7),
        ("pos_class_weight", ctypes.c_float),
        ("neg_class_weight", ctypes.c_float),
        ("classification_weight", ctypes.c_float),
        ("localization_weight", ctypes.c_float),
        ("direction_loss_weight", ctypes.c_float),
        ("norm_by_num_positives", ctypes.c_int32),
        ("encode_rad_error_by_sin", ctypes.c_int32),
        ("use_direction_classifier", ctypes.c_int32),
    ]


class PPDetection(ctypes.Structure):
    _fields_ = [
        ("box3d_camera", ctypes.c_double * 7),
        ("box3d_lidar", ctypes.c_float * 7),
        ("score", ctypes.c_float),
        ("label", ctypes.c_int32),
        ("dir_label", ctypes.c_int32),
        ("anchor_index", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libpp_hip.so")


_OBJ = os.path.join(_CSRC, "_obj" if _VARIANT == "libpp_hip" else "_obj_" + _VARIANT)
_EXTRA = os.environ.get("PP_HIPCC_EXTRA", "").split()


def _deps():
    """Every header of csrc/ (and the C-ABI header): a change to any of them rebuilds every translation unit."""
    return sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")) + [_INCLUDE]


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(_CSRC, s) for s in SOURCES] + _deps()
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(src, verbose):
    """One translation unit -> csrc/_obj/<name>.o (skipped when the object is newer than its inputs)."""
    obj = os.path.join(_OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(_CSRC, src)
    if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [path] + _deps()):
        return obj
    tmp = f"{obj}.{os.getpid()}.tmp"
    cmd = [_hipcc()] + [f for f in HIPCC_FLAGS if f != "-shared"] + _EXTRA + ["-c", "-o", tmp, path]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, obj)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return obj


def build(force=False, verbose=False):
    """Compiles every HIP source for gfx950 (one object per translation unit, in parallel) and links
    <package>/libpp_hip.so."""
    if not force and not needs_build():
        return SO_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(_OBJ, exist_ok=True)
    if force:
        for f in os.listdir(_OBJ):
            os.remove(os.path.join(_OBJ, f))
    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, verbose), SOURCES))
    tmp = f"{SO_PATH}.{os.getpid()}.tmp.so"      # atomic: a concurrent loader never sees a half-written library
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, SO_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return SO_PATH


_lib = None


def _one_hip_runtime():
    """One HIP runtime per process, whatever the import order.  libpp_hip.so needs `libamdhip64.so.7`; the torch
    wheel bundles its own copy (same SONAME) that libtorch_hip.so asks for as plain `libamdhip64.so`.  torch imported
    first: its copy is mapped, the SONAME matches, libpp_hip.so binds to it.  libpp_hip.so first: the loader knows the
    system copy only as `libamdhip64.so.7`, a later `import torch` maps the bundled one beside it, and the runtime
    that initialises second finds no device (hipErrorNoDevice: "no ROCm-capable device is detected").  Loading the
    system library once under the plain name as well makes the later request resolve to the same mapping."""
    try:
        with open("/proc/self/maps") as f:
            if any("libamdhip64" in line for line in f):
                return
    except OSError:
        pass
    try:
        ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
    except OSError:
        pass            # not on the default search path: libpp_hip.so's own RUNPATH still finds its runtime


def lib():
    """Loads libpp_hip.so (raises if absent -- the HIP path is the only path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
    _one_hip_runtime()
    L = ctypes.CDLL(SO_PATH)
    vp, i32, i64, f32p = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    L.pp_abi_version.restype = ctypes.c_int
    L.pp_create.argtypes = [ctypes.POINTER(PPConfig), ctypes.c_int, ctypes.POINTER(vp)]
    L.pp_destroy.argtypes = [vp]
    L.pp_last_error.argtypes = [vp]
    L.pp_last_error.restype = ctypes.c_char_p
    L.pp_set_weight.argtypes = [vp, ctypes.c_char_p, f32p, ctypes.POINTER(i64), i32]
    L.pp_finalize_weights.argtypes = [vp]
    L.pp_set_anchors.argtypes = [vp, f32p, vp, i64]
    L.pp_points_to_voxel.argtypes = [vp, f32p, i64, f32p, vp, vp, ctypes.POINTER(i32)]
    L.pp_anchor_mask.argtypes = [vp, vp, i64, i32, vp]
    L.pp_forward_voxels.argtypes = [vp, f32p, vp, vp, i64, i32, f32p, f32p, f32p, f32p, f32p]
    L.pp_predict.argtypes = [vp, f32p, f32p, f32p, vp, f32p, f32p, i32, vp, vp]
    L.pp_upload_points.argtypes = [vp, f32p, vp, i32]
    L.pp_upload_points_async.argtypes = [vp, vp, vp, i32]
    L.pp_host_alloc.argtypes = [i64, ctypes.POINTER(vp)]
    L.pp_host_free.argtypes = [vp]
    L.pp_upload_points_device.argtypes = [vp, vp, vp, i32, vp]
    L.pp_current_batch.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.pp_set_calib.argtypes = [vp, f32p, f32p, i32]
    L.pp_detect_async.argtypes = [vp]
    L.pp_sync.argtypes = [vp]
    L.pp_get_detections.argtypes = [vp, vp, vp]
    L.pp_set_gemm_precision.argtypes = [vp, i32]
    L.pp_get_gemm_precision.argtypes = [vp, ctypes.POINTER(i32)]
    L.pp_set_cache_budget.argtypes = [vp, i32]
    L.pp_detect.argtypes = [vp, f32p, vp, i32, f32p, f32p, vp, vp]
    L.pp_fetch_intermediates.argtypes = [vp, vp, vp, vp, vp, f32p, f32p, f32p, f32p]
    L.pp_set_profiling.argtypes = [vp, i32]
    L.pp_get_kernel_times.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(i32)]
    L.pp_timer_start.argtypes = [vp]
    L.pp_timer_stop.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.pp_device_info.argtypes = [vp, ctypes.c_char_p, i32, ctypes.POINTER(i32), ctypes.POINTER(i64)]
    L.pp_device_copy_bench.argtypes = [vp, i64, i32, ctypes.POINTER(ctypes.c_float)]
    L.pp_device_mem_free.argtypes = [vp, ctypes.POINTER(i64)]
    L.pp_bench_layer.argtypes = [vp, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_float)]
    L.pp_layer_count.argtypes = [vp, ctypes.POINTER(i32)]
    L.pp_layer_tag.argtypes = [vp, i32]
    L.pp_layer_tag.restype = ctypes.c_char_p
    L.pp_rotate_iou_eval.argtypes = [ctypes.c_int, f32p, i64, f32p, i64, i32, f32p]
    L.pp_d3_box_overlap.argtypes = [ctypes.c_int, vp, i64, vp, i64, i32, vp]
    L.pp_head_loss.argtypes = [vp, vp, f32p, i32, ctypes.POINTER(PPLossConfig), f32p, f32p]
    L.pp_train_layout.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.pp_train_layout_entry.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i64), ctypes.POINTER(i64),
                                        ctypes.POINTER(i32)]
    L.pp_train_step.argtypes = [vp, vp, vp, vp, vp, f32p, i32, ctypes.POINTER(PPLossConfig), f32p]
    L.pp_train_step_async.argtypes = [vp, vp, vp, vp, vp, f32p, i32, ctypes.POINTER(PPLossConfig)]
    L.pp_train_step_wait.argtypes = [vp, f32p]
    L.pp_stream.argtypes = [vp, ctypes.POINTER(vp)]
    L.pp_train_graph_stats.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.pp_train_fetch_decisions.argtypes = [vp, i32, vp, i64, ctypes.POINTER(i64)]
    L.pp_adamw_step_device.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, i64, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_float]
    for name in EXPORTS:
        fn = getattr(L, name)  # raises AttributeError if the symbol is not exported
        if name not in ("pp_last_error", "pp_layer_tag"):
            fn.restype = ctypes.c_int
    _lib = L
    return L
