"""Engine: one handle of the HIP library per GPU, numpy in / numpy out.

Thin host layer over the C-ABI (include/pp_hip.h).  Owns nothing numerical:
every stage runs in libpp_hip.so on the GPU; a failing call raises
RuntimeError carrying pp_last_error().
"""
import collections
import ctypes
import weakref

import numpy as np

from . import _lib
from .anchors import build_anchor_cells, build_anchors
from .config import Derived
from .weights import check_weights

_STATUS = {1: "PP_ERR_ARG", 2: "PP_ERR_STATE", 3: "PP_ERR_HIP", 4: "PP_ERR_SHAPE", 5: "PP_ERR_UNSUPPORTED",
           6: "PP_ERR_NUMERIC"}
PP_ERR_NUMERIC = 6
_PRECISIONS = {"split_f16": 0, "f32": 1}


class NumericError(RuntimeError):
    """PP_ERR_NUMERIC: a frame's head maps hold a non-finite value (an activation beyond the float16 operand pieces'
    range in the default arithmetic, or a network that overflows float32).  No detections were handed out."""

DET_DTYPE = np.dtype([
    ("box3d_camera", np.float64, (7,)), ("box3d_lidar", np.float32, (7,)), ("score", np.float32),
    ("label", np.int32), ("dir_label", np.int32), ("anchor_index", np.int32), ("reserved", np.int32)],
    align=True)
assert DET_DTYPE.itemsize == ctypes.sizeof(_lib.PPDetection)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Staging:
    """A page-locked host buffer (pp_host_alloc) viewed as a flat float32 numpy array."""

    def __init__(self, lib, nbytes):
        self._lib = lib
        p = ctypes.c_void_p()
        st = lib.pp_host_alloc(ctypes.c_int64(nbytes), ctypes.byref(p))
        if st != 0:
            raise RuntimeError(f"pp_host_alloc({nbytes}) failed ({_STATUS.get(st, st)})")
        self._p = p
        self.array = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(max(nbytes // 4, 1),))
        self.points = None
        self.offsets = None
        self._users = weakref.WeakSet()     # engines that were handed this buffer (zero-copy passes read it late)

    def close(self):
        """Frees the buffer -- after every engine that was fed from it has finished the pass that reads it (small
        batches are not copied: the pass's first kernel reads this memory over the host link)."""
        if self._p:
            for eng in list(self._users):
                if getattr(eng, "_h", None):
                    eng.sync()
            self._users.clear()
            self.array = self.points = None
            self._lib.pp_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedArray:
    """A page-locked host buffer (pp_host_alloc) viewed as a numpy array of the given shape / dtype: what a data
    loader fills in place so that the host-to-device copy of a training batch's targets is one DMA, not a staged
    pageable copy."""

    def __init__(self, lib, shape, dtype):
        self._lib = lib
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = ctypes.c_void_p()
        st = lib.pp_host_alloc(ctypes.c_int64(max(n * dt.itemsize, 4)), ctypes.byref(p))
        if st != 0:
            raise RuntimeError(f"pp_host_alloc({n * dt.itemsize}) failed ({_STATUS.get(st, st)})")
        self._p = p
        raw = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(max(n * dt.itemsize, 4),))
        self.array = raw[:n * dt.itemsize].view(dt).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            self._lib.pp_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """config: reference-schema dict (or a config.Derived).  max_batch /
    max_points_per_frame size the device workspaces."""

    def __init__(self, config, max_batch=None, max_points_per_frame=32768, device=0, weights=None):
        self.d = config if isinstance(config, Derived) else Derived(config)
        d = self.d
        self.max_batch = int(max_batch if max_batch is not None else d.batch_size)
        self.max_points_per_frame = int(max_points_per_frame)
        self._lib = _lib.lib()
        c = _lib.PPConfig()
        c.pc_range[:] = [float(v) for v in d.pc_range]
        c.voxel_size[:] = [float(v) for v in d.voxel_size]
        c.max_points, c.max_voxels = d.max_points, d.max_voxels
        c.num_point_features, c.pfn_filters = d.num_point_features, d.pfn_filters
        c.layer_nums[:] = d.layer_nums
        c.layer_strides[:] = d.layer_strides
        c.num_filters[:] = d.num_filters
        c.upsample_strides[:] = d.upsample_strides
        c.num_upsample_filters[:] = d.num_upsample_filters
        c.num_anchor_per_loc, c.num_class = d.num_anchor_per_loc, d.num_class
        c.nms_pre_max_size, c.nms_post_max_size = d.nms_pre_max_size, d.nms_post_max_size
        c.nms_score_threshold, c.nms_iou_threshold = d.nms_score_threshold, d.nms_iou_threshold
        thr = d.anchor_area_threshold
        c.anchor_area_threshold = float(thr) if thr is not None else -1.0
        c.max_batch, c.max_points_per_frame = self.max_batch, self.max_points_per_frame
        c.use_direction_classifier = 1 if d.use_direction_classifier else 0
        c.with_distance = 1 if d.with_distance else 0
        h = ctypes.c_void_p()
        st = self._lib.pp_create(ctypes.byref(c), int(device), ctypes.byref(h))
        if st != 0:
            msg = self._lib.pp_last_error(None)
            raise RuntimeError(f"pp_create failed ({_STATUS.get(st, st)}): {msg.decode() if msg else ''}")
        self._h = h
        self._train_targets = None      # labels / regression targets of a step in flight (train_step_async)
        self._staged = collections.deque(maxlen=2)   # Stagings of the last two upload_async calls (see there)
        self.anchors = build_anchors(d)
        self.anchor_cells = build_anchor_cells(self.anchors, d)
        self._check(self._lib.pp_set_anchors(self._h, _ptr(self.anchors), _ptr(self.anchor_cells),
                                             ctypes.c_int64(self.anchors.shape[0])), "pp_set_anchors")
        self.weights_loaded = False
        if weights is not None:
            self.load_weights(weights)

    @staticmethod
    def det_dtype():
        return DET_DTYPE

    # ---- plumbing ----
    def _check(self, st, what):
        if st != 0:
            msg = self._lib.pp_last_error(self._h)
            cls = NumericError if st == PP_ERR_NUMERIC else RuntimeError
            raise cls(f"{what} failed ({_STATUS.get(st, st)}): {msg.decode() if msg else ''}")

    # ---- GEMM arithmetic (pp_set_gemm_precision) ----
    def set_gemm_precision(self, precision):
        """'split_f16' (default: fp32 results from two float16 pieces per operand on the 16-bit matrix pipe) or 'f32'
        (the float32 matrix instruction everywhere: float32's range, about a sixth of the matrix rate)."""
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self._check(self._lib.pp_set_gemm_precision(self._h, _PRECISIONS[precision]), "pp_set_gemm_precision")

    def set_cache_budget(self, megabytes):
        """Last-level-cache budget of a pass in MiB (pp_set_cache_budget; default 256, 0 = off): layers whose maps exceed
        it run over sub-ranges of the batch's frames.  Right for one engine in flight per GPU; set 0 when several
        engines share the GPU (their working sets evict each other)."""
        self._check(self._lib.pp_set_cache_budget(self._h, int(megabytes)), "pp_set_cache_budget")

    def gemm_precision(self):
        v = ctypes.c_int32(0)
        self._check(self._lib.pp_get_gemm_precision(self._h, ctypes.byref(v)), "pp_get_gemm_precision")
        return {v_: k for k, v_ in _PRECISIONS.items()}[v.value]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pp_destroy(self._h)      # waits for the handle's streams
            self._h = None
        if getattr(self, "_staged", None) is not None:
            self._staged.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights (net.load_weights, train.py:731-734) ----
    def load_weights(self, w):
        check_weights(self.d, w)
        for name, arr in w.items():
            a = _f32(arr)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            self._check(self._lib.pp_set_weight(self._h, name.encode(), _ptr(a), shape, a.ndim), f"pp_set_weight({name})")
        self._check(self._lib.pp_finalize_weights(self._h), "pp_finalize_weights")
        self.weights_loaded = True

    # ---- a1 ----
    def points_to_voxel(self, points):
        d = self.d
        points = _f32(points)
        if points.ndim != 2 or points.shape[1] != d.num_point_features:
            raise ValueError(f"points must be [N,{d.num_point_features}], got {points.shape}")
        voxels = np.empty((d.max_voxels, d.max_points, d.num_point_features), dtype=np.float32)
        coors = np.empty((d.max_voxels, 3), dtype=np.int32)
        num = np.empty((d.max_voxels,), dtype=np.int32)
        P = ctypes.c_int32(0)
        self._check(self._lib.pp_points_to_voxel(self._h, _ptr(points), ctypes.c_int64(points.shape[0]),
                                                 _ptr(voxels), _ptr(coors), _ptr(num), ctypes.byref(P)),
                    "pp_points_to_voxel")
        n = P.value
        return voxels[:n].copy(), coors[:n].copy(), num[:n].copy()

    # ---- a4 ----
    def anchor_mask(self, coors4, batch):
        coors4 = _i32(coors4).reshape(-1, 4)
        mask = np.empty((batch, self.anchors.shape[0]), dtype=np.uint8)
        self._check(self._lib.pp_anchor_mask(self._h, _ptr(coors4), ctypes.c_int64(coors4.shape[0]), int(batch),
                                             _ptr(mask)), "pp_anchor_mask")
        return mask

    # ---- a5-a7, a13 ----
    def forward_voxels(self, voxels, num_points, coors4, batch, want_features=False, want_canvas=False):
        d = self.d
        voxels, num_points, coors4 = _f32(voxels), _i32(num_points), _i32(coors4)
        P = voxels.shape[0]
        if voxels.shape[1:] != (d.max_points, d.num_point_features):
            raise ValueError(f"voxels must be [P,{d.max_points},{d.num_point_features}], got {voxels.shape}")
        if num_points.shape != (P,) or coors4.shape != (P, 4):
            raise ValueError("num_points must be [P] and coors [P,4] (b,z,y,x)")
        H, W, k = d.head_h, d.head_w, d.num_anchor_per_loc
        box = np.empty((batch, H, W, k * 7), dtype=np.float32)
        cls = np.empty((batch, H, W, k * d.num_class), dtype=np.float32)
        dr = np.empty((batch, H, W, k * 2), dtype=np.float32) if d.use_direction_classifier else None
        feat = np.empty((P, d.pfn_filters), dtype=np.float32) if want_features else None
        canvas = np.empty((batch, d.ny, d.nx, d.pfn_filters), dtype=np.float32) if want_canvas else None
        self._check(self._lib.pp_forward_voxels(self._h, _ptr(voxels), _ptr(num_points), _ptr(coors4),
                                                ctypes.c_int64(P), int(batch), _ptr(box), _ptr(cls), _ptr(dr),
                                                _ptr(feat), _ptr(canvas)), "pp_forward_voxels")
        out = {"box_preds": box, "cls_preds": cls}
        if dr is not None:             # model/voxelnet.py:714: the key exists only with the direction head
            out["dir_cls_preds"] = dr
        if want_features:
            out["pillar_features"] = feat
        if want_canvas:
            out["canvas"] = canvas
        return out

    # ---- a8-a12 ----
    def predict(self, box_preds, cls_preds, dir_cls_preds, anchors_mask, rect, trv2c):
        batch = box_preds.shape[0]
        post = self.d.nms_post_max_size
        dets = np.zeros((batch, post), dtype=DET_DTYPE)
        n = np.zeros((batch,), dtype=np.int32)
        m = np.ascontiguousarray(anchors_mask, dtype=np.uint8).reshape(batch, -1)
        if m.shape[1] != self.anchors.shape[0]:
            raise ValueError(f"anchors_mask must be [batch,{self.anchors.shape[0]}]")
        dirp = _f32(dir_cls_preds) if self.d.use_direction_classifier else None
        self._check(self._lib.pp_predict(self._h, _ptr(_f32(box_preds)), _ptr(_f32(cls_preds)), _ptr(dirp),
                                         _ptr(m), _ptr(_f32(rect).reshape(batch, 16)), _ptr(_f32(trv2c).reshape(batch, 16)),
                                         int(batch), _ptr(dets), _ptr(n)), "pp_predict")
        return dets, n

    # ---- fused path ----
    @staticmethod
    def _pack(frames, F):
        offs = np.zeros((len(frames) + 1,), dtype=np.int32)
        for i, f in enumerate(frames):
            if f.ndim != 2 or f.shape[1] != F:
                raise ValueError(f"frame {i} must be [N,{F}], got {f.shape}")
            offs[i + 1] = offs[i] + f.shape[0]
        pts = np.concatenate([_f32(f) for f in frames], axis=0) if frames else np.zeros((0, F), np.float32)
        return np.ascontiguousarray(pts), offs

    def upload(self, frames, rect=None, trv2c=None):
        """frames: list of [N_b,F] float32 clouds -> engine's resident input buffer."""
        pts, offs = self._pack(frames, self.d.num_point_features)
        self._check(self._lib.pp_upload_points(self._h, _ptr(pts), _ptr(offs), len(frames)), "pp_upload_points")
        if rect is not None:
            self.set_calib(rect, trv2c, len(frames))

    def set_calib(self, rect, trv2c, batch):
        self._check(self._lib.pp_set_calib(self._h, _ptr(_f32(rect).reshape(batch, 16)),
                                           _ptr(_f32(trv2c).reshape(batch, 16)), batch), "pp_set_calib")

    def staging(self, frames):
        """Packs frames into a page-locked staging buffer (pp_host_alloc) for upload_async.  Returns a
        Staging whose `.points` / `.offsets` may be refilled in place between uses."""
        pts, offs = self._pack(frames, self.d.num_point_features)
        st = Staging(self._lib, max(pts.nbytes, 4))
        st.points = st.array[:pts.size].reshape(pts.shape)
        st.points[...] = pts
        st.offsets = offs
        return st

    def upload_async(self, staging):
        """Queues the host-to-device copy of a Staging on the engine's stream and returns at once; the staging
        buffer must not be rewritten before the sync() that follows the detect_async() consuming it."""
        self._check(self._lib.pp_upload_points_async(self._h, _ptr(staging.points), _ptr(staging.offsets),
                                                     staging.offsets.shape[0] - 1), "pp_upload_points_async")
        # The handle's two input buffers alternate and the library waits for the pass that read a buffer before it
        # re-stages it, so the buffers of the last two uploads are the ones a pass may still read: the engine keeps
        # them alive (a temporary Staging would otherwise be freed under the zero-copy kernel).
        users = getattr(staging, "_users", None)     # (anything with .points / .offsets may be passed)
        if users is not None:
            users.add(self)
        self._staged.append(staging)

    def upload_device(self, dev_ptr, offsets, producer_stream=None):
        """dev_ptr: integer device address of concatenated [sum N, F] float32 points.  producer_stream: integer
        hipStream_t handle the points were written on (e.g. torch.cuda.current_stream().cuda_stream), or None
        if that work has already completed."""
        offs = _i32(offsets)
        ps = ctypes.c_void_p(int(producer_stream)) if producer_stream else None
        self._check(self._lib.pp_upload_points_device(self._h, ctypes.c_void_p(int(dev_ptr)), _ptr(offs),
                                                      offs.shape[0] - 1, ps), "pp_upload_points_device")

    def _batches(self):
        up, res = ctypes.c_int32(0), ctypes.c_int32(0)
        self._check(self._lib.pp_current_batch(self._h, ctypes.byref(up), ctypes.byref(res)), "pp_current_batch")
        return up.value, res.value

    def detect_async(self):
        self._check(self._lib.pp_detect_async(self._h), "pp_detect_async")

    def sync(self):
        self._check(self._lib.pp_sync(self._h), "pp_sync")

    def detections(self, out=None):
        """Results of the last detect_async (waits for it).  out: optional (dets, n) arrays to fill."""
        B, post = self._batches()[1], self.d.nms_post_max_size
        if out is None:
            out = (np.zeros((max(B, 1), post), dtype=DET_DTYPE), np.zeros((max(B, 1),), dtype=np.int32))
        dets, n = out
        if dets.shape[0] < B or n.shape[0] < B:
            raise ValueError("detections(out=...): arrays smaller than the batch")
        self._check(self._lib.pp_get_detections(self._h, _ptr(dets), _ptr(n)), "pp_get_detections")
        return dets, n

    def detect(self, frames, rect=None, trv2c=None, on_numeric="f32"):
        """upload + detect_async + sync + detections.  on_numeric: what to do when the default arithmetic reports
        activations outside the float16 pieces' range (NumericError) -- "f32": switch this engine to the float32
        matrix instruction (it stays there: `gemm_precision()`), run the still-resident frames again and return those
        results; "raise": propagate.  A network that overflows float32 itself always raises."""
        self.upload(frames, rect, trv2c)
        self.detect_async()
        self.sync()
        try:
            return self.detections()
        except NumericError:
            if on_numeric != "f32" or self.gemm_precision() == "f32":
                raise
        self.set_gemm_precision("f32")
        self.detect_async()
        self.sync()
        return self.detections()

    def intermediates(self, canvas=False):
        d, B = self.d, max(self._batches()[1], 1)
        H, W, k = d.head_h, d.head_w, d.num_anchor_per_loc
        out = {
            "n_pillars": np.zeros((B,), np.int32),
            "coors": np.zeros((B, d.max_voxels, 3), np.int32),
            "num_points": np.zeros((B, d.max_voxels), np.int32),
            "anchors_mask": np.zeros((B, self.anchors.shape[0]), np.uint8),
            "box_preds": np.zeros((B, H, W, k * 7), np.float32),
            "cls_preds": np.zeros((B, H, W, k * d.num_class), np.float32),
            "dir_cls_preds": np.zeros((B, H, W, k * 2), np.float32) if d.use_direction_classifier else None,
        }
        cv = np.zeros((B, d.ny, d.nx, d.pfn_filters), np.float32) if canvas else None
        self._check(self._lib.pp_fetch_intermediates(
            self._h, _ptr(out["n_pillars"]), _ptr(out["coors"]), _ptr(out["num_points"]), _ptr(out["anchors_mask"]),
            _ptr(out["box_preds"]), _ptr(out["cls_preds"]), _ptr(out["dir_cls_preds"]), _ptr(cv)),
            "pp_fetch_intermediates")
        if canvas:
            out["canvas"] = cv
        if out["dir_cls_preds"] is None:
            del out["dir_cls_preds"]
        return out

    # ---- measurement ----
    def set_profiling(self, on):
        self._check(self._lib.pp_set_profiling(self._h, 1 if on else 0), "pp_set_profiling")

    def kernel_times(self):
        cap = 2048
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        n = ctypes.c_int32(0)
        self._check(self._lib.pp_get_kernel_times(self._h, cap, names, ms, ctypes.byref(n)), "pp_get_kernel_times")
        return [(names[i].decode(), float(ms[i])) for i in range(min(n.value, cap))]

    def layer_tags(self):
        n = ctypes.c_int32(0)
        self._check(self._lib.pp_layer_count(self._h, ctypes.byref(n)), "pp_layer_count")
        return [self._lib.pp_layer_tag(self._h, i).decode() for i in range(n.value)]

    def bench_layer(self, layer, batch, reps=20, ablate=0):
        t = ctypes.c_float(0)
        self._check(self._lib.pp_bench_layer(self._h, int(layer), int(batch), int(reps), int(ablate), ctypes.byref(t)),
                    "pp_bench_layer")
        return float(t.value)

    def loss_config(self):
        """The reference's loss keys (model.second.loss..., configs/train.yaml:147-167) as the C-ABI struct."""
        s = self.d.config["model"]["second"]
        lc = _lib.PPLossConfig()
        focal = s["loss"]["classification_loss"]["weighted_sigmoid_focal"]
        l1 = s["loss"]["localization_loss"]["weighted_smooth_l1"]
        lc.alpha = -1.0 if focal["alpha"] is None else float(focal["alpha"])
        lc.gamma = float(focal["gamma"] or 0.0)
        lc.sigma = float(l1["sigma"])
        for i, v in enumerate(l1["code_weight"]):
            lc.code_weight[i] = float(v)
        lc.pos_class_weight = float(s["pos_class_weight"])
        lc.neg_class_weight = float(s["neg_class_weight"])
        lc.classification_weight = float(s["loss"]["classification_weight"])
        lc.localization_weight = float(s["loss"]["localization_weight"])
        lc.direction_loss_weight = float(s["direction_loss_weight"])
        lc.norm_by_num_positives = 1 if s["loss_norm_type"] == "NormByNumPositives" else 0
        lc.encode_rad_error_by_sin = 1 if s["encode_rad_error_by_sin"] else 0
        lc.use_direction_classifier = 1 if s["use_direction_classifier"] else 0
        return lc

    def head_loss(self, labels, reg_targets, want_grad=True):
        """Training loss of the head maps the last forward pass left on the device (VoxelNet.call in training
        mode, model/voxelnet.py:922-1049) and its gradient with respect to them.  labels [B, A] int32,
        reg_targets [B, A, 7] float32 (the dataloader's `labels` / `reg_targets`).  Returns the reference's
        scalar keys and, if asked, `box_preds_grad` / `cls_preds_grad` / `dir_cls_preds_grad` shaped like
        the head maps."""
        labels = _i32(np.asarray(labels))
        batch = labels.shape[0]
        reg_targets = _f32(np.asarray(reg_targets).reshape(batch, self.d.num_anchors, 7))
        if labels.shape != (batch, self.d.num_anchors):
            raise ValueError(f"labels must be [B, {self.d.num_anchors}]")
        losses = np.zeros(8, np.float32)
        hh, hw = self.d.head_h, self.d.head_w
        grad = np.zeros((batch, hh * hw, 32), np.float32) if want_grad else None
        lc = self.loss_config()
        self._check(self._lib.pp_head_loss(self._h, _ptr(labels), _ptr(reg_targets), batch, ctypes.byref(lc),
                                           _ptr(losses), _ptr(grad) if want_grad else None), "pp_head_loss")
        out = {"loss": float(losses[0]), "loc_loss_reduced": float(losses[1]), "cls_loss_reduced": float(losses[2]),
               "dir_loss_reduced": float(losses[3]), "cls_pos_loss": float(losses[4]), "cls_neg_loss": float(losses[5]),
               "num_positives": int(losses[6])}
        if want_grad:
            na = self.d.num_anchor_per_loc
            nb, nc = na * 7, na * self.d.num_class      # head row: [box na*7 | cls na*num_class | dir na*2 | pad]
            out["head_grad"] = grad
            out["box_preds_grad"] = grad[:, :, :nb].reshape(batch, hh, hw, nb)
            out["cls_preds_grad"] = grad[:, :, nb:nb + nc].reshape(batch, hh, hw, nc)
            if self.d.use_direction_classifier:
                out["dir_cls_preds_grad"] = grad[:, :, nb + nc:nb + nc + 2 * na].reshape(batch, hh, hw, 2 * na)
        return out

    # ---- training step (f3) ----
    def train_layout(self):
        """[(name, offset, size, is_state)] of the flat parameter / BatchNorm-state buffers (pp_train_layout)."""
        n, npar, nst = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.pp_train_layout(self._h, ctypes.byref(n), ctypes.byref(npar), ctypes.byref(nst)), "pp_train_layout")
        out = []
        for i in range(n.value):
            name, off, size, st = ctypes.c_char_p(), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0)
            self._check(self._lib.pp_train_layout_entry(self._h, i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(size),
                                                        ctypes.byref(st)), "pp_train_layout_entry")
            out.append((name.value.decode(), off.value, size.value, bool(st.value)))
        return out, npar.value, nst.value

    def train_graph_stats(self):
        """(captures, replays) of pp_train_step's hipGraphs: steady-state steps must replay."""
        c, r = ctypes.c_int32(0), ctypes.c_int32(0)
        self._check(self._lib.pp_train_graph_stats(self._h, ctypes.byref(c), ctypes.byref(r)), "pp_train_graph_stats")
        return c.value, r.value

    def pinned(self, shape, dtype):
        """A page-locked numpy array (PinnedArray) for targets / frames the loader fills in place."""
        return PinnedArray(self._lib, shape, dtype)

    def stream_ptr(self):
        """The handle's HIP stream as an integer (pp_stream): device work enqueued on it runs behind the handle's own."""
        p = ctypes.c_void_p(0)
        self._check(self._lib.pp_stream(self._h, ctypes.byref(p)), "pp_stream")
        return int(p.value or 0)

    @staticmethod
    def _loss_dict(losses):
        return {"loss": float(losses[0]), "loc_loss_reduced": float(losses[1]), "cls_loss_reduced": float(losses[2]),
                "dir_loss_reduced": float(losses[3]), "cls_pos_loss": float(losses[4]), "cls_neg_loss": float(losses[5]),
                "num_positives": int(losses[6])}

    def train_step_async(self, params_ptr, grads_ptr, state_ptr, labels, reg_targets):
        """Enqueue forward (training mode) + loss + backward on the resident frames (pp_train_step_async) and return.
        Until train_step_wait() the next batch may be uploaded (upload_async: the handle's other input buffer, on the
        copy stream beside the running kernels); labels / reg_targets are held here until then."""
        labels = _i32(np.asarray(labels))
        batch = labels.shape[0]
        reg_targets = _f32(np.asarray(reg_targets).reshape(batch, self.d.num_anchors, 7))   # (no copy when already so)
        if labels.shape != (batch, self.d.num_anchors):
            raise ValueError(f"labels must be [B, {self.d.num_anchors}]")
        lc = self.loss_config()
        self._check(self._lib.pp_train_step_async(self._h, ctypes.c_void_p(int(params_ptr)),
                                                  ctypes.c_void_p(int(grads_ptr)), ctypes.c_void_p(int(state_ptr)),
                                                  _ptr(labels), _ptr(reg_targets), batch, ctypes.byref(lc)),
                    "pp_train_step_async")
        self._train_targets = (labels, reg_targets)      # the copy engine reads them while the forward pass runs

    def train_step_wait(self):
        """Wait for the step train_step_async() launched; returns the reference's loss scalars."""
        losses = np.zeros(8, np.float32)
        self._check(self._lib.pp_train_step_wait(self._h, _ptr(losses)), "pp_train_step_wait")
        self._train_targets = None
        return self._loss_dict(losses)

    def train_step(self, params_ptr, grads_ptr, state_ptr, labels, reg_targets):
        """Forward (training mode) + loss + backward on the resident frames (pp_train_step).  The three pointers
        are integer device addresses of the flat float32 buffers; returns the reference's loss scalars."""
        self.train_step_async(params_ptr, grads_ptr, state_ptr, labels, reg_targets)
        return self.train_step_wait()

    def timer_start(self):
        self._check(self._lib.pp_timer_start(self._h), "pp_timer_start")

    def timer_stop(self):
        t = ctypes.c_float(0)
        self._check(self._lib.pp_timer_stop(self._h, ctypes.byref(t)), "pp_timer_stop")
        return float(t.value)

    def device_mem_free(self):
        v = ctypes.c_int64(0)
        self._check(self._lib.pp_device_mem_free(self._h, ctypes.byref(v)), "pp_device_mem_free")
        return v.value

    def device_copy_GBps(self, nbytes=1 << 30, reps=5):
        """Device-to-device copy rate (read + written GB/s) measured on the engine's stream."""
        g = ctypes.c_float(0)
        self._check(self._lib.pp_device_copy_bench(self._h, ctypes.c_int64(int(nbytes)), int(reps), ctypes.byref(g)),
                    "pp_device_copy_bench")
        return float(g.value)

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cu = ctypes.c_int32(0)
        mem = ctypes.c_int64(0)
        self._check(self._lib.pp_device_info(self._h, name, 256, ctypes.byref(cu), ctypes.byref(mem)), "pp_device_info")
        return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": mem.value}
