/* pp_hip.h -- C-ABI of the MI355X-native PointPillars inference path (libpp_hip.so).
 *
 * This is the drop-in boundary for the hot path of
 * krullgit/3D-Object-Detection-for-autonomous-navigation's `train.py evaluate`
 * (reference paths below are relative to that repository).  The reference has
 * no FFI for this path -- its callers are Python call sites
 * -- so each entry point names the Python interface it replaces; the binding a
 * maintainer adds on the reference side is the ctypes stub in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no framework types.  Every pointer is a
 *     HOST pointer unless its parameter is documented "device pointer".
 *   - integer status codes (PP_OK == 0); no exceptions cross the ABI; the text
 *     of the last failure is returned by pp_last_error().
 *   - one handle per GPU; a handle is not thread-safe; distinct handles are
 *     fully independent (own stream, own device workspaces sized at create
 *     for max_batch x max_points_per_frame).
 *   - every compute entry point runs on the GPU (gfx950).  There is no CPU
 *     fallback: without a device, pp_create fails.
 */
#ifndef PP_HIP_H
#define PP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: PP_ERR_NUMERIC, pp_set_gemm_precision / pp_get_gemm_precision, pp_set_cache_budget.  4: pp_train_fetch_decisions. */
#define PP_ABI_VERSION 4

enum pp_status {
    PP_OK = 0,
    PP_ERR_ARG = 1,    /* null / out-of-range argument */
    PP_ERR_STATE = 2,  /* call order: weights or anchors not set */
    PP_ERR_HIP = 3,    /* a HIP runtime call failed (see pp_last_error) */
    PP_ERR_SHAPE = 4,  /* tensor shape does not match the configuration */
    PP_ERR_UNSUPPORTED = 5,
    PP_ERR_NUMERIC = 6 /* non-finite head outputs: pp_get_detections / pp_detect / pp_predict never hand out NaN boxes */
};

/* GEMM arithmetic of the backbone (pp_set_gemm_precision) */
enum pp_gemm_precision {
    PP_PREC_SPLIT_F16 = 0, /* default: fp32 results on the 16-bit matrix pipe, every operand as two float16 pieces.
                            * Needs |BN-folded weight| < 32768 (checked per layer at pp_finalize_weights: a layer that
                            * fails runs in PP_PREC_F32 by itself) and |activation| < 65504 (checked on the device:
                            * a frame whose activations leave the range ends in PP_ERR_NUMERIC) */
    PP_PREC_F32 = 1        /* every layer on the float32 matrix instruction (v_mfma_f32_32x32x2_f32): float32's range */
};

typedef struct pp_engine* pp_handle;

/* Mirrors the keys the reference's hot path reads from configs/train.yaml
 * (SURVEY.md section 5 "Config / flag system"; values of the shipped config in
 * SURVEY Appendix B). */
typedef struct pp_config {
    double pc_range[6];   /* model.second.voxel_generator.point_cloud_range: xmin ymin zmin xmax ymax zmax */
    double voxel_size[3]; /* ...voxel_size (float64, as load_data.py:2573-2574 builds them) */
    int32_t max_points;   /* ...max_number_of_points_per_voxel  (T) */
    int32_t max_voxels;   /* ...max_number_of_voxels */
    int32_t num_point_features; /* model.second.num_point_features (F: 3 or 4) */
    int32_t pfn_filters;  /* voxel_feature_extractor.num_filters (C) */
    int32_t layer_nums[3];            /* rpn.layer_nums */
    int32_t layer_strides[3];         /* rpn.layer_strides */
    int32_t num_filters[3];           /* rpn.num_filters */
    int32_t upsample_strides[3];      /* rpn.upsample_strides */
    int32_t num_upsample_filters[3];  /* rpn.num_upsample_filters */
    int32_t num_anchor_per_loc;       /* len(rotations) * len(sizes) */
    int32_t num_class;                /* model.second.num_class.  1 in the shipped config; > 1 is an extension the
                                         reference leaves as a TF stub (model/voxelnet.py:1183-1185): score = max,
                                         label = argmax over the class logits of an anchor; the fused head row must
                                         hold num_anchor_per_loc * (7 + num_class + 2) <= 32 columns */
    int32_t nms_pre_max_size;         /* model.second.nms_pre_max_size */
    int32_t nms_post_max_size;        /* model.second.nms_post_max_size */
    float nms_score_threshold;        /* model.second.nms_score_threshold */
    float nms_iou_threshold;          /* model.second.nms_iou_threshold */
    float anchor_area_threshold;      /* eval_input_reader.anchor_area_threshold */
    int32_t max_batch;                /* frames per call the workspaces are sized for */
    int32_t max_points_per_frame;     /* points per frame the workspaces are sized for */
    int32_t use_direction_classifier; /* model.second.use_direction_classifier (model/voxelnet.py:690,714,1093,1297):
                                         0 = no conv_dir_cls head, no direction flip */
    int32_t with_distance;            /* voxel_feature_extractor.with_distance (model/pointpillars.py:185-188): one more
                                         PFN input feature, the point's Euclidean norm */
} pp_config;

/* One detection, in NMS keep order (descending score), as VoxelNet.predict
 * assembles it (model/voxelnet.py:1281-1369). */
typedef struct pp_detection {
    double box3d_camera[7]; /* x y z l h w r, float64 as box_lidar_to_camera returns (eval_helper_functions.py:735-740) */
    float box3d_lidar[7];   /* x y z w l h r after the direction flip (model/voxelnet.py:1305-1310) */
    float score;            /* sigmoid(cls) (model/voxelnet.py:1150) */
    int32_t label;          /* label_preds: argmax over the class logits (0 when num_class == 1) */
    int32_t dir_label;      /* argmax of the direction head */
    int32_t anchor_index;   /* flat anchor index (y, x, rot) of the source anchor */
    int32_t reserved;
} pp_detection;

/* ---- lifetime -------------------------------------------------------- */

/* Replaces VoxelNet.__init__ (model/voxelnet.py:727-787) + the dataloader's
 * per-config constants.  Fails with PP_ERR_HIP when no gfx950 device `device`
 * is usable. */
int pp_create(const pp_config* cfg, int device, pp_handle* out);
int pp_destroy(pp_handle h);
/* Text of the last failure on `h` (or of the last pp_create failure when h is NULL). */
const char* pp_last_error(pp_handle h);
int pp_abi_version(void);

/* ---- weights: replaces net.load_weights (train.py:731-734) ------------ */

/* One named float32 tensor in the Keras layout (names / layouts listed in
 * INTEGRATION.md and <package>/weights.py).  The data is copied. */
int pp_set_weight(pp_handle h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* Verifies that every tensor is present, folds BatchNorm (eps 1e-3) into the
 * following GEMM and uploads the kernel-side layouts. */
int pp_finalize_weights(pp_handle h);

/* Static anchors [A,7] (x y z w l h r) in the reference's (y, x, rot) order and
 * their clamped integral-image cells [A,4] (x0 y0 x1 y1), built once on the
 * host (replaces the per-frame generate_anchors / rbbox2d_to_near_bbox calls,
 * load_data.py:3029-3043). */
int pp_set_anchors(pp_handle h, const float* anchors, const int32_t* cells, int64_t num_anchors);

/* ---- stage entry points (parity checkpoints) -------------------------- */
/* These reuse the device buffers of the fused path: after any of them the handle holds no resident frames
 * and no fused-path results (pp_detect_async / pp_get_detections / pp_fetch_intermediates then return
 * PP_ERR_STATE until the next upload + detect). */

/* points_to_voxel(points, voxel_size, coors_range, max_points, True, max_voxels)
 * (load_data.py:695-771) for ONE frame of n points [n,F].  Outputs are sized
 * by the caller for max_voxels pillars: voxels [max_voxels,T,F] (only the
 * first *n_pillars rows are written, zero padded), coors [max_voxels,3] (z y x),
 * num_points [max_voxels]. */
int pp_points_to_voxel(pp_handle h, const float* points, int64_t n, float* voxels, int32_t* coors,
                       int32_t* num_points, int32_t* n_pillars);

/* The dataloader's anchors_mask for `batch` frames (load_data.py:3043-3072)
 * from batched pillar coordinates coors [P,4] (b z y x).  mask [batch,A] u8. */
int pp_anchor_mask(pp_handle h, const int32_t* coors, int64_t num_pillars, int32_t batch, uint8_t* mask);

/* VoxelNet.__call__(voxels, num_points, coors, batch_anchors) eval branch
 * (model/voxelnet.py:850-916): voxels [P,T,F], num_points [P], coors [P,4]
 * (b z y x, unique per frame as points_to_voxel produces them).  Outputs NHWC:
 * box_preds [batch,H',W',2*7], cls_preds [batch,H',W',2], dir_cls_preds
 * [batch,H',W',4].  Optional checkpoints (may be NULL): pillar_features [P,C]
 * (PillarFeatureNet output) and canvas [batch,ny,nx,C] (PointPillarsScatter
 * output, NHWC). */
int pp_forward_voxels(pp_handle h, const float* voxels, const int32_t* num_points, const int32_t* coors,
                      int64_t num_pillars, int32_t batch, float* box_preds, float* cls_preds,
                      float* dir_cls_preds, float* pillar_features, float* canvas);

/* VoxelNet.predict(example, preds_dict) (model/voxelnet.py:1060-1390) on the
 * head maps.  anchors_mask [batch,A] u8; rect, trv2c [batch,16] row-major 4x4.
 * dets [batch * nms_post_max_size]; n_dets [batch] (0 == the reference's
 * all-None dict). */
int pp_predict(pp_handle h, const float* box_preds, const float* cls_preds, const float* dir_cls_preds,
               const uint8_t* anchors_mask, const float* rect, const float* trv2c, int32_t batch,
               pp_detection* dets, int32_t* n_dets);

/* ---- fused path: raw points -> detections ----------------------------- */

/* Copies `batch` frames of raw points into the engine's device input buffer.
 * points: concatenated [sum n_b, F]; frame_offsets [batch+1] (row offsets). */
int pp_upload_points(pp_handle h, const float* points, const int32_t* frame_offsets, int32_t batch);
/* Same without waiting: `points_pinned` must be page-locked host memory (pp_host_alloc, or the caller's own
 * hipHostMalloc / hipHostRegister) and must stay unchanged until the copy has run (pp_sync after the
 * pp_detect_async that consumes these frames).  frame_offsets is copied before the call returns.
 * The handle's input is double-buffered and this copy runs on the handle's own copy stream into the buffer
 * the pass in flight is not reading, so the upload of batch k+1 proceeds beside the kernels of batch k:
 *     upload_async(k+1); sync + get_detections (batch k); detect_async (batch k+1); ...
 * pp_detect_async orders itself behind the copy.  This is the double-buffered feed of raw points that replaces
 * the per-frame host-to-device hand-over of train.py:748.
 * Batches of up to 4 frames (the latency case) are not copied at all: the call only fills a page-locked descriptor
 * and the next pass's first kernel reads offsets and points straight from `points_pinned` over the host link while
 * it writes the device copies the later kernels use -- no copy-engine transfer, no event chain (batch 1, upload
 * included: 0.29 -> 0.25 ms).  `points_pinned` must then be device-mapped page-locked memory (pp_host_alloc and
 * hipHostMalloc are; memory that is not falls back to the copy); PP_NO_ZERO_COPY=1 always copies. */
int pp_upload_points_async(pp_handle h, const float* points_pinned, const int32_t* frame_offsets, int32_t batch);
/* Page-locked host memory for the staging buffers above (stateless; any thread). */
int pp_host_alloc(int64_t bytes, void** out);
int pp_host_free(void* p);
/* Same, from a DEVICE pointer `points_dev` (device-to-device copy on the engine's stream, no wait).
 * Ordering contract: `producer_stream` is the hipStream_t on which the work that writes `points_dev` was
 * queued (the engine's stream then waits for an event recorded there), or NULL when that work has already
 * completed (the caller synchronised).  `points_dev` must stay valid until the engine's stream has passed
 * the copy. */
int pp_upload_points_device(pp_handle h, const void* points_dev, const int32_t* frame_offsets, int32_t batch,
                            void* producer_stream);
/* Frames currently resident for pp_detect_async (`uploaded`) and frames of the last enqueued
 * pp_detect_async whose results pp_get_detections returns (`results`); 0 after a stage entry point reused
 * the buffers.  Either pointer may be NULL. */
int pp_current_batch(pp_handle h, int32_t* uploaded, int32_t* results);
/* Calibration for the uploaded frames: rect, trv2c [batch,16]. */
int pp_set_calib(pp_handle h, const float* rect, const float* trv2c, int32_t batch);

/* Enqueues the whole path (a1..a12 of SURVEY section 8a) for the frames resident in
 * the engine's input buffer on the engine's stream: voxelise -> PFN + scatter
 * -> anchor mask -> backbone + heads -> top-k / decode / NMS -> detections in
 * device memory, then an async copy into the engine's pinned result buffer.
 * Returns without waiting. */
int pp_detect_async(pp_handle h);
/* Waits for the engine's stream. */
int pp_sync(pp_handle h);
/* Copies the results of the last pp_detect_async (waits for the engine's stream first; immediate after
 * pp_sync).  dets [batch*nms_post_max_size], n_dets [batch].  PP_ERR_STATE when there are none.
 * PP_ERR_NUMERIC when a frame's head maps hold a non-finite value (pp_last_error names the frame): with
 * PP_PREC_SPLIT_F16 that is an activation beyond the float16 pieces' range -- the frames are still resident, so
 * pp_set_gemm_precision(h, PP_PREC_F32) + pp_detect_async + pp_get_detections re-runs them in float32; in PP_PREC_F32
 * the network itself overflowed (the reference, model/voxelnet.py:1060-1390, would return NaN boxes).  Nothing is
 * written to dets / n_dets in that case. */
int pp_get_detections(pp_handle h, pp_detection* dets, int32_t* n_dets);
/* Selects the backbone's GEMM arithmetic (enum pp_gemm_precision) for this handle; re-derives the device weights
 * (pp_finalize_weights' work) when they are loaded.  Waits for the handle's stream.  The default comes from the
 * environment (PP_GEMM_PREC=f32) or is PP_PREC_SPLIT_F16. */
int pp_set_gemm_precision(pp_handle h, int32_t precision);
int pp_get_gemm_precision(pp_handle h, int32_t* precision);
/* Last-level-cache budget of a pass, in MiB (default 256, 0 = off).  Layers whose input + output maps exceed it are run
 * over sub-ranges of the batch's frames, a block's consecutive separable layers sub-range by sub-range, so that a layer
 * reads what the layer before has just written while it still sits in the 256 MB cache (KITTI-shaped B = 32: -5 % per
 * pass; the shipped 80 x 64 grid fits as it is).  Results do not change.  Right for ONE handle in flight per GPU; a
 * caller that keeps several handles in flight (bench.py's feeder) sets 0: their working sets evict each other.
 * No reference counterpart (TensorFlow's executor owns that schedule).  Waits for the handle's stream. */
int pp_set_cache_budget(pp_handle h, int32_t megabytes);
/* Convenience: upload + calib + detect + sync + get (the evaluate loop body,
 * train.py:689-786 minus annotation formatting). */
int pp_detect(pp_handle h, const float* points, const int32_t* frame_offsets, int32_t batch,
              const float* rect, const float* trv2c, pp_detection* dets, int32_t* n_dets);

/* Debug / parity taps of the fused path (after pp_sync), any pointer may be
 * NULL: per-frame pillar counts [batch]; coors [batch*max_voxels,3];
 * num_points [batch*max_voxels]; anchors mask [batch,A]; head maps as in
 * pp_forward_voxels; canvas [batch,ny,nx,C]. */
int pp_fetch_intermediates(pp_handle h, int32_t* n_pillars, int32_t* coors, int32_t* num_points,
                           uint8_t* anchors_mask, float* box_preds, float* cls_preds,
                           float* dir_cls_preds, float* canvas);

/* ---- measurement ------------------------------------------------------ */

/* level 0: no events.  level 1: HIP events on the engine's stream around
 * every kernel launch of pp_detect_async (for bench.py's roofline leg). */
int pp_set_profiling(pp_handle h, int32_t level);
/* After pp_sync with profiling on: number of timed kernel launches of the last
 * pp_detect_async, their names and durations (ms).  Buffers sized by the
 * caller for `capacity` entries; names are static strings. */
int pp_get_kernel_times(pp_handle h, int32_t capacity, const char** names, float* ms, int32_t* count);
/* HIP-event stopwatch on the engine's stream. */
int pp_timer_start(pp_handle h);
int pp_timer_stop(pp_handle h, float* elapsed_ms); /* records, waits, returns the elapsed time */

/* Kernel-tuning aid: runs RPN layer `layer` (0-based, in launch order; the last one is
 * the heads) `reps` times on whatever the activation buffers hold for `batch` frames and
 * returns the average launch duration.  `ablate` is a debug bit mask (0 = the real
 * kernel; bits switch off phases of the GEMM kernel, results are then wrong). */
int pp_bench_layer(pp_handle h, int32_t layer, int32_t batch, int32_t reps, int32_t ablate, float* avg_ms);
/* Number of RPN layer launches per forward pass and the tag ("<kernel>:<layer>") of one. */
int pp_layer_count(pp_handle h, int32_t* count);
const char* pp_layer_tag(pp_handle h, int32_t layer);

/* ---- AP-evaluator overlaps ------------------ */

/* Replaces rotate_iou_gpu_eval (second/core/non_max_suppression/nms_gpu.py:618-653; kernel
 * :579-615, device functions :180-415, :564-576).  boxes [n,5], query_boxes [k,5] float32 rows
 * (centre x, centre y, x size, y size, angle -- clockwise positive); out [n,k] float32 row-major:
 * rotated-rectangle intersection of (query k, box n) divided by: -1 the union, 0 the query's area,
 * 1 the box's area, 2 nothing (raw area).  Stateless; host pointers; `device` is the HIP device. */
int pp_rotate_iou_eval(int device, const float* boxes, int64_t n, const float* query_boxes, int64_t k,
                       int32_t criterion, float* out);
/* Replaces d3_box_overlap (second/utils/eval.py:159-163 with its kernel :132-156): camera-frame
 * boxes [n,7] / [k,7] float64 rows (x, y, z, l, h, w, ry); out [n,k] float64: BEV intersection
 * (float32, as above with criterion 2) x height overlap / {union | box volume | query volume | 1}. */
int pp_d3_box_overlap(int device, const double* boxes, int64_t n, const double* query_boxes, int64_t k,
                      int32_t criterion, double* out);

/* ---- training loss at the head maps ---------- */

/* Mirrors model.second.loss / pos_class_weight / ... of configs/train.yaml:147-167. */
typedef struct pp_loss_config {
    float alpha;             /* weighted_sigmoid_focal.alpha (0.25); negative = no alpha weighting */
    float gamma;             /* weighted_sigmoid_focal.gamma (2.0) */
    float sigma;             /* weighted_smooth_l1.sigma (3.0) */
    float code_weight[7];    /* weighted_smooth_l1.code_weight */
    float pos_class_weight, neg_class_weight;
    float classification_weight, localization_weight, direction_loss_weight;
    int32_t norm_by_num_positives;     /* loss_norm_type == "NormByNumPositives" */
    int32_t encode_rad_error_by_sin;   /* model.second.encode_rad_error_by_sin */
    int32_t use_direction_classifier;
} pp_loss_config;

/* Replaces the loss half of VoxelNet.call in training mode (model/voxelnet.py:922-1049: prepare_loss_weights
 * :461-512, create_loss :74-155, sigmoid_focal_classification_loss :262-364, WeightedSmoothL1LocalizationLoss
 * :407-459, get_direction_target :38-46, weighted_softmax_classification_loss :180-235) on the head maps the
 * last forward pass of this handle left on the device (pp_detect / pp_detect_async + pp_sync /
 * pp_forward_voxels with the same `batch`).  labels [batch][A] int32 (>0 class, 0 background, -1 ignored),
 * reg_targets [batch][A][7] float32: the dataloader's `labels` / `reg_targets` (load_data.py:3096-3100).
 * losses[8] = {loss, loc_loss_reduced, cls_loss_reduced, dir_loss_reduced, cls_pos_loss, cls_neg_loss,
 * number of positive anchors, 0}.  head_grad (may be NULL): d loss / d head map, [batch][H'*W'][32] float32 in
 * the fused head-map layout [box napl*7 | cls napl | dir napl*2 | zero pad] -- what the backward pass of the
 * head GEMM consumes.  Host pointers; synchronous. */
int pp_head_loss(pp_handle h, const int32_t* labels, const float* reg_targets, int32_t batch,
                 const pp_loss_config* cfg, float* losses, float* head_grad);

/* Replaces optimizer.apply_gradients for one flat float32 parameter buffer (train.py:228-239, :301:
 * tfa.optimizers.AdamW over tf.keras Adam): var -= weight_decay * var; m, v moments; var -= lr_t * m /
 * (sqrt(v) + epsilon) with lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) computed by the caller (t = step + 1,
 * lr from the ExponentialDecay schedule).  DEVICE pointers (params, grads, m, v: n floats each, on `device`);
 * `stream` is a hipStream_t or NULL; asynchronous on that stream.  Stateless. */
int pp_adamw_step_device(int device, void* stream, float* params, const float* grads, float* m, float* v,
                         int64_t n, float lr_t, float beta1, float beta2, float epsilon, float weight_decay);

/* ---- training step ---------------------------- */

/* The trainable tensors live in ONE flat float32 device buffer owned by the caller (a second one of the same
 * order receives the gradients, a third, smaller one holds the BatchNorm moving statistics): what the AdamW kernel
 * above and the data-parallel all-reduce work on.  These two calls describe the order: entry i is the Keras tensor
 * `name` (layouts of <package>/weights.py: depthwise [3,3,Cin,1], pointwise [1,1,Cin,Cout], Conv2DTranspose
 * [k,k,Cout,Cin], Dense [Fa,C], heads [1,1,Cin,Cout] + bias), `size` floats at `offset` of the parameter buffer
 * (is_state 0) or of the state buffer (is_state 1: moving_mean / moving_variance).  `name` stays valid for the
 * life of the handle. */
int pp_train_layout(pp_handle h, int32_t* n_entries, int64_t* n_param_floats, int64_t* n_state_floats);
int pp_train_layout_entry(pp_handle h, int32_t i, const char** name, int64_t* offset, int64_t* size, int32_t* is_state);

/* Replaces one trainStep of train.py:265-304 up to (not including) optimizer.apply_gradients: VoxelNet.call in
 * training mode (model/voxelnet.py:850-1049: PillarFeatureNet, scatter, RPN with batch-statistics BatchNorm, the
 * losses) on the `batch` frames resident in the handle (pp_upload_points*; they are voxelised here), and the
 * gradient of `loss` with respect to every trainable tensor.  params_dev / grads_dev / state_dev: DEVICE pointers to
 * the flat buffers described by pp_train_layout (grads overwritten; the moving statistics in state_dev updated in
 * place as Keras does).  labels [batch][A] int32, reg_targets [batch][A][7] float32: HOST pointers, the dataloader's
 * targets (load_data.py:3096-3100).  losses[8] as pp_head_loss.  Runs on the handle's stream and returns when the
 * step has finished (the caller then all-reduces grads_dev and calls pp_adamw_step_device). */
int pp_train_step(pp_handle h, const float* params_dev, float* grads_dev, float* state_dev, const int32_t* labels,
                  const float* reg_targets, int32_t batch, const pp_loss_config* cfg, float* losses);

/* The same step in two halves, for the loader's hand-over (the reference's tf.data pipeline prepares batch n + 1 while
 * trainStep n runs, train.py:228-304): _async enqueues the step on the handle's stream and returns; between the two
 * calls the caller may upload the NEXT batch (pp_upload_points_async: it goes into the handle's other input buffer on
 * the copy stream, beside the running kernels); _wait returns when the step has finished, with its losses.  labels /
 * reg_targets must stay unchanged until _wait returns.  pp_train_step = _async + _wait. */
int pp_train_step_async(pp_handle h, const float* params_dev, float* grads_dev, float* state_dev, const int32_t* labels,
                        const float* reg_targets, int32_t batch, const pp_loss_config* cfg);
int pp_train_step_wait(pp_handle h, float* losses);

/* The handle's HIP stream (hipStream_t as void*).  A caller that enqueues its own device work behind a pp_train_step_async
 * -- the gradient all-reduce and pp_adamw_step_device of the optimizer step (train.py:301) -- does it on this stream and
 * needs no host synchronisation in between: pp_train_step_wait then waits for that work too. */
int pp_stream(pp_handle h, void** stream);

/* Debug / parity tap of the last finished training step: the decisions taken at its non-differentiable points.
 * layer >= 0: the layer-th BatchNorm + ReLU of the RPN in forward order (separable layers and transposed convolutions
 * as the network lists them: block1/0 .., deconv1, block2/0 .., deconv2, ...): relu_mask receives one byte per element
 * of the layer's pre-BatchNorm output ([b][y][x][c]; a transposed convolution: [b][y][x][tap][c] over its INPUT
 * pixels), 1 where the backward pass lets the gradient through.  layer < 0: the PFN's max -- relu_mask receives
 * int32 values (4 bytes each) [batch][max_voxels][C]: the winning row of the pillar, -1 = a zero-padded row, -2 = the
 * maximum is not positive (no gradient); slots >= the frame's pillar count are undefined.  *count = elements of that
 * layer; relu_mask NULL: only the count.  A test compares the step's gradients with a float64 graph that takes the
 * SAME decisions (ReLU and max are not differentiable where a value is within round-off of the kink). */
int pp_train_fetch_decisions(pp_handle h, int32_t layer, uint8_t* relu_mask, int64_t capacity, int64_t* count);

/* How often pp_train_step captured a hipGraph and how often it replayed one (one graph per input buffer of the
 * handle): steady-state steps must replay -- a regression check, not part of the reference's surface. */
int pp_train_graph_stats(pp_handle h, int32_t* captures, int32_t* replays);

/* Measurement helper: `reps` device-to-device copies of `bytes` on the handle's stream, timed with HIP events;
 * *gbytes_per_s = read + written bytes per second (what an HBM-bound kernel can reach on this part, next to the
 * spec constant the roofline fractions use). */
int pp_device_copy_bench(pp_handle h, int64_t bytes, int32_t reps, float* gbytes_per_s);

/* Device properties for reports: name (<=255 chars), CU count, bytes of HBM. */
int pp_device_info(pp_handle h, char* name, int32_t name_capacity, int32_t* compute_units, int64_t* hbm_bytes);
/* Bytes of HBM currently free on the handle's device (hipMemGetInfo): leak checks, sizing. */
int pp_device_mem_free(pp_handle h, int64_t* free_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PP_HIP_H */
