// Internal declarations shared by the HIP translation units of libpp_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts are assumed throughout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include <hip/hip_ext.h>

#include "../../include/pp_hip.h"

#define PP_WAVE 64
// split-precision operand pieces of the GEMM kernels (backbone.hip explains the modes); pp_api.hip splits the
// weights accordingly.  A build-time choice: -DPP_SPLIT_MODE=n
#ifndef PP_SPLIT_MODE
#define PP_SPLIT_MODE 1
#endif
// pieces per value in the pre-split weight layouts ([cin/16][PP_NPIECE][n][16] 16-bit words)
#define PP_NPIECE ((PP_SPLIT_MODE == 0) ? 3 : 2)
// zeroed floats in front of every activation buffer (>= the widest layer input, 384 channels):
// the GEMM producers read convolution zero-padding from there instead of masking loaded values
#define PP_ZPAD_FLOATS 512
// fused head map: one 128-byte row per head-map pixel: [box napl*7 | cls napl*ncls | dir napl*2 | 0 pad]
#define PP_HEAD_COLS 32

// ----- kernel launches ----------------------------------------------------
// Every hot-path kernel is launched through PP_LAUNCH.  Normally that is a plain hipLaunchKernelGGL.  While a
// handle collects per-kernel times (pp_set_profiling; plain launches, never inside a graph capture) the launch
// carries a start / stop event pair of its own (hipExtLaunchKernelGGL): the pair brackets the kernel's execution
// on the device -- the same interval rocprofv3's kernel trace reports -- not the gaps between launches.
struct PpProf {
    pp_engine* e;       // handle collecting times on this thread (NULL: none)
    const char* tag;    // name to record ("<kernel symbol>:<layer>"), or NULL: the launch site's own name
};
extern thread_local PpProf g_pp_prof;
bool pp_prof_events(const char* name, hipEvent_t* start, hipEvent_t* stop);   // pp_api.hip
#define PP_LAUNCH(NAME, KERNEL, GRID, BLOCK, SHMEM, STREAM, ...)                                            \
    do {                                                                                                    \
        hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                          \
        if (g_pp_prof.e != nullptr && pp_prof_events(NAME, &pe0_, &pe1_))                                   \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, pe0_, pe1_, 0, __VA_ARGS__);          \
        else                                                                                                \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, __VA_ARGS__);                            \
    } while (0)

// inclusive prefix sum over the 64 lanes of a wave with DPP adds (row shifts inside the rows of 16, then the two row
// broadcasts): six short VALU operations instead of six ds_bpermute round trips (__shfl_up) -- the scans of the
// single-workgroup stages are dependent chains: their latency is what they cost
// Requirements: all 64 lanes of the wave active at the call (the row shifts read inactive lanes as 0 only through
// bound_ctrl; every caller scans whole waves), and a wave64 GFX9 target: row_bcast:15 / :31 do not exist elsewhere.
// This library is written for gfx950 only -- any other --offload-arch stops here instead of mis-scanning.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "pp_common.h: the DPP wave scans (row_bcast) and the MFMA kernels of this library target gfx950 (MI355X) only"
#endif
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}


// ----- numeric guard of the split-precision GEMM path ----------------------------------
// v_max_f32 (fmaxf) returns the OTHER operand when one is a NaN: a ReLU written with it would turn the NaN an
// out-of-range activation produces (two float16 pieces: |x| >= 65520 -> inf - inf) into a clean 0 and the frame into
// silently wrong boxes.  The backbone's ReLUs keep a NaN, so it reaches the head maps, where the post-process sees it
// (bit PP_NDETS_NONFINITE of a frame's detection count -> PP_ERR_NUMERIC).
__device__ __forceinline__ float relu_keep_nan(float x) { return (x < 0.f) ? 0.f : x; }
__device__ __forceinline__ bool pp_finite(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }
#define PP_NDETS_NONFINITE (1 << 30)

// ----- voxel grid geometry (float64, as the reference's index math) -----
struct VoxGeom {
    double lo[3];   // range minimum x y z
    double vs[3];   // voxel size x y z
    int grid[3];    // nx ny nz = round((max-min)/vs)
    int ncell;      // nx*ny*nz
};

// ----- one GEMM-shaped layer of the RPN ---------------------------------
enum LayerKind { LAYER_SEP = 0, LAYER_DECONV = 1, LAYER_HEAD = 2 };

struct LayerDesc {
    int kind;
    int cin, cout;        // channels in / out (cout of ONE tap for deconv)
    int stride;           // depthwise stride (sep)
    int k;                // deconv kernel == stride
    int in_h, in_w;       // input map
    int out_h, out_w;     // output map (sep: strided; deconv: in*k; head: in)
    int n_total;          // GEMM N: cout (sep), k*k*cout (deconv), 32 (head, zero padded)
    float* d_dw;          // [9][cin] depthwise taps (sep)
    float* d_wt;          // [n_total][cin] BN-folded, transposed
    float* d_wt16;        // the same weights as three bf16 pieces, [cin/16][3][n_total][16] 16-bit words (or NULL)
    float* d_bias;        // [cout] (sep/deconv: folded BN shift) or [32] (head)
    const float* in;      // input activation  [B, in_h, in_w, cin]
    float* out;           // output base
    int ld_out;           // channels per output pixel row (row stride, floats)
    int co_off;           // channel offset inside the output row (concat placement)
    // heads fused into a deconv's epilogue (head_mode 1: write partial + bias, 2: add partial)
    int head_mode;
    float* d_head_wt;     // [PP_HEAD_COLS][cout] slice of the head kernels, transposed
    float* d_head_bias;   // [PP_HEAD_COLS]
    float* d_head_wt16;   // d_head_wt as three bf16 pieces, [cout/16][3][PP_HEAD_COLS][16] 16-bit words (or NULL)
    // sparse canvas (first layer only): cell -> pillar map [batch][occ_nz][in_h][in_w]; a window position whose
    // cell holds no pillar is read from the zero header instead of the (unwritten) canvas.  NULL: dense input
    const int* d_occ;
    int occ_nz;
    // the same occupancy as a bitmap (PfnParams::occbits), when this pass's PFN launch wrote it: three 16-byte-free
    // lookups per window instead of one per window element and z-cell.  NULL: the cell map is consulted
    const unsigned long long* d_occbits;
    // compact class-logit plane [B * H' * W'][cls_ncol], written by the LAST fused-head deconv (NULL elsewhere)
    float* d_cls_plane;
    int cls_col0, cls_ncol;
    const char* name;
};

// ----- launchers (each in its own .hip file) ----------------------------
// small batches (the latency case) are fed without a copy engine: the first kernel reads the frames' offsets and
// points straight from the caller's page-locked buffer (device-mapped) through this page-locked descriptor and
// leaves the device copies the later kernels use (pts_dst / offsets_dst); feed == NULL: inputs are on the device
struct PpFeed {
    const float* src;        // device address of the caller's page-locked points [sum N, F]
    int pad_[2];
    int offsets[1];          // [batch + 1] follow
};
// (occbits != NULL: the frames' occupancy bitmaps, occ_n 64-bit words each, are cleared by the same launch)
void launch_cell_first(const float* pts, const int* offsets, int batch, int max_n, int F, const VoxGeom& g,
                       int* cell, int* first, int* cellmap, const PpFeed* feed, float* pts_dst, int* offsets_dst,
                       hipStream_t s, unsigned long long* occbits = nullptr, int occ_n = 0);
// returns (through *sorted_in_b) nothing; the host derives the final buffer from voxel_sort_passes()
int voxel_sort_passes(int max_voxels);
bool voxel_first_in_lds(int max_n, int ncell, int max_voxels);   // pass first = NULL to both launchers below
void launch_voxel_frame(const int* offsets, const int* cell, const int* first, int* cellmap, unsigned* keyA,
                        unsigned* idxA, unsigned* keyB, unsigned* idxB, int* pillar_start, int* pillar_cell,
                        int* npillars, int* nvalid, int batch, int max_n, int ncell, int max_voxels, hipStream_t s);
// pts_sorted[n0 + j] = pts[n0 + sorted_idx[n0 + j]], j < nvalid[frame]: the pillar-sorted copy the PFN streams
void launch_sort_points(const float* pts, const int* offsets, const unsigned* sorted_idx, const int* nvalid, int batch,
                        int max_n, int F, float* pts_sorted, hipStream_t s);
// pts_sorted: the pillar-sorted copy k_voxel_frame leaves ([sum N][F], frame b's valid points at offsets[b]..)
void launch_voxel_expand(const float* pts_sorted, const int* offsets, const unsigned* sorted_idx, const int* pillar_start,
                         const int* pillar_cell, const int* npillars, int frame, int F, int T, int max_voxels,
                         int ny, int nx, float* voxels, int* coors, int* num_points, hipStream_t s);
void launch_build_cellmap(const int* coors4, int64_t P, int ncell, int ny, int nx, int* cellmap, hipStream_t s);

struct PfnParams {
    // geometry
    int batch, nz, ny, nx, C, F, T, max_voxels;
    float vx, vy, x_off, y_off;
    // weights: w [Fa][C] folded, bias [C] folded
    const float* w;
    const float* bias;
    // cell -> pillar map [batch][nz][ny][nx]
    const int* cellmap;
    // CSR source: pillar-sorted points (frame b: rows offsets[b] + pillar_start[b][p] .. of pts_sorted)
    const float* pts_sorted;
    const int* offsets;
    const int* pillar_start;
    // pillar-centric launch (sparse canvas): linear (z, y, x) cell of every pillar [batch][max_voxels], pillars per frame [batch]
    const int* pillar_cell;
    const int* npillars;
    // ... which also leaves the frame's occupancy bitmap (occ_words(nx) 64-bit words per grid row, bit x + 1 of a row
    // = "a pillar in column x, any z"; cleared by k_cell_first): what the sparse first layer and the one-launch
    // anchor mask read instead of the cell map.  NULL: not written
    unsigned long long* occbits;
    // padded source (compat)
    const float* voxels;
    const int* num_points;
    // outputs
    float* canvas;    // [batch][ny][nx][C]
    int sparse;       // 1: only cells that hold a pillar are written (the first layer consults the cell map)
    int with_distance;  // 1: one more input feature, the point's Euclidean norm (w has F + 6 rows)
    float* feat_out;  // optional [P][C]
    // few frames (the latency case): one extra workgroup per frame of the PFN launch computes the frame's anchor mask
    // (it depends on the voxeliser's cell map only, and nothing before the post-process reads it): NULL = not fused
    const int* am_cells;   // [A][4] static anchor cells
    int64_t am_A;
    float am_threshold;
    uint8_t* am_mask;      // [batch][A]
};
bool pfn_can_carry_anchor_mask(const PfnParams& p, bool padded_source);
bool pfn_writes_occbits(const PfnParams& p, bool padded_source);           // would launch_pfn's kernel set PfnParams::occbits?   // would launch_pfn run the extra workgroups?
int launch_pfn(const PfnParams& p, bool padded_source, hipStream_t s);  // returns 0 or PP_ERR_UNSUPPORTED

void launch_anchor_mask(const int* cellmap, int batch, int nz, int ny, int nx, const int* cells, int64_t A,
                        float threshold, int* integ, uint8_t* mask, hipStream_t s);
// 64-bit words per grid row of the occupancy bitmap: bits 0 .. nx + 1 (bit x + 1 = column x; bit 0 = the padding
// column x = -1) plus one spare word, so that a reader may always fetch the word after the one its window starts in
__host__ __device__ static inline int occ_words(int nx) { return (nx + 2 + 63) / 64 + 1; }
// the anchor mask from the occupancy bitmap (nz == 1 grids: a bit is a count): one launch, no integral image
void launch_anchor_mask_bits(const unsigned long long* occbits, int batch, int ny, int nx, const int* cells, int64_t A,
                             float threshold, uint8_t* mask, hipStream_t s);

bool deconv_can_fuse_heads(const LayerDesc& L);
bool layer_writes_cls_plane(const LayerDesc& L);   // does this layer's kernel leave the compact class-logit plane?
bool sparse_input_supported(const LayerDesc& L, int batch);   // may L's input be a canvas with unwritten empty cells?
std::string layer_kernel_name(const LayerDesc& L, int batch);  // template instantiation that runs L at this batch
// training-mode forward of one separable layer (backbone.hip: k_sep_u<..., TR = 1>; called by train.hip)
struct SepTrainArgs {
    const float* in;             // input map [batch][in_h][in_w][cin] with a PP_ZPAD_FLOATS header in front: NaN-filled when
                                 // `coef` is given (relu(NaN * sc + sh) = 0 is the padding), zero-filled otherwise
    const float4* coef;          // [cin] (sc, sh, ., .) when `in` is a pre-BatchNorm map, NULL when it is an activation
    const float* dw;             // depthwise kernel [3][3][cin]
    const unsigned short* wt16;  // this step's pointwise kernel as two float16 pieces, [cin / 16][2][cout][16]
    float* Z;                    // out: pre-BatchNorm map [rows][cout]
    float* D;                    // out: depthwise output [rows][cin]
    float* stat;                 // out: statistics partials [rows written][2][cout] (one row per workgroup; at most
                                 // ceil(rows / 128) + 7 of them)
    int batch, in_h, in_w, cin, out_h, out_w, cout, stride;
    const char* tag;             // profiler name of the launch ("k_sep_u_tr:<layer>")
};
int launch_sep_train(const SepTrainArgs& t, hipStream_t s);   // rows of `stat` written, 0: shape not supported
// ... and of a plain product out[rows][N] = in[rows][K] . W (+ bias): the transposed convolutions' forward GEMMs, the heads
struct RowsTrainArgs {
    const float* in;             // [rows][K], rows contiguous (no header needed)
    const unsigned short* wt16;  // W as two float16 pieces, [K / 16][2][N][16]
    const float* bias;           // [N] or NULL
    float* out;                  // [rows][ld_out]
    float* stat;                 // statistics partials [rows written][2][N], or NULL
    long long rows;
    int K, N, ld_out;
    const char* tag;
};
int launch_rows_train(const RowsTrainArgs& t, hipStream_t s);
int launch_layer(const LayerDesc& L, int batch, float* d_head, hipStream_t s,
                 int ablate = 0, int frame0 = 0);
// may L be launched over the frames [frame0, frame0 + batch) of a total_batch-frame batch (k_sep_u's tile sub-ranges)?
bool launch_layer_subrange_ok(const LayerDesc& L, int frame0, int batch, int total_batch);  // returns 0 or PP_ERR_UNSUPPORTED

struct PostParams {
    int batch;
    int64_t A;
    int pre_max, post_max;
    float score_thr, iou_thr;
    const float* head;     // [batch][H'*W'][PP_HEAD_COLS] fused head map
    const float* cls;      // compact class logits [batch][H'*W'][napl*ncls] (same values as the head map's cls
                           // columns), or NULL: the candidate scan reads the head rows
    int napl;              // anchors per location
    int ncls;              // class logits per anchor (score = max, label = argmax)
    int use_dir;           // 0: no direction head, no flip
    const uint8_t* mask;   // [batch][A]
    const float* anchors;  // [A][7]
    const float* calib;    // [batch][16]  rect @ Trv2c (float32)
    pp_detection* dets;    // [batch][post_max]
    int* n_dets;           // [batch]
    // page-locked host copies the kernel fills itself (or NULL): the kept detections of a frame are a few hundred
    // bytes -- stored straight over the host link they replace two copy nodes (9 us of a replayed step)
    pp_detection* dets_host;
    int* n_dets_host;
};
void launch_postprocess(const PostParams& p, hipStream_t s);

// loss.hip: training loss at the head maps + gradient with respect to them
struct LossParams {
    int batch;
    int64_t A;             // anchors per frame
    int npx, napl;         // head pixels per frame, anchors per pixel
    int ncls;              // class logits per anchor (num_class; the background column is not encoded)
    const float* head;     // [batch][npx][PP_HEAD_COLS]
    const int* labels;     // [batch][A]  (>0 class, 0 background, -1 ignored)
    const float* reg_targets;  // [batch][A][7]
    const float* anchors;  // [A][7]
    int* npos;             // [batch] scratch: positives per frame
    double* partials;      // [batch * blocks][5] scratch
    float* losses;         // [8] out
    float* head_grad;      // [batch][npx][PP_HEAD_COLS] out, may be NULL
    float alpha, gamma, sigma;
    float code_weight[7];
    float pos_cls_weight, neg_cls_weight;
    float cls_weight, loc_weight, dir_weight;
    int norm_by_num_positives, encode_rad_error_by_sin, use_direction;
};
int loss_blocks(int npx);
int launch_head_loss(const LossParams& p, hipStream_t s);   // 0 or PP_ERR_UNSUPPORTED (anchors per pixel > 3)

// optim.hip: AdamW update of one flat parameter buffer
void launch_adamw(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                  float eps, float wd, hipStream_t s);

// rotate_iou.hip: rotated-box overlaps of the AP evaluator
void launch_riou_corners(const float* boxes, int64_t n, float* corners, hipStream_t s);
void launch_riou_pairs(const float* bc, int64_t N, const float* qc, int64_t K, int criterion, float* out, hipStream_t s);
void launch_d3_finish(const double* boxes, int64_t N, const double* qboxes, int64_t K, int criterion,
                      const float* rinc, double* out, hipStream_t s);
