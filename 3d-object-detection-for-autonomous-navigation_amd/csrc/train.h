// Training step (train.hip): shapes, the flat parameter layout and the device buffers pp_api.hip hands over.
#pragma once

#include <string>
#include <vector>

#include "pp_common.h"

struct TrainEntry {
    std::string name;    // Keras tensor name (weights.py), e.g. "rpn/block2/3/pointwise_kernel"
    int64_t offset;      // floats from the start of the parameter (or state) buffer
    int64_t size;
    int is_state;        // 1: BatchNorm moving statistic (not trained; its own buffer)
};

struct TrainShape {
    int nx, ny, nz, C, F, FA, T, max_voxels, with_dist;
    float vx, vy, x_off, y_off;
    int head_h, head_w, napl, ncls, use_dir, CC;
    std::vector<LayerDesc> layers;   // the engine's layer table (shapes only are read)
};

struct TrainLayerBuf {
    float* D;       // depthwise output [rows][cin] (separable layers)
    float* Z;       // pre-BatchNorm GEMM output [rows][cout] / [pixels][k*k*cout]
    float* A;       // activation [rows][cout]: block-final separable layers only (the transposed convolution and the next
                    // block read it as a tensor; in-block readers evaluate relu(bn(Z)) from Z and coef); NULL otherwise
    float* dA;      // gradient of A (separable layers)
    float* stats;   // [cout][2] batch mean, 1/sqrt(var + eps)
    float* sums;    // [2][cout] scratch of the reductions
    float4* coef;   // [cout] (sc, sh, inv, -mean * inv): act = z * sc + sh, zhat = z * inv + nmi (k_tr_bn_finalize)
    long pw16_off;  // separable layers: offset (16-bit words) of the layer's split pointwise kernel in TrainCtx::pw16
};

struct TrainCtx {
    hipStream_t stream;
    // voxeliser products of the resident frames
    const float* pts_sorted;
    const int* offsets;
    const int* pillar_start;
    const int* pillar_cell;
    const int* npillars;
    const int* cellmap;
    // PFN
    float* pfn_feat;     // [B * max_voxels][C]
    int* pfn_arg;        // [B * max_voxels][C]
    float* pfn_stats;    // [C][2]
    float* pfn_sums;     // [2][C]
    float* pfn_nrows;    // [1] rows of the padded PFN tensor (pillars of the batch * T), computed on the device
    int* pfn_prefix;     // [B + 1] exclusive prefix of the frames' pillar counts
    float4* pfn_rec;     // [B * max_voxels][3] pillar records (row range, slot, canvas row | mean, centre): written by k_tr_pfn_lin
    float* canvas;       // [B][ny][nx][C]
    float* dcanvas;
    // RPN
    std::vector<TrainLayerBuf> lbuf;
    float* cat;          // [B * H' * W'][CC]
    float* dcat;
    float* head;         // [B * H' * W'][32] (the loss kernel's layout)
    float* dhead;
    float* head_w;       // [CC][32] packed head kernels
    float* head_b;       // [32]
    float* dhead_w;
    float* dhead_b;      // [2][32] scratch (row 0 = the bias gradient)
    float* dZ;           // scratch [max rows * cout]
    float* dD;           // scratch [max rows * cin]
    float* part;         // partial sums of the persistent reductions
    float* stat_part;    // BatchNorm statistics partials of the forward products: [row tiles][2][GEMM columns]
    float* gemm_part;    // split-K partial tiles
    long gemm_part_floats;
    // fused training forward of the separable layers (k_sep_u<..., TR = 1>): every pointwise kernel of the step as two
    // float16 pieces, [cin / 16][2][cout][16] per layer (k_tr_split_pw, once per step); NULL: not available.  The
    // maps those launches read (canvas, Z of in-block layers, A of block-final ones) carry a NaN header in front.
    unsigned short* pw16;
    unsigned short* head_w16;   // the packed head matrix [CC][32] in the same form
    long stat_part_floats;
};

std::vector<TrainEntry> train_layout(const TrainShape& s, int64_t* n_params, int64_t* n_state);
size_t train_part_floats(const TrainShape& s);
// forward (training mode) + loss + backward for `batch` resident, voxelised frames; grads overwritten, state updated
int train_step(const TrainCtx& cx, const TrainShape& s, const std::vector<TrainEntry>& layout, const float* params,
               float* grads, float* state, int batch, const LossParams& loss, int phase = 3);

// parity tap (pp_train_fetch_decisions): mask[i] = 1 where the backward pass lets the gradient through element i of
// Z[n] ([rows][C]; the test the BatchNorm-backward kernels make: fmaf(z, sc, sh) > 0)
void launch_relu_mask(const float* Z, const float4* coef, long n, int C, unsigned char* mask, hipStream_t s);
