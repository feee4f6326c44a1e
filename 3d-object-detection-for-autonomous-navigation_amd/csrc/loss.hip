// Training loss at the head maps and its gradient with respect to them.
//
// Replaces the loss half of VoxelNet.call in training mode (reference model/voxelnet.py:922-1049):
//   prepare_loss_weights                     :461-512  cls / reg weights, NormByNumPositives per frame
//   create_loss + add_sin_difference         :74-155, :63-69
//   sigmoid_focal_classification_loss        :262-364  (with _sigmoid_cross_entropy_with_logits :237-260)
//   WeightedSmoothL1LocalizationLoss.call    :407-459
//   get_direction_target + weighted softmax  :38-46, :180-235
//   reductions and loss weights              :964-1031
// One thread per head pixel: it reads the pixel's 128-byte head row once ([box napl*7 | cls napl | dir
// napl*2 | pad], the layout the fused deconv epilogues write), evaluates its napl anchors in float32 and
// writes the 128-byte gradient row.  HBM-bound elementwise work: 256 B per pixel + 36 B per anchor.
// The five sums are reduced in float64 per workgroup and added in a fixed order (bit-reproducible).
#include "pp_common.h"

#define LT 256
#define LOSS_NSUM 5   // loc, cls, dir, cls on positives, cls on negatives

__global__ __launch_bounds__(LT) void k_loss_count(const int* __restrict__ labels, int64_t A, int* __restrict__ npos) {
    __shared__ int s_part[LT / 64];
    const int b = blockIdx.x;
    int c = 0;
    const int* lab = labels + (size_t)b * A;
    if ((A & 3) == 0) {   // 16-byte loads, several in flight (the kernel is one memory round trip long)
        const int4* l4 = reinterpret_cast<const int4*>(lab);
        const int64_t n4 = A >> 2;
#pragma unroll 4
        for (int64_t a = threadIdx.x; a < n4; a += LT) {
            const int4 v = l4[a];
            c += (v.x > 0) + (v.y > 0) + (v.z > 0) + (v.w > 0);
        }
    } else {
        for (int64_t a = threadIdx.x; a < A; a += LT) c += lab[a] > 0 ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < LT / 64; ++w) t += s_part[w];
        npos[b] = t;
    }
}

// NAPL / NCLS: anchors per pixel and class logits per anchor as compile-time constants (every index into the two
// 32-float rows is static: they stay in registers).  Head row: [box NAPL*7 | cls NAPL*NCLS | dir NAPL*2 | pad].
template <int NAPL, int NCLS>
__global__ __launch_bounds__(LT) void k_loss_pixels(LossParams p) {
    __shared__ double s_red[LT / 64][LOSS_NSUM];
    // the workgroup's 256 head rows pass through LDS: coalesced 16-byte global accesses on one side (a
    // thread reading its own 128-byte row touches 64 different lines per instruction), row-per-thread
    // on the other (row stride 33 floats: conflict-free)
    constexpr int RS = PP_HEAD_COLS + 1;
    __shared__ float s_rows[LT * RS];
    const int b = blockIdx.y;
    const int px0 = blockIdx.x * LT;
    const int px = px0 + threadIdx.x;
    constexpr int napl = NAPL, nb = NAPL * 7, nc = NAPL * NCLS;
    double acc[LOSS_NSUM] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const int nrows = min(LT, p.npx - px0);
    {
        const float4* h4 = reinterpret_cast<const float4*>(p.head + ((size_t)b * p.npx + px0) * PP_HEAD_COLS);
        for (int e = threadIdx.x; e < nrows * (PP_HEAD_COLS / 4); e += LT) {
            const float4 t = h4[e];
            float* d = s_rows + (e >> 3) * RS + (e & 7) * 4;
            d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        }
    }
    __syncthreads();
    float grow[PP_HEAD_COLS];
#pragma unroll
    for (int q = 0; q < PP_HEAD_COLS; ++q) grow[q] = 0.f;
    if (px < p.npx) {
        float hrow[PP_HEAD_COLS];
#pragma unroll
        for (int q = 0; q < PP_HEAD_COLS; ++q) hrow[q] = s_rows[threadIdx.x * RS + q];
        const float norm = p.norm_by_num_positives ? fminf(fmaxf((float)p.npos[b], 1.f), 100000.f) : 1.f;
        const float dir_norm = fminf(fmaxf((float)p.npos[b], 1.f), 9999999.f);
        const float inv_b = 1.f / (float)p.batch;
        const float s2 = p.sigma * p.sigma;
#pragma unroll
        for (int r = 0; r < napl; ++r) {
            const size_t a = (size_t)px * napl + r;
            const int label = p.labels[(size_t)b * p.A + a];
            const float pos = label > 0 ? 1.f : 0.f, neg = label == 0 ? 1.f : 0.f;
            // ---- classification: sigmoid focal loss on the one-hot target without the background column
            // (create_loss: one_hot(labels * cared, num_class + 1)[..., 1:]; class k is logit column k - 1) ----
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const float x = hrow[nb + r * NCLS + c];
                const float t = (label == c + 1) ? 1.f : 0.f;
                const float w = (neg * p.neg_cls_weight + pos * p.pos_cls_weight) / norm;
                const float ce = fminf(fmaxf(x, 0.f), 10000.f) - x * t + log1pf(expf(-fabsf(x)));
                const float pr = 1.f / (1.f + expf(-x));
                const float p_t = t * pr + (1.f - t) * (1.f - pr);
                const float om = 1.f - p_t;
                const float mod = (p.gamma != 0.f) ? powf(om, p.gamma) : 1.f;
                const float aw = (p.alpha >= 0.f) ? (t * p.alpha + (1.f - t) * (1.f - p.alpha)) : 1.f;
                const float l = mod * aw * ce * w;
                acc[1] += (double)l;
                if (NCLS == 1) {                 // _get_pos_neg_loss (model/voxelnet.py:48-61): by the anchor's label ...
                    acc[3] += (double)(pos * l);
                    acc[4] += (double)(neg * l);
                } else if (c == 0) {             // ... or, with several class columns, column 0 against the rest
                    acc[4] += (double)l;
                } else {
                    acc[3] += (double)l;
                }
                // d/dx: ce' = sigmoid(x) - t (inside the clip range); (1 - p_t)' = (1 - 2t) p (1 - p)
                float dmod = 0.f;
                if (p.gamma != 0.f) dmod = p.gamma * powf(om, p.gamma - 1.f) * (1.f - 2.f * t) * pr * (1.f - pr);
                grow[nb + r * NCLS + c] = w * aw * (dmod * ce + mod * (pr - t)) * inv_b * p.cls_weight;
            }
            // ---- localisation: smooth L1 (sigma) on the code, the angle through sin(a - b) = sin a cos b - cos a sin b ----
            {
                const float w = pos / norm;
                const float* g = p.reg_targets + ((size_t)b * p.A + a) * 7;
                float lsum = 0.f;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const float e = hrow[r * 7 + i], gt = g[i];
                    float diff, dde = 1.f;
                    if (i == 6 && p.encode_rad_error_by_sin) {
                        diff = sinf(e) * cosf(gt) - cosf(e) * sinf(gt);
                        dde = cosf(e) * cosf(gt) + sinf(e) * sinf(gt);
                    } else {
                        diff = e - gt;
                    }
                    diff *= p.code_weight[i];
                    const float ad = fabsf(diff);
                    const bool lt = ad <= 1.f / s2;
                    const float l = lt ? 0.5f * (ad * p.sigma) * (ad * p.sigma) : ad - 0.5f / s2;
                    lsum += l * w;
                    const float dl = lt ? s2 * diff : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
                    grow[r * 7 + i] = w * dl * p.code_weight[i] * dde * inv_b * p.loc_weight;
                }
                acc[0] += (double)lsum;
            }
            // ---- direction: softmax cross entropy against (target angle + anchor angle > 0) ----
            if (p.use_direction) {
                const float w = pos / dir_norm;
                const float rot_gt = p.reg_targets[((size_t)b * p.A + a) * 7 + 6] + p.anchors[a * 7 + 6];
                const int c = rot_gt > 0.f ? 1 : 0;
                const float d0 = hrow[nb + nc + 2 * r], d1 = hrow[nb + nc + 2 * r + 1];
                const float m = fmaxf(d0, d1);
                const float e0 = expf(d0 - m), e1 = expf(d1 - m);
                const float lse = m + logf(e0 + e1);
                acc[2] += (double)((lse - (c ? d1 : d0)) * w);
                const float s0 = e0 / (e0 + e1), s1 = e1 / (e0 + e1);
                grow[nb + nc + 2 * r] = w * (s0 - (c ? 0.f : 1.f)) * inv_b * p.dir_weight;
                grow[nb + nc + 2 * r + 1] = w * (s1 - (c ? 1.f : 0.f)) * inv_b * p.dir_weight;
            }
        }
    }
    if (p.head_grad != nullptr) {   // gradient rows back through LDS (every thread has read its head row)
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PP_HEAD_COLS; ++q) s_rows[threadIdx.x * RS + q] = grow[q];
        __syncthreads();
        float4* g4 = reinterpret_cast<float4*>(p.head_grad + ((size_t)b * p.npx + px0) * PP_HEAD_COLS);
        for (int e = threadIdx.x; e < nrows * (PP_HEAD_COLS / 4); e += LT) {
            const float* d = s_rows + (e >> 3) * RS + (e & 7) * 4;
            g4[e] = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
    // workgroup sums (float64): wave shuffle, then the four waves in order
#pragma unroll
    for (int k = 0; k < LOSS_NSUM; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_NSUM) {
        double t = 0.0;
        for (int w = 0; w < LT / 64; ++w) t += s_red[w][threadIdx.x];
        p.partials[((size_t)b * gridDim.x + blockIdx.x) * LOSS_NSUM + threadIdx.x] = t;
    }
}

// fixed-order sum of the workgroup partials -> the reported scalars: lane l adds partials l, l + 64, ... in
// order, then the 64 lane sums are added by a shuffle tree (the same tree every time)
__global__ __launch_bounds__(64) void k_loss_finish(LossParams p, int nblocks) {
    const int lane = threadIdx.x;
    const int n = p.batch * nblocks;
    double sum[LOSS_NSUM];
#pragma unroll
    for (int k = 0; k < LOSS_NSUM; ++k) sum[k] = 0.0;
#pragma unroll 4
    for (int i = lane; i < n; i += 64)
#pragma unroll
        for (int k = 0; k < LOSS_NSUM; ++k) sum[k] += p.partials[(size_t)i * LOSS_NSUM + k];
    int np = 0;
    for (int b = lane; b < p.batch; b += 64) np += p.npos[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < LOSS_NSUM; ++k) sum[k] += __shfl_down(sum[k], off);
        np += __shfl_down(np, off);
    }
    if (lane == 0) {
        const float fb = (float)p.batch;
        const float loc = (float)sum[0] / fb * p.loc_weight;
        const float cls = (float)sum[1] / fb * p.cls_weight;
        const float dir = p.use_direction ? (float)sum[2] / fb * p.dir_weight : 0.f;
        p.losses[0] = loc + cls + dir;
        p.losses[1] = loc;
        p.losses[2] = cls;
        p.losses[3] = dir;
        p.losses[4] = (float)sum[3] / fb;
        p.losses[5] = (float)sum[4] / fb;
        p.losses[6] = (float)np;
        p.losses[7] = 0.f;
    }
}

int loss_blocks(int npx) { return (npx + LT - 1) / LT; }

template <int NAPL>
static int launch_pixels(const LossParams& p, dim3 grid, hipStream_t s) {
    switch (p.ncls) {
        case 1: hipLaunchKernelGGL((k_loss_pixels<NAPL, 1>), grid, dim3(LT), 0, s, p); return 0;
        case 2: hipLaunchKernelGGL((k_loss_pixels<NAPL, 2>), grid, dim3(LT), 0, s, p); return 0;
        case 3: hipLaunchKernelGGL((k_loss_pixels<NAPL, 3>), grid, dim3(LT), 0, s, p); return 0;
        case 4: hipLaunchKernelGGL((k_loss_pixels<NAPL, 4>), grid, dim3(LT), 0, s, p); return 0;
        default: return PP_ERR_UNSUPPORTED;
    }
}

int launch_head_loss(const LossParams& p, hipStream_t s) {
    if (p.batch <= 0) return 0;
    // 7 + ncls + 2 head columns per anchor (the direction pair only with the direction head), 32 per row
    if (p.napl < 1 || p.napl > 3 || p.ncls < 1 || p.ncls > 4 ||
        p.napl * (7 + p.ncls + (p.use_direction ? 2 : 0)) > PP_HEAD_COLS) return PP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_loss_count, dim3(p.batch), dim3(LT), 0, s, p.labels, p.A, p.npos);
    const int nblocks = loss_blocks(p.npx);
    const dim3 grid(nblocks, p.batch);
    int st;
    if (p.napl == 1) st = launch_pixels<1>(p, grid, s);
    else if (p.napl == 2) st = launch_pixels<2>(p, grid, s);
    else st = launch_pixels<3>(p, grid, s);
    if (st) return st;
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64), 0, s, p, nblocks);
    return 0;
}
