// Training step of the PointPillars network: training-mode forward pass
// (batch-statistics BatchNorm, activations kept), the loss at the head maps (loss.hip), and the backward
// pass down to every trainable tensor.
//
// Replaces VoxelNet.call in training mode (reference model/voxelnet.py:850-1049) with the layers it runs
//   PillarFeatureNet  Dense(no bias) -> BatchNormalization(eps 1e-3, momentum 0.01) -> ReLU -> max over T rows
//                     (model/pointpillars.py:97-115, :128-225), PointPillarsScatter (:285-341),
//   RPN               SeparableConv2D -> BatchNormalization -> ReLU blocks, Conv2DTranspose -> BN -> ReLU,
//                     concat, three 1x1 heads with bias (model/voxelnet.py:573-717),
// and the tf.GradientTape gradient of `loss` with respect to net.trainable_variables (train.py:265-304).
//
// Layout: float32 NHWC activations, one [rows][channels] matrix per tensor; the parameters live in ONE flat
// device buffer owned by the caller (order: train_layout()), gradients go to a second flat buffer of the same
// order -- what the AdamW kernel (optim.hip) and the data-parallel all-reduce consume.
//   * the 1x1 convolutions, the transposed convolutions (kernel == stride: a GEMM per input pixel) and the heads
//     run as float32 MFMA GEMMs (v_mfma_f32_32x32x2_f32), forward NN / NT, input gradients with the other
//     operand transposed, weight gradients as split-K TN GEMMs (K = pixels) reduced in a fixed order;
//   * depthwise 3x3, BatchNorm statistics / normalisation / backward, the PFN and the scatter are HBM-bound
//     element kernels; every reduction goes through per-workgroup partial sums added in a fixed order, so a
//     step is bit-reproducible (no floating-point atomics).
// BatchNorm in training mode normalises with the batch mean and the biased batch variance; the moving
// statistics are updated as Keras does (moving = moving * momentum + batch * (1 - momentum); the RPN's fused
// BatchNorm feeds the unbiased variance, the PFN's rank-3 BatchNorm the biased one).  The PFN statistics run
// over ALL P * T rows of the reference's padded tensor: the zero rows contribute nothing to the sums but count
// in N, and a padded row that wins the max receives the gradient (it only reaches beta / gamma / the statistics).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "pp_common.h"
#include "train.h"

typedef float tf32x16 __attribute__((ext_vector_type(16)));

#define TR_EPS 1e-3f
#define TR_NPART 256   // workgroups of the persistent reduction kernels (= rows of their partial-sum buffers)

// ------------------------------------------------------------------------------------------------------------
// float32 MFMA GEMM:  C[M][N] (+)= A(m,k) * B(k,n) [+ bias(n)],  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// 64 x 64 tile per 4-wave workgroup (one 32 x 32 MFMA accumulator per wave), K in chunks of 16 through LDS with
// the next chunk's global loads in flight.  grid.z > 1: split-K, slice z writes its partial tile to Cpart[z][M][N].
// ------------------------------------------------------------------------------------------------------------
struct TGemm {
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    float* C; long ldc;
    int M, N, K;
    const float* bias;
    int accumulate;
    int kper;        // K range per z slice (multiple of 16)
    float* Cpart;    // split-K partials (ksplit > 1)
};

__global__ __launch_bounds__(256) void k_tr_gemm(TGemm g) {
    constexpr int LD = 64 + 4;
    __shared__ float As[16 * LD];
    __shared__ float Bs[16 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int kbeg = blockIdx.z * g.kper, kend = min(g.K, kbeg + g.kper);
    // element (mi, ki) of the A tile handled by this thread in round i; consecutive threads walk the unit-stride axis
    int am[4], ak[4], bk[4], bn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        if (g.sak == 1) { ak[i] = idx & 15; am[i] = idx >> 4; } else { am[i] = idx & 63; ak[i] = idx >> 6; }
        if (g.sbn == 1) { bn[i] = idx & 63; bk[i] = idx >> 6; } else { bk[i] = idx & 15; bn[i] = idx >> 4; }
    }
    float ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + am[i], k = k0 + ak[i];
            ra[i] = (m < g.M && k < kend) ? g.A[(long)m * g.sam + (long)k * g.sak] : 0.f;
            const int kb = k0 + bk[i], n = n0 + bn[i];
            rb[i] = (n < g.N && kb < kend) ? g.B[(long)kb * g.sbk + (long)n * g.sbn] : 0.f;
        }
    };
    tf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int h = lane >> 5, r32 = lane & 31;
    if (kbeg < kend) load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            As[ak[i] * LD + am[i]] = ra[i];
            Bs[bk[i] * LD + bn[i]] = rb[i];
        }
        __syncthreads();
        if (k0 + 16 < kend) load(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a = As[(kk + h) * LD + wm * 32 + r32];
            const float b = Bs[(kk + h) * LD + wn * 32 + r32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // D[row][col]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int n = n0 + wn * 32 + r32;
    if (n >= g.N) return;
    const float bv = (g.bias != nullptr) ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= g.M) continue;
        if (gridDim.z > 1) {
            g.Cpart[((size_t)blockIdx.z * g.M + m) * g.N + n] = acc[r];
        } else {
            float* c = g.C + (size_t)m * g.ldc + n;
            const float v = acc[r] + bv;
            *c = g.accumulate ? (*c + v) : v;
        }
    }
}

// out[i] (+)= scale * sum_p part[p][i] in a fixed order (deterministic): 16 lanes share an output element, lane l
// adds parts l, l + 16, ... and the 16 lane sums are added by a shuffle tree
__global__ __launch_bounds__(256) void k_tr_reduce(const float* __restrict__ part, int nparts, long n, long pstride,
                                                   float* __restrict__ out, long ldo, int ncols, int accumulate, float scale) {
    const int l = threadIdx.x & 15;
    const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    float s = 0.f;
    if (i < n)
        for (int p = l; p < nparts; p += 16) s += part[(size_t)p * pstride + i];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (i >= n || l != 0) return;
    s *= scale;
    // optional re-striding of the output ([rows][ncols] with leading dimension ldo)
    float* o = (ncols > 0) ? out + (i / ncols) * ldo + (i % ncols) : out + i;
    *o = accumulate ? (*o + s) : s;
}

static unsigned reduce_blocks(long n) { return (unsigned)((n + 15) / 16); }

static void tr_gemm(const TrainCtx& cx, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                    long ldc, int M, int N, int K, const float* bias, int accumulate, int ksplit) {
    TGemm g;
    g.A = A; g.sam = sam; g.sak = sak; g.B = B; g.sbk = sbk; g.sbn = sbn; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.accumulate = accumulate; g.Cpart = cx.gemm_part;
    if (ksplit < 1) ksplit = 1;
    int kper = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
    ksplit = (K + kper - 1) / kper;
    g.kper = kper;
    dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
    PP_LAUNCH("k_tr_gemm", k_tr_gemm, grid, dim3(256), 0, cx.stream, g);
    if (ksplit > 1) {
        const long n = (long)M * N;
        PP_LAUNCH("k_tr_reduce", k_tr_reduce, dim3(reduce_blocks(n)), dim3(256), 0, cx.stream,
                  (const float*)cx.gemm_part, ksplit, n, n, C, ldc, N, accumulate, 1.0f);
    }
}

// weight gradients: K = rows (pixels); enough slices to fill the chip, bounded by the partial buffer
static int wgrad_split(const TrainCtx& cx, int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    long want = std::max<long>(1, 1024 / tiles);
    want = std::min<long>(want, (K + 255) / 256);
    while (want > 1 && want * (long)M * N > cx.gemm_part_floats) --want;
    return (int)want;
}

// ------------------------------------------------------------------------------------------------------------
// element kernels
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tr_fill(float* p, long n, float v) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// depthwise 3x3, symmetric padding 1, stride S (ZeroPadding2D(1) + 'valid' == 'same' for stride 1):
// D[b,y,x,c] = sum_t X[b, y*S-1+dy, x*S-1+dx, c] * w[t][c]
__global__ __launch_bounds__(256) void k_tr_dw_fwd(const float* __restrict__ X, const float* __restrict__ w,
                                                   float* __restrict__ D, int B, int ih, int iw, int oh, int ow, int C, int S) {
    const int c4n = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * oh * ow * c4n;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    long p = i / c4n;
    const int x = (int)(p % ow); p /= ow;
    const int y = (int)(p % oh);
    const int b = (int)(p / oh);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y * S - 1 + dy;
        if ((unsigned)yy >= (unsigned)ih) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x * S - 1 + dx;
            if ((unsigned)xx >= (unsigned)iw) continue;
            const float4 v = *reinterpret_cast<const float4*>(X + (((size_t)b * ih + yy) * iw + xx) * C + c);
            const float4 k = *reinterpret_cast<const float4*>(w + (size_t)(dy * 3 + dx) * C + c);
            o.x = fmaf(v.x, k.x, o.x); o.y = fmaf(v.y, k.y, o.y); o.z = fmaf(v.z, k.z, o.z); o.w = fmaf(v.w, k.w, o.w);
        }
    }
    *reinterpret_cast<float4*>(D + (size_t)(i / c4n) * C + c) = o;
}

// gradient of the depthwise convolution with respect to its input:
// dX[b,yy,xx,c] (+)= sum over taps with (yy+1-dy) % S == 0 ... of dD[b,(yy+1-dy)/S,(xx+1-dx)/S,c] * w[t][c]
__global__ __launch_bounds__(256) void k_tr_dw_bwd_in(const float* __restrict__ dD, const float* __restrict__ w,
                                                      float* __restrict__ dX, int B, int ih, int iw, int oh, int ow, int C,
                                                      int S, int accumulate) {
    const int c4n = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * ih * iw * c4n;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    long p = i / c4n;
    const int xx = (int)(p % iw); p /= iw;
    const int yy = (int)(p % ih);
    const int b = (int)(p / ih);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ty = yy + 1 - dy;
        if (ty < 0 || ty % S != 0) continue;
        const int y = ty / S;
        if (y >= oh) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int tx = xx + 1 - dx;
            if (tx < 0 || tx % S != 0) continue;
            const int x = tx / S;
            if (x >= ow) continue;
            const float4 v = *reinterpret_cast<const float4*>(dD + (((size_t)b * oh + y) * ow + x) * C + c);
            const float4 k = *reinterpret_cast<const float4*>(w + (size_t)(dy * 3 + dx) * C + c);
            o.x = fmaf(v.x, k.x, o.x); o.y = fmaf(v.y, k.y, o.y); o.z = fmaf(v.z, k.z, o.z); o.w = fmaf(v.w, k.w, o.w);
        }
    }
    float4* dst = reinterpret_cast<float4*>(dX + (size_t)(i / c4n) * C + c);
    if (accumulate) { const float4 q = *dst; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
    *dst = o;
}

// gradient of the depthwise kernel: part[blk][t][c] = sum over this workgroup's output pixels of X(window t) * dD.
// 256 / C pixel lanes per workgroup (thread = channel x pixel lane), workgroups stride over the pixels;
// k_tr_reduce adds the TR_NPART partial rows.
__global__ __launch_bounds__(256) void k_tr_dw_bwd_w(const float* __restrict__ X, const float* __restrict__ dD,
                                                     float* __restrict__ part, int B, int ih, int iw, int oh, int ow,
                                                     int C, int S) {
    __shared__ float sred[9 * 256];
    const int tid = threadIdx.x;
    const int lanes_per_row = (C >= 256) ? 256 : C;      // C in {32, 64, 128, 256}
    const int rsub = tid / lanes_per_row, nsub = 256 / lanes_per_row;
    const int c = tid % lanes_per_row;
    const long npix = (long)B * oh * ow;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (long p = (long)blockIdx.x * nsub + rsub; p < npix; p += (long)gridDim.x * nsub) {
        const int x = (int)(p % ow);
        const int y = (int)((p / ow) % oh);
        const int b = (int)(p / ((long)ow * oh));
        const float g = dD[(size_t)p * C + c];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y * S - 1 + dy;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x * S - 1 + dx;
                if ((unsigned)yy < (unsigned)ih && (unsigned)xx < (unsigned)iw)
                    acc[dy * 3 + dx] = fmaf(X[(((size_t)b * ih + yy) * iw + xx) * C + c], g, acc[dy * 3 + dx]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) sred[t * 256 + tid] = acc[t];
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = acc[t];
            for (int k = 1; k < nsub; ++k) v += sred[t * 256 + tid + k * lanes_per_row];
            part[((size_t)blockIdx.x * 9 + t) * C + c] = v;
        }
    }
}

// per-channel sums over the rows of Z[rows][C]: part[blk][0][c] = sum z, part[blk][1][c] = sum z^2
__global__ __launch_bounds__(256) void k_tr_colstats(const float* __restrict__ Z, long rows, int C, float* __restrict__ part) {
    __shared__ float s1[256], s2[256];
    const int tid = threadIdx.x;
    const int lanes_per_row = (C >= 256) ? 256 : C;      // C in {32, 64, 128, 256}
    const int rsub = tid / lanes_per_row, nsub = 256 / lanes_per_row;
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + tid % lanes_per_row;
        float a = 0.f, q = 0.f;
        for (long r = (long)blockIdx.x * nsub + rsub; r < rows; r += (long)gridDim.x * nsub) {
            const float z = Z[(size_t)r * C + c];
            a += z; q = fmaf(z, z, q);
        }
        s1[tid] = a; s2[tid] = q;
        __syncthreads();
        if (rsub == 0) {
            for (int k = 1; k < nsub; ++k) { a += s1[tid + k * lanes_per_row]; q += s2[tid + k * lanes_per_row]; }
            part[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
            part[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
        }
        __syncthreads();
    }
}

// BatchNorm statistics from the per-workgroup partial sums part[p][0/1][c] (16 lanes per channel add the partial
// rows in a fixed order -- the reduction and the finalisation in one launch): stats[c] = (mean, 1/sqrt(var + eps))
// with the biased batch variance; moving statistics updated in place as Keras does
__global__ __launch_bounds__(256) void k_tr_bn_finalize(const float* __restrict__ part, int nparts, int C, float n_rows_arg,
                                                        const float* __restrict__ n_rows_dev, float momentum,
                                                        int unbiased_moving, float* __restrict__ stats,
                                                        float* __restrict__ moving_mean, float* __restrict__ moving_var) {
    const int l = threadIdx.x & 15;
    const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
    float s1 = 0.f, s2 = 0.f;
    if (c < C)
        for (int p = l; p < nparts; p += 16) {
            s1 += part[((size_t)p * 2 + 0) * C + c];
            s2 += part[((size_t)p * 2 + 1) * C + c];
        }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if (c >= C || l != 0) return;
    const float n_rows = fmaxf((n_rows_dev != nullptr) ? *n_rows_dev : n_rows_arg, 1.f);   // PFN: P * T, known on the device only
    const float mean = s1 / n_rows;
    float var = s2 / n_rows - mean * mean;
    var = fmaxf(var, 0.f);
    stats[2 * c] = mean;
    stats[2 * c + 1] = 1.0f / sqrtf(var + TR_EPS);
    if (moving_mean != nullptr) {
        const float vm = (unbiased_moving && n_rows > 1.f) ? var * (n_rows / (n_rows - 1.f)) : var;
        moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
        moving_var[c] = moving_var[c] * momentum + vm * (1.f - momentum);
    }
}

// row of the [rows][C] matrix -> pixel row of the destination activation (identity, or the pixel shuffle of a
// transposed convolution with kernel == stride k: row = input pixel * k*k + tap)
struct RowMap { int k, in_h, in_w; };
__device__ __forceinline__ long map_row(const RowMap& m, long r) {
    if (m.k <= 1) return r;
    const int kk = m.k * m.k;
    const int tap = (int)(r % kk);
    long p = r / kk;
    const int x = (int)(p % m.in_w); p /= m.in_w;
    const int y = (int)(p % m.in_h);
    const long b = p / m.in_h;
    const int ti = tap / m.k, tj = tap - ti * m.k;
    return ((b * m.in_h * m.k + (long)y * m.k + ti) * ((long)m.in_w * m.k)) + (long)x * m.k + tj;
}

// A[map(r)][co_off + c] = relu((Z[r][c] - mean) * inv * gamma + beta)
__global__ __launch_bounds__(256) void k_tr_bn_relu(const float* __restrict__ Z, long rows, int C, const float* __restrict__ stats,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ A, int ld, int co_off, RowMap rm) {
    const int c4n = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const long r = i / c4n;
    const float4 z = *reinterpret_cast<const float4*>(Z + (size_t)r * C + c);
    const float zz[4] = {z.x, z.y, z.z, z.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = (zz[j] - stats[2 * (c + j)]) * stats[2 * (c + j) + 1] * gamma[c + j] + beta[c + j];
        o[j] = fmaxf(v, 0.f);
    }
    *reinterpret_cast<float4*>(A + (size_t)map_row(rm, r) * ld + co_off + c) = make_float4(o[0], o[1], o[2], o[3]);
}

// backward of BN + ReLU, pass 1: g = dA * (A > 0); part[blk][0][c] = sum g, part[blk][1][c] = sum g * zhat
__global__ __launch_bounds__(256) void k_tr_bn_bwd_reduce(const float* __restrict__ dA, int ld, int co_off, RowMap rm,
                                                          const float* __restrict__ Z, long rows, int C,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ part) {
    __shared__ float s1[256], s2[256];
    const int tid = threadIdx.x;
    const int lanes_per_row = (C >= 256) ? 256 : C;
    const int rsub = tid / lanes_per_row, nsub = 256 / lanes_per_row;
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + tid % lanes_per_row;
        const float mean = stats[2 * c], inv = stats[2 * c + 1], ga = gamma[c], be = beta[c];
        float a = 0.f, q = 0.f;
        for (long r = (long)blockIdx.x * nsub + rsub; r < rows; r += (long)gridDim.x * nsub) {
            const float zh = (Z[(size_t)r * C + c] - mean) * inv;
            const float act = zh * ga + be;
            const float g = (act > 0.f) ? dA[(size_t)map_row(rm, r) * ld + co_off + c] : 0.f;
            a += g; q = fmaf(g, zh, q);
        }
        s1[tid] = a; s2[tid] = q;
        __syncthreads();
        if (rsub == 0) {
            for (int k = 1; k < nsub; ++k) { a += s1[tid + k * lanes_per_row]; q += s2[tid + k * lanes_per_row]; }
            part[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
            part[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
        }
        __syncthreads();
    }
}

// pass 2: dZ[r][c] = gamma * inv * (g - Sg / n - zhat * Sgz / n)   (sums[0] = Sg = d beta, sums[1] = Sgz = d gamma)
__global__ __launch_bounds__(256) void k_tr_bn_bwd_apply(const float* __restrict__ dA, int ld, int co_off, RowMap rm,
                                                         const float* __restrict__ Z, long rows, int C,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ sums,
                                                         float n_rows, float* __restrict__ dZ) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C) return;
    const int c = (int)(i % C);
    const long r = i / C;
    const float mean = stats[2 * c], inv = stats[2 * c + 1], ga = gamma[c];
    const float zh = (Z[i] - mean) * inv;
    const float act = zh * ga + beta[c];
    const float g = (act > 0.f) ? dA[(size_t)map_row(rm, r) * ld + co_off + c] : 0.f;
    dZ[i] = ga * inv * (g - sums[c] / n_rows - zh * (sums[C + c] / n_rows));
}

// ------------------------------------------------------------------------------------------------------------
// PillarFeatureNet, training mode.  One wavefront per pillar (persistent grid), lanes = channels (C / 64 per lane).
// Rows of a pillar = its first min(count, T) points in the pillar-sorted order the voxeliser leaves.
// ------------------------------------------------------------------------------------------------------------
struct PfnT {
    int batch, nx, ny, C, F, FA, T, max_voxels, with_distance;
    float vx, vy, x_off, y_off;
    const float* pts_sorted;   // [sum N][F]
    const int* offsets;        // [batch + 1]
    const int* pillar_start;   // [batch][max_voxels + 1]
    const int* pillar_cell;    // [batch][max_voxels] linear (z, y, x) cell
    const int* npillars;       // [batch]
    const float* W;            // [FA][C]
};

// the decorated features of point j of a pillar (wave-uniform inputs): raw F | xyz - mean | xy - centre | [norm]
// (f[0..9], unused tail zero; written with selects so that f stays in registers)
template <int CPL>
__device__ __forceinline__ void pfn_row_features(const PfnT& p, const float* q, float mx, float my, float mz, float cx,
                                                 float cy, float (&f)[10]) {
    const bool f4 = p.F > 3;
    const float x = q[0], y = q[1], z = q[2], it = f4 ? q[3] : 0.f;
    const float e0 = x - mx, e1 = y - my, e2 = z - mz, e3 = x - cx, e4 = y - cy;
    const float e5 = p.with_distance ? sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z))) : 0.f;
    f[0] = x; f[1] = y; f[2] = z;
    f[3] = f4 ? it : e0; f[4] = f4 ? e0 : e1; f[5] = f4 ? e1 : e2; f[6] = f4 ? e2 : e3; f[7] = f4 ? e3 : e4;
    f[8] = f4 ? e4 : e5; f[9] = f4 ? e5 : 0.f;
}

__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// pillar geometry shared by the PFN kernels: frame, pillar id -> row range, mean, centre
struct PillarInfo { int b, pid, n; long row0; float mx, my, mz, cx, cy; };
__device__ __forceinline__ bool pillar_info(const PfnT& p, long gp, int lane, PillarInfo& o) {
    o.b = (int)(gp / p.max_voxels);
    o.pid = (int)(gp - (long)o.b * p.max_voxels);
    if (o.b >= p.batch || o.pid >= p.npillars[o.b]) return false;
    const int* ps = p.pillar_start + (size_t)o.b * (p.max_voxels + 1);
    const int start = ps[o.pid];
    o.n = min(ps[o.pid + 1] - start, p.T);
    o.row0 = (long)p.offsets[o.b] + start;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int j = lane; j < o.n; j += 64) {
        const float* q = p.pts_sorted + (size_t)(o.row0 + j) * p.F;
        sx += q[0]; sy += q[1]; sz += q[2];
    }
    sx = wave_sum_f(sx); sy = wave_sum_f(sy); sz = wave_sum_f(sz);
    const float fn = (float)o.n;
    o.mx = sx / fn; o.my = sy / fn; o.mz = sz / fn;
    const int cell = p.pillar_cell[(size_t)o.b * p.max_voxels + o.pid];
    const int xi = cell % p.nx, yi = (cell / p.nx) % p.ny;
    o.cx = __fadd_rn(__fmul_rn((float)xi, p.vx), p.x_off);      // model/pointpillars.py:156-171
    o.cy = __fadd_rn(__fmul_rn((float)yi, p.vy), p.y_off);
    return true;
}

// Y[row][c] = features(row) . W[:, c]; part[blk][0/1][c] = sums of y, y^2 over this workgroup's rows
template <int CPL>
__global__ __launch_bounds__(256) void k_tr_pfn_lin(PfnT p, float* __restrict__ Y, float* __restrict__ part) {
    __shared__ float sp[4][2][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    float w[10][CPL];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int q = 0; q < CPL; ++q) w[k][q] = (k < p.FA && lane * CPL + q < C) ? p.W[k * C + lane * CPL + q] : 0.f;
    float s1[CPL], s2[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) s1[q] = s2[q] = 0.f;
    const long total = (long)p.batch * p.max_voxels;
    for (long gp = (long)blockIdx.x * 4 + wave; gp < total; gp += (long)gridDim.x * 4) {
        PillarInfo pi;
        if (!pillar_info(p, gp, lane, pi)) continue;
        for (int j = 0; j < pi.n; ++j) {
            float f[10];
            pfn_row_features<CPL>(p, p.pts_sorted + (size_t)(pi.row0 + j) * p.F, pi.mx, pi.my, pi.mz, pi.cx, pi.cy, f);
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                float y = 0.f;
#pragma unroll
                for (int k = 0; k < 10; ++k) y = fmaf(f[k], w[k][q], y);   // rows >= FA of w are zero
                if (lane * CPL + q < C) Y[(size_t)(pi.row0 + j) * C + lane * CPL + q] = y;
                s1[q] += y; s2[q] = fmaf(y, y, s2[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) { sp[wave][0][lane * CPL + q] = s1[q]; sp[wave][1][lane * CPL + q] = s2[q]; }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int which = e / C, c = e - which * C;
        part[((size_t)blockIdx.x * 2 + which) * C + c] = ((sp[0][which][c] + sp[1][which][c]) + sp[2][which][c]) + sp[3][which][c];
    }
}

// feat[pillar][c] = max over the T rows of relu(bn(y)) (padded rows: bn(0)); arg[pillar][c] = winning row, -1 = a padded
// row, -2 = the max is not positive (no gradient)
template <int CPL>
__global__ __launch_bounds__(256) void k_tr_pfn_max(PfnT p, const float* __restrict__ Y, const float* __restrict__ stats,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ feat, int* __restrict__ arg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    float sc[CPL], sh[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane * CPL + q;
        const float inv = (c < C) ? stats[2 * c + 1] * gamma[c] : 0.f;
        sc[q] = inv;
        sh[q] = (c < C) ? beta[c] - stats[2 * c] * inv : 0.f;
    }
    const long total = (long)p.batch * p.max_voxels;
    for (long gp = (long)blockIdx.x * 4 + wave; gp < total; gp += (long)gridDim.x * 4) {
        const int b = (int)(gp / p.max_voxels), pid = (int)(gp - (long)b * p.max_voxels);
        if (pid >= p.npillars[b]) continue;
        const int* ps = p.pillar_start + (size_t)b * (p.max_voxels + 1);
        const int start = ps[pid];
        const int n = min(ps[pid + 1] - start, p.T);
        const long row0 = (long)p.offsets[b] + start;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane * CPL + q;
            if (c >= C) continue;
            float best = -3.0e38f;
            int bi = -2;
            for (int j = 0; j < n; ++j) {
                const float v = fmaf(Y[(size_t)(row0 + j) * C + c], sc[q], sh[q]);
                if (v > best) { best = v; bi = j; }
            }
            if (n < p.T && sh[q] > best) { best = sh[q]; bi = -1; }   // a zero-padded row: Dense(0) = 0 -> BN
            if (!(best > 0.f)) { best = 0.f; bi = -2; }
            feat[(size_t)gp * C + c] = best;
            arg[(size_t)gp * C + c] = bi;
        }
    }
}

// canvas[b][y][x][c] = sum over the z cells of the pillar features that map to (y, x) (tf.scatter_nd adds duplicates)
__global__ __launch_bounds__(256) void k_tr_scatter(const int* __restrict__ cellmap, const float* __restrict__ feat,
                                                    float* __restrict__ canvas, int batch, int nz, int ncanvas, int C,
                                                    int max_voxels) {
    const int c4n = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)batch * ncanvas * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const long cellg = i / c4n;
    const int b = (int)(cellg / ncanvas), cell = (int)(cellg - (long)b * ncanvas);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nz; ++z) {
        const int pid = cellmap[((size_t)b * nz + z) * ncanvas + cell];
        if (pid < 0) continue;
        const float4 v = *reinterpret_cast<const float4*>(feat + ((size_t)b * max_voxels + pid) * C + c);
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    }
    *reinterpret_cast<float4*>(canvas + (size_t)cellg * C + c) = o;
}

// PFN backward, pass 1: the gradient of a pillar feature goes to its winning row; sums of g and g * yhat
template <int CPL>
__global__ __launch_bounds__(256) void k_tr_pfn_bwd_reduce(PfnT p, const float* __restrict__ Y, const float* __restrict__ stats,
                                                           const int* __restrict__ arg, const float* __restrict__ dcanvas,
                                                           float* __restrict__ part) {
    __shared__ float sp[4][2][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    float s1[CPL], s2[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) s1[q] = s2[q] = 0.f;
    const long total = (long)p.batch * p.max_voxels;
    const int ncanvas = p.nx * p.ny;
    for (long gp = (long)blockIdx.x * 4 + wave; gp < total; gp += (long)gridDim.x * 4) {
        const int b = (int)(gp / p.max_voxels), pid = (int)(gp - (long)b * p.max_voxels);
        if (pid >= p.npillars[b]) continue;
        const int cell = p.pillar_cell[(size_t)b * p.max_voxels + pid] % ncanvas;    // (y, x): the z index drops out
        const long row0 = (long)p.offsets[b] + p.pillar_start[(size_t)b * (p.max_voxels + 1) + pid];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane * CPL + q;
            if (c >= C) continue;
            const int a = arg[(size_t)gp * C + c];
            if (a == -2) continue;
            const float g = dcanvas[((size_t)b * ncanvas + cell) * C + c];
            const float y = (a >= 0) ? Y[(size_t)(row0 + a) * C + c] : 0.f;
            const float yh = (y - stats[2 * c]) * stats[2 * c + 1];
            s1[q] += g; s2[q] = fmaf(g, yh, s2[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) { sp[wave][0][lane * CPL + q] = s1[q]; sp[wave][1][lane * CPL + q] = s2[q]; }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int which = e / C, c = e - which * C;
        part[((size_t)blockIdx.x * 2 + which) * C + c] = ((sp[0][which][c] + sp[1][which][c]) + sp[2][which][c]) + sp[3][which][c];
    }
}

// pass 2: dy of every real row (the mean terms of the BatchNorm gradient reach all of them), dW[f][c] partials
template <int CPL>
__global__ __launch_bounds__(256) void k_tr_pfn_bwd_apply(PfnT p, const float* __restrict__ Y, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const int* __restrict__ arg,
                                                          const float* __restrict__ dcanvas, const float* __restrict__ sums,
                                                          const float* __restrict__ n_rows_dev, float* __restrict__ part) {
    __shared__ float sp[4][10][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    float dw[10][CPL];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int q = 0; q < CPL; ++q) dw[k][q] = 0.f;
    const float n_rows = fmaxf(*n_rows_dev, 1.f);
    float mean[CPL], inv[CPL], gi[CPL], m1[CPL], m2[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane * CPL + q;
        const bool ok = c < C;
        mean[q] = ok ? stats[2 * c] : 0.f;
        inv[q] = ok ? stats[2 * c + 1] : 0.f;
        gi[q] = ok ? gamma[c] * inv[q] : 0.f;
        m1[q] = ok ? sums[c] / n_rows : 0.f;
        m2[q] = ok ? sums[C + c] / n_rows : 0.f;
    }
    const long total = (long)p.batch * p.max_voxels;
    const int ncanvas = p.nx * p.ny;
    for (long gp = (long)blockIdx.x * 4 + wave; gp < total; gp += (long)gridDim.x * 4) {
        PillarInfo pi;
        if (!pillar_info(p, gp, lane, pi)) continue;
        const int cell = p.pillar_cell[(size_t)pi.b * p.max_voxels + pi.pid] % ncanvas;
        float g[CPL];
        int a[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane * CPL + q;
            a[q] = (c < C) ? arg[(size_t)gp * C + c] : -2;
            g[q] = (c < C && a[q] >= 0) ? dcanvas[((size_t)pi.b * ncanvas + cell) * C + c] : 0.f;
        }
        for (int j = 0; j < pi.n; ++j) {
            float f[10];
            pfn_row_features<CPL>(p, p.pts_sorted + (size_t)(pi.row0 + j) * p.F, pi.mx, pi.my, pi.mz, pi.cx, pi.cy, f);
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int c = lane * CPL + q;
                if (c >= C) continue;
                const float yh = (Y[(size_t)(pi.row0 + j) * C + c] - mean[q]) * inv[q];
                const float dy = gi[q] * (((a[q] == j) ? g[q] : 0.f) - m1[q] - yh * m2[q]);
#pragma unroll
                for (int k = 0; k < 10; ++k) dw[k][q] = fmaf(f[k], dy, dw[k][q]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int q = 0; q < CPL; ++q) sp[wave][k][lane * CPL + q] = dw[k][q];
    __syncthreads();
    for (int e = threadIdx.x; e < p.FA * C; e += 256) {
        const int k = e / C, c = e - k * C;
        part[(size_t)blockIdx.x * p.FA * C + e] = ((sp[0][k][c] + sp[1][k][c]) + sp[2][k][c]) + sp[3][k][c];
    }
}

// head weights: the three 1x1 kernels [CC][n_i] + biases <-> one [CC][32] matrix + [32] bias (zero padded)
__global__ __launch_bounds__(256) void k_tr_pack_heads(const float* __restrict__ kb, const float* __restrict__ kc,
                                                       const float* __restrict__ kd, const float* __restrict__ bb,
                                                       const float* __restrict__ bc, const float* __restrict__ bd, int CC,
                                                       int nb, int nc, int nd, float* __restrict__ W, float* __restrict__ bias) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < CC * PP_HEAD_COLS) {
        const int cc = i / PP_HEAD_COLS, col = i % PP_HEAD_COLS;
        float v = 0.f;
        if (col < nb) v = kb[cc * nb + col];
        else if (col < nb + nc) v = kc[cc * nc + col - nb];
        else if (col < nb + nc + nd) v = kd[cc * nd + col - nb - nc];
        W[i] = v;
    }
    if (i < PP_HEAD_COLS) {
        float v = 0.f;
        if (i < nb) v = bb[i];
        else if (i < nb + nc) v = bc[i - nb];
        else if (i < nb + nc + nd) v = bd[i - nb - nc];
        bias[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_tr_unpack_head_grads(const float* __restrict__ dW, const float* __restrict__ dbias,
                                                              int CC, int nb, int nc, int nd, float* __restrict__ gkb,
                                                              float* __restrict__ gkc, float* __restrict__ gkd,
                                                              float* __restrict__ gbb, float* __restrict__ gbc,
                                                              float* __restrict__ gbd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < CC * PP_HEAD_COLS) {
        const int cc = i / PP_HEAD_COLS, col = i % PP_HEAD_COLS;
        if (col < nb) gkb[cc * nb + col] = dW[i];
        else if (col < nb + nc) gkc[cc * nc + col - nb] = dW[i];
        else if (col < nb + nc + nd) gkd[cc * nd + col - nb - nc] = dW[i];
    }
    if (i < PP_HEAD_COLS) {
        if (i < nb) gbb[i] = dbias[i];
        else if (i < nb + nc) gbc[i - nb] = dbias[i];
        else if (i < nb + nc + nd) gbd[i - nb - nc] = dbias[i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
std::vector<TrainEntry> train_layout(const TrainShape& s, int64_t* n_params, int64_t* n_state) {
    std::vector<TrainEntry> v;
    int64_t po = 0, so = 0;
    auto P = [&](const std::string& name, int64_t size) { v.push_back({name, po, size, 0}); po += size; };
    auto S = [&](const std::string& name, int64_t size) { v.push_back({name, so, size, 1}); so += size; };
    auto BN = [&](const std::string& pre, int c) {
        P(pre + "/gamma", c); P(pre + "/beta", c);
        S(pre + "/moving_mean", c); S(pre + "/moving_variance", c);
    };
    P("pfn/dense/kernel", (int64_t)s.FA * s.C);
    BN("pfn/bn", s.C);
    int bi = 0, li = 0;
    for (const LayerDesc& L : s.layers) {
        if (L.kind == LAYER_SEP) {
            const std::string pre = "rpn/block" + std::to_string(bi + 1) + "/" + std::to_string(li);
            P(pre + "/depthwise_kernel", (int64_t)9 * L.cin);
            P(pre + "/pointwise_kernel", (int64_t)L.cin * L.cout);
            BN(pre + "/bn", L.cout);
            ++li;
        } else if (L.kind == LAYER_DECONV) {
            const std::string pre = "rpn/deconv" + std::to_string(bi + 1);
            P(pre + "/kernel", (int64_t)L.k * L.k * L.cout * L.cin);
            BN(pre + "/bn", L.cout);
            ++bi; li = 0;
        }
    }
    const int nb = s.napl * 7, nc = s.napl * s.ncls, nd = s.use_dir ? s.napl * 2 : 0;
    P("rpn/conv_box/kernel", (int64_t)s.CC * nb); P("rpn/conv_box/bias", nb);
    P("rpn/conv_cls/kernel", (int64_t)s.CC * nc); P("rpn/conv_cls/bias", nc);
    if (nd) { P("rpn/conv_dir_cls/kernel", (int64_t)s.CC * nd); P("rpn/conv_dir_cls/bias", nd); }
    if (n_params) *n_params = po;
    if (n_state) *n_state = so;
    return v;
}

namespace {

struct Lookup {
    const std::vector<TrainEntry>& v;
    const float* params; float* grads; float* state;
    const TrainEntry& e(const std::string& n) const {
        for (const TrainEntry& t : v) if (t.name == n) return t;
        static TrainEntry none{"", 0, 0, 0};
        return none;
    }
    const float* p(const std::string& n) const { return params + e(n).offset; }
    float* g(const std::string& n) const { return grads + e(n).offset; }
    float* s(const std::string& n) const { return state + e(n).offset; }
};

unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

void col_reduce(const TrainCtx& cx, int C, float* sums) {   // TR_NPART partial rows of [2][C] -> sums[2][C]
    PP_LAUNCH("k_tr_reduce", k_tr_reduce, dim3(reduce_blocks(2 * C)), dim3(256), 0, cx.stream, (const float*)cx.part, TR_NPART,
              (long)2 * C, (long)2 * C, sums, 0L, 0, 0, 1.0f);
}

// BatchNorm (training) + ReLU over Z[rows][C] -> A (mapped rows), statistics kept in `stats`, moving stats updated
void bn_relu_forward(const TrainCtx& cx, const float* Z, long rows, int C, const float* gamma, const float* beta,
                     float* stats, float* sums, float* mmean, float* mvar, float momentum, float* A, int ld, int co_off,
                     RowMap rm) {
    PP_LAUNCH("k_tr_colstats", k_tr_colstats, dim3(TR_NPART), dim3(256), 0, cx.stream, Z, rows, C, cx.part);
    PP_LAUNCH("k_tr_bn_finalize", k_tr_bn_finalize, dim3((C + 15) / 16), dim3(256), 0, cx.stream, (const float*)cx.part, TR_NPART,
              C, (float)rows, (const float*)nullptr, momentum, 1, stats, mmean, mvar);
    (void)sums;
    PP_LAUNCH("k_tr_bn_relu", k_tr_bn_relu, dim3(blocks_for(rows * (C / 4))), dim3(256), 0, cx.stream, Z, rows, C,
              (const float*)stats, gamma, beta, A, ld, co_off, rm);
}

// backward of the same: dA (mapped rows) -> dZ[rows][C]; d gamma, d beta written to the gradient buffer
void bn_relu_backward(const TrainCtx& cx, const float* dA, int ld, int co_off, RowMap rm, const float* Z, long rows, int C,
                      const float* stats, const float* gamma, const float* beta, float* sums, float* dgamma, float* dbeta,
                      float* dZ) {
    PP_LAUNCH("k_tr_bn_bwd_reduce", k_tr_bn_bwd_reduce, dim3(TR_NPART), dim3(256), 0, cx.stream, dA, ld, co_off, rm, Z, rows,
              C, stats, gamma, beta, cx.part);
    col_reduce(cx, C, sums);
    (void)hipMemcpyAsync(dbeta, sums, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, cx.stream);
    (void)hipMemcpyAsync(dgamma, sums + C, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, cx.stream);
    PP_LAUNCH("k_tr_bn_bwd_apply", k_tr_bn_bwd_apply, dim3(blocks_for(rows * C)), dim3(256), 0, cx.stream, dA, ld, co_off, rm,
              Z, rows, C, stats, gamma, beta, (const float*)sums, (float)rows, dZ);
}

// n_rows[0] = (sum of the frames' pillar counts) * T: the rows of the reference's padded [P, T, C] tensor
__global__ void k_tr_pfn_rows(const int* __restrict__ npillars, int batch, int T, float* __restrict__ n_rows) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long tot = 0;
        for (int b = 0; b < batch; ++b) tot += npillars[b];
        n_rows[0] = (float)tot * (float)T;
    }
}

template <int CPL>
void pfn_forward(const TrainCtx& cx, const PfnT& p, const Lookup& L) {
    PP_LAUNCH("k_tr_pfn_rows", k_tr_pfn_rows, dim3(1), dim3(64), 0, cx.stream, p.npillars, p.batch, p.T, cx.pfn_nrows);
    PP_LAUNCH("k_tr_pfn_lin", (k_tr_pfn_lin<CPL>), dim3(TR_NPART), dim3(256), 0, cx.stream, p, cx.pfn_y, cx.part);
    PP_LAUNCH("k_tr_bn_finalize", k_tr_bn_finalize, dim3((p.C + 15) / 16), dim3(256), 0, cx.stream, (const float*)cx.part, TR_NPART,
              p.C, 0.f, (const float*)cx.pfn_nrows, 0.01f, 0, cx.pfn_stats, L.s("pfn/bn/moving_mean"), L.s("pfn/bn/moving_variance"));
    PP_LAUNCH("k_tr_pfn_max", (k_tr_pfn_max<CPL>), dim3(TR_NPART), dim3(256), 0, cx.stream, p, (const float*)cx.pfn_y,
              (const float*)cx.pfn_stats, L.p("pfn/bn/gamma"), L.p("pfn/bn/beta"), cx.pfn_feat, cx.pfn_arg);
}

template <int CPL>
void pfn_backward(const TrainCtx& cx, const PfnT& p, const Lookup& L, const float* dcanvas) {
    PP_LAUNCH("k_tr_pfn_bwd_reduce", (k_tr_pfn_bwd_reduce<CPL>), dim3(TR_NPART), dim3(256), 0, cx.stream, p,
              (const float*)cx.pfn_y, (const float*)cx.pfn_stats, (const int*)cx.pfn_arg, dcanvas, cx.part);
    col_reduce(cx, p.C, cx.pfn_sums);
    (void)hipMemcpyAsync(L.g("pfn/bn/beta"), cx.pfn_sums, (size_t)p.C * sizeof(float), hipMemcpyDeviceToDevice, cx.stream);
    (void)hipMemcpyAsync(L.g("pfn/bn/gamma"), cx.pfn_sums + p.C, (size_t)p.C * sizeof(float), hipMemcpyDeviceToDevice, cx.stream);
    PP_LAUNCH("k_tr_pfn_bwd_apply", (k_tr_pfn_bwd_apply<CPL>), dim3(TR_NPART), dim3(256), 0, cx.stream, p, (const float*)cx.pfn_y,
              (const float*)cx.pfn_stats, L.p("pfn/bn/gamma"), (const int*)cx.pfn_arg, dcanvas, (const float*)cx.pfn_sums,
              (const float*)cx.pfn_nrows, cx.part);
    const long n = (long)p.FA * p.C;
    PP_LAUNCH("k_tr_reduce", k_tr_reduce, dim3(reduce_blocks(n)), dim3(256), 0, cx.stream, (const float*)cx.part, TR_NPART, n, n,
              L.g("pfn/dense/kernel"), 0L, 0, 0, 1.0f);
}

}  // namespace

size_t train_part_floats(const TrainShape& s) {
    size_t m = (size_t)2 * 512;
    m = std::max(m, (size_t)10 * s.C);
    for (const LayerDesc& L : s.layers) m = std::max(m, (size_t)9 * std::max(L.cin, L.cout));
    return (size_t)TR_NPART * m;
}

int train_step(const TrainCtx& cx, const TrainShape& s, const std::vector<TrainEntry>& layout, const float* params,
               float* grads, float* state, int batch, const LossParams& loss_in) {
    Lookup L{layout, params, grads, state};
    const int B = batch;
    if (s.C > 256 || s.C % 4 != 0 || s.FA > 10) return PP_ERR_UNSUPPORTED;
    for (const LayerDesc& l : s.layers)   // channel counts the reduction kernels are written for
        if (l.kind != LAYER_HEAD && (l.cin % 16 != 0 || l.cout < 32 || l.cout > 256 || (l.cout & (l.cout - 1)) != 0 ||
                                     l.cin > 256 || (l.kind == LAYER_SEP && l.stride != 1 && l.stride != 2)))
            return PP_ERR_UNSUPPORTED;

    // ---------------- forward ----------------
    PfnT p;
    memset(&p, 0, sizeof(p));
    p.batch = B; p.nx = s.nx; p.ny = s.ny; p.C = s.C; p.F = s.F; p.FA = s.FA; p.T = s.T; p.max_voxels = s.max_voxels;
    p.with_distance = s.with_dist;
    p.vx = s.vx; p.vy = s.vy; p.x_off = s.x_off; p.y_off = s.y_off;
    p.pts_sorted = cx.pts_sorted; p.offsets = cx.offsets; p.pillar_start = cx.pillar_start; p.pillar_cell = cx.pillar_cell;
    p.npillars = cx.npillars; p.W = L.p("pfn/dense/kernel");
    const int cpl = (s.C + 63) / 64;
    if (cpl == 1) pfn_forward<1>(cx, p, L);
    else if (cpl == 2) pfn_forward<2>(cx, p, L);
    else pfn_forward<4>(cx, p, L);
    const int ncanvas = s.nx * s.ny;
    PP_LAUNCH("k_tr_scatter", k_tr_scatter, dim3(blocks_for((long)B * ncanvas * (s.C / 4))), dim3(256), 0, cx.stream, cx.cellmap,
              (const float*)cx.pfn_feat, cx.canvas, B, s.nz, ncanvas, s.C, s.max_voxels);

    const RowMap ident{1, 0, 0};
    const size_t HW = (size_t)s.head_h * s.head_w;
    const float* cur = cx.canvas;
    int bi = 0, li = 0, co_off = 0;
    for (size_t i = 0; i < s.layers.size(); ++i) {
        const LayerDesc& l = s.layers[i];
        const TrainLayerBuf& tb = cx.lbuf[i];
        if (l.kind == LAYER_SEP) {
            const std::string pre = "rpn/block" + std::to_string(bi + 1) + "/" + std::to_string(li);
            const long rows = (long)B * l.out_h * l.out_w;
            PP_LAUNCH("k_tr_dw_fwd", k_tr_dw_fwd, dim3(blocks_for(rows * (l.cin / 4))), dim3(256), 0, cx.stream, cur,
                      L.p(pre + "/depthwise_kernel"), tb.D, B, l.in_h, l.in_w, l.out_h, l.out_w, l.cin, l.stride);
            tr_gemm(cx, tb.D, l.cin, 1, L.p(pre + "/pointwise_kernel"), l.cout, 1, tb.Z, l.cout, (int)rows, l.cout, l.cin,
                    nullptr, 0, 1);
            bn_relu_forward(cx, tb.Z, rows, l.cout, L.p(pre + "/bn/gamma"), L.p(pre + "/bn/beta"), tb.stats, tb.sums,
                            L.s(pre + "/bn/moving_mean"), L.s(pre + "/bn/moving_variance"), 0.99f, tb.A, l.cout, 0, ident);
            cur = tb.A;
            ++li;
        } else if (l.kind == LAYER_DECONV) {
            const std::string pre = "rpn/deconv" + std::to_string(bi + 1);
            const long m = (long)B * l.in_h * l.in_w;
            const int N = l.k * l.k * l.cout;
            // Zs[m][tap * cout + co] = X[m][:] . K[tap][co][:]   (Keras Conv2DTranspose kernel [k, k, Cout, Cin])
            tr_gemm(cx, cur, l.cin, 1, L.p(pre + "/kernel"), 1, l.cin, tb.Z, N, (int)m, N, l.cin, nullptr, 0, 1);
            const RowMap rm{l.k, l.in_h, l.in_w};
            bn_relu_forward(cx, tb.Z, m * l.k * l.k, l.cout, L.p(pre + "/bn/gamma"), L.p(pre + "/bn/beta"), tb.stats, tb.sums,
                            L.s(pre + "/bn/moving_mean"), L.s(pre + "/bn/moving_variance"), 0.99f, cx.cat, s.CC, co_off, rm);
            co_off += l.cout;
            ++bi; li = 0;
        }
    }
    // heads: head[px][32] = cat[px][CC] . Wh[CC][32] + bias
    const int nb = s.napl * 7, nc = s.napl * s.ncls, nd = s.use_dir ? s.napl * 2 : 0;
    const float* kd = nd ? L.p("rpn/conv_dir_cls/kernel") : L.p("rpn/conv_box/kernel");
    const float* bd = nd ? L.p("rpn/conv_dir_cls/bias") : L.p("rpn/conv_box/bias");
    PP_LAUNCH("k_tr_pack_heads", k_tr_pack_heads, dim3(blocks_for((long)s.CC * PP_HEAD_COLS)), dim3(256), 0, cx.stream,
              L.p("rpn/conv_box/kernel"), L.p("rpn/conv_cls/kernel"), kd, L.p("rpn/conv_box/bias"), L.p("rpn/conv_cls/bias"), bd,
              s.CC, nb, nc, nd, cx.head_w, cx.head_b);
    const long px = (long)B * HW;
    tr_gemm(cx, cx.cat, s.CC, 1, cx.head_w, PP_HEAD_COLS, 1, cx.head, PP_HEAD_COLS, (int)px, PP_HEAD_COLS, s.CC, cx.head_b, 0, 1);

    // ---------------- loss + gradient at the head maps ----------------
    LossParams lp = loss_in;
    lp.head = cx.head;
    lp.head_grad = cx.dhead;
    int st = launch_head_loss(lp, cx.stream);
    if (st) return st;

    // ---------------- backward ----------------
    // heads: dWh = cat^T . dhead, dbias = column sums, dcat = dhead . Wh^T
    tr_gemm(cx, cx.cat, 1, s.CC, cx.dhead, PP_HEAD_COLS, 1, cx.dhead_w, PP_HEAD_COLS, s.CC, PP_HEAD_COLS, (int)px, nullptr, 0,
            wgrad_split(cx, s.CC, PP_HEAD_COLS, (int)px));
    PP_LAUNCH("k_tr_colstats", k_tr_colstats, dim3(TR_NPART), dim3(256), 0, cx.stream, (const float*)cx.dhead, px, PP_HEAD_COLS,
              cx.part);
    col_reduce(cx, PP_HEAD_COLS, cx.dhead_b);   // [0][c] = column sums (the [1][c] half is unused)
    {
        float* gkd = nd ? L.g("rpn/conv_dir_cls/kernel") : L.g("rpn/conv_box/kernel");
        float* gbd = nd ? L.g("rpn/conv_dir_cls/bias") : L.g("rpn/conv_box/bias");
        PP_LAUNCH("k_tr_unpack_head_grads", k_tr_unpack_head_grads, dim3(blocks_for((long)s.CC * PP_HEAD_COLS)), dim3(256), 0,
                  cx.stream, (const float*)cx.dhead_w, (const float*)cx.dhead_b, s.CC, nb, nc, nd, L.g("rpn/conv_box/kernel"),
                  L.g("rpn/conv_cls/kernel"), gkd, L.g("rpn/conv_box/bias"), L.g("rpn/conv_cls/bias"), gbd);
    }
    tr_gemm(cx, cx.dhead, PP_HEAD_COLS, 1, cx.head_w, 1, PP_HEAD_COLS, cx.dcat, s.CC, (int)px, s.CC, PP_HEAD_COLS, nullptr, 0, 1);

    // blocks in reverse: the gradient of a block's output arrives from the next block's first layer (stored by
    // its depthwise backward) and from its own transposed convolution (accumulated on top)
    std::vector<int> first_of_block, deconv_of_block;
    {
        int start = 0;
        for (size_t i = 0; i < s.layers.size(); ++i)
            if (s.layers[i].kind == LAYER_DECONV) { first_of_block.push_back(start); deconv_of_block.push_back((int)i); start = (int)i + 1; }
    }
    const int nblocks = (int)deconv_of_block.size();
    std::vector<int> cat_off(nblocks, 0);
    for (int b = 1; b < nblocks; ++b) cat_off[b] = cat_off[b - 1] + s.layers[deconv_of_block[b - 1]].cout;
    for (int b = nblocks - 1; b >= 0; --b) {
        const int di = deconv_of_block[b];
        const LayerDesc& d = s.layers[di];
        const TrainLayerBuf& db = cx.lbuf[di];
        const std::string dpre = "rpn/deconv" + std::to_string(b + 1);
        const int last = di - 1;                                  // last separable layer of the block
        const float* Xd = cx.lbuf[last].A;                        // the block's output = the deconv's input
        const long m = (long)B * d.in_h * d.in_w;
        const int N = d.k * d.k * d.cout;
        const RowMap rm{d.k, d.in_h, d.in_w};
        bn_relu_backward(cx, cx.dcat, s.CC, cat_off[b], rm, db.Z, m * d.k * d.k, d.cout, db.stats, L.p(dpre + "/bn/gamma"),
                         L.p(dpre + "/bn/beta"), db.sums, L.g(dpre + "/bn/gamma"), L.g(dpre + "/bn/beta"), cx.dZ);
        // dK[n][cin] = dZs^T . X     dX[m][cin] (+)= dZs . K
        tr_gemm(cx, cx.dZ, 1, N, Xd, d.cin, 1, L.g(dpre + "/kernel"), d.cin, N, d.cin, (int)m, nullptr, 0,
                wgrad_split(cx, N, d.cin, (int)m));
        float* dAct = cx.lbuf[last].dA;
        tr_gemm(cx, cx.dZ, N, 1, L.p(dpre + "/kernel"), d.cin, 1, dAct, d.cin, (int)m, d.cin, N, nullptr,
                (b + 1 < nblocks) ? 1 : 0, 1);
        for (int i = last; i >= first_of_block[b]; --i) {
            const LayerDesc& l = s.layers[i];
            const TrainLayerBuf& tb = cx.lbuf[i];
            const std::string pre = "rpn/block" + std::to_string(b + 1) + "/" + std::to_string(i - first_of_block[b]);
            const long rows = (long)B * l.out_h * l.out_w;
            const float* X = (i == 0) ? cx.canvas : cx.lbuf[i - 1 - ((i == first_of_block[b] && b > 0) ? 1 : 0)].A;
            bn_relu_backward(cx, tb.dA, l.cout, 0, ident, tb.Z, rows, l.cout, tb.stats, L.p(pre + "/bn/gamma"),
                             L.p(pre + "/bn/beta"), tb.sums, L.g(pre + "/bn/gamma"), L.g(pre + "/bn/beta"), cx.dZ);
            // dWp[cin][cout] = D^T . dZ      dD[rows][cin] = dZ . Wp^T
            tr_gemm(cx, tb.D, 1, l.cin, cx.dZ, l.cout, 1, L.g(pre + "/pointwise_kernel"), l.cout, l.cin, l.cout, (int)rows,
                    nullptr, 0, wgrad_split(cx, l.cin, l.cout, (int)rows));
            tr_gemm(cx, cx.dZ, l.cout, 1, L.p(pre + "/pointwise_kernel"), 1, l.cout, cx.dD, l.cin, (int)rows, l.cin, l.cout,
                    nullptr, 0, 1);
            PP_LAUNCH("k_tr_dw_bwd_w", k_tr_dw_bwd_w, dim3(TR_NPART), dim3(256), 0, cx.stream, X, (const float*)cx.dD, cx.part, B,
                      l.in_h, l.in_w, l.out_h, l.out_w, l.cin, l.stride);
            {
                const long n = (long)9 * l.cin;
                PP_LAUNCH("k_tr_reduce", k_tr_reduce, dim3(reduce_blocks(n)), dim3(256), 0, cx.stream, (const float*)cx.part, TR_NPART,
                          n, n, L.g(pre + "/depthwise_kernel"), 0L, 0, 0, 1.0f);
            }
            // gradient of this layer's input: the previous layer's dA, the previous block's output gradient, or the canvas
            float* dX = (i == 0) ? cx.dcanvas : cx.lbuf[i - 1 - ((i == first_of_block[b] && b > 0) ? 1 : 0)].dA;
            PP_LAUNCH("k_tr_dw_bwd_in", k_tr_dw_bwd_in, dim3(blocks_for((long)B * l.in_h * l.in_w * (l.cin / 4))), dim3(256), 0,
                      cx.stream, (const float*)cx.dD, L.p(pre + "/depthwise_kernel"), dX, B, l.in_h, l.in_w, l.out_h, l.out_w,
                      l.cin, l.stride, 0);
        }
    }
    // canvas -> pillar features -> PFN
    if (cpl == 1) pfn_backward<1>(cx, p, L, cx.dcanvas);
    else if (cpl == 2) pfn_backward<2>(cx, p, L, cx.dcanvas);
    else pfn_backward<4>(cx, p, L, cx.dcanvas);
    return PP_OK;
}
