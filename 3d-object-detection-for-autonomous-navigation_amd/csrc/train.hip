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
//     run as products on the 16-bit matrix pipe with float32-equivalent results (k_tr_gemm2: three bfloat16 pieces
//     per value, six piece products; k_tr_gemm on v_mfma_f32_32x32x2_f32 is the fallback for odd shapes), forward
//     NN / NT, input gradients with the other operand transposed, weight gradients as split-K TN products (K =
//     pixels) reduced in a fixed order; the two gradient products of a layer share a launch (k_tr_gemm2_pair);
//   * the pre-BatchNorm map Z of a layer is what is kept: the activation of an in-block layer is never written --
//     its readers (next depthwise, the backward kernels) evaluate relu(z * sc + sh) from Z and a per-channel
//     coefficient table (k_tr_bn_finalize); the depthwise backward is one pass (k_tr_dw_bwd: input gradient, kernel
//     gradient, and the BatchNorm-backward sums of the layer before); the PFN kernels keep a pillar's points in the
//     lanes and recompute the ten-FMA Dense row instead of storing it;
//   * every reduction goes through per-workgroup partial sums added in a fixed order, so a step is bit-reproducible
//     (no floating-point atomics); persistent reduction grids never exceed one resident round of workgroups.
// BatchNorm in training mode normalises with the batch mean and the biased batch variance; the moving
// statistics are updated as Keras does (moving = moving * momentum + batch * (1 - momentum); the RPN's fused
// BatchNorm feeds the unbiased variance, the PFN's rank-3 BatchNorm the biased one).  The PFN statistics run
// over ALL P * T rows of the reference's padded tensor: the zero rows contribute nothing to the sums but count
// in N, and a padded row that wins the max receives the gradient (it only reaches beta / gamma / the statistics).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "pp_common.h"
#include "train.h"

typedef float tf32x16 __attribute__((ext_vector_type(16)));

#define TR_EPS 1e-3f
// workgroups of the persistent reduction kernels (= rows of their partial-sum buffers): one per CU for the reference's
// 2-frame batches (every launch is a few microseconds: more workgroups only add dispatch time), up to TR_NPART_MAX
// for large per-GPU batches, where 256 workgroups leave the chip a quarter full.  Chosen per step by train_step.
#define TR_NPART_MAX 2048
static thread_local int g_tr_npart = 256;   // (per-step scratch state is thread-local: distinct handles may step on distinct threads)
#define TR_NPART g_tr_npart

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

// ------------------------------------------------------------------------------------------------------------
// Split-precision GEMM (round 3): the same C = A . B contract as k_tr_gemm on the 16-bit matrix pipe.  Every
// float32 operand value is carried as THREE bfloat16 pieces (hi + mid + lo: 24 significand bits, float32's exponent
// range -- gradients of 1e-8 and activations of 1e3 alike) and a product as the six piece products of weight
// >= 2^-16, accumulated in float32: float32-equivalent results (the inference kernels' PP_SPLIT_MODE 0 arithmetic)
// at 6 x 32 matrix-pipe cycles per 32x32x16 block instead of 8 x 64 with v_mfma_f32_32x32x2_f32.
//   * a 4-wave workgroup owns a (64*WM) x (64*WN) tile, wave (wm, wn) a (32*WM) x (32*WN) part of it; K in chunks
//     of 32 through LDS, the next chunk's global loads (16 bytes per lane) in flight while the current one is
//     multiplied;
//   * the split happens ONCE per loaded element, on the way into LDS (three planes of [rows][32 + 8 pad] bfloat16:
//     80-byte rows make the 16-byte fragment reads conflict-free); an operand whose unit-stride axis is not K
//     (TN / NT forms: weight gradients, transposed weights) is transposed in registers on that way;
//   * epilogue: bias, accumulate into C, split-K partial tiles (reduced by k_tr_reduce in a fixed order), and --
//     for the forward products that feed a BatchNorm -- the per-column sum and sum of squares of the workgroup's
//     rows (statistics partials [row tile][2][N]: k_tr_colstats' pass over Z is gone).
// AKC: A(m, k) has unit stride along k (sak == 1), else along m (sam == 1);  BKC: B(k, n) along k, else along n.
// Needs K % 4 == 0, 16-byte aligned rows (leading dimensions % 4 == 0); the launcher falls back to k_tr_gemm.
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tbf16x2 __attribute__((ext_vector_type(2)));
typedef float tf32x2 __attribute__((ext_vector_type(2)));
// bfloat16 per LDS row: the KCH of a chunk + 8 pad (80-byte rows for KCH = 32, 144-byte rows for 64: the 16-byte
// fragment reads of 16 consecutive rows fall on 16 distinct bank quads either way)
#define T2_LDR(KCH) ((KCH) + 8)

// v[0..1] -> packed bfloat16 pairs of the three pieces
__device__ __forceinline__ void t2_split2(float v0, float v1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const tf32x2 x = {v0, v1};
    const tbf16x2 h = __builtin_convertvector(x, tbf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const tf32x2 r1 = {v0 - __uint_as_float(hi << 16), v1 - __uint_as_float(hi & 0xffff0000u)};
    const tbf16x2 m = __builtin_convertvector(r1, tbf16x2);
    mid = __builtin_bit_cast(unsigned, m);
    const tf32x2 r2 = {r1.x - __uint_as_float(mid << 16), r1.y - __uint_as_float(mid & 0xffff0000u)};
    const tbf16x2 l = __builtin_convertvector(r2, tbf16x2);
    lo = __builtin_bit_cast(unsigned, l);
}

// One operand tile of ROWS (m or n) x KCH (k) per chunk, 256 threads.  KC form: a thread owns EPT consecutive k of one
// row; transposed form: 4 consecutive rows x EPT / 4 consecutive k.
template <int ROWS, bool KC, int KCH>
struct T2Stage {
    static constexpr int EPT = ROWS * KCH / 256;           // elements per thread per chunk: 8, 16 or 32
    static constexpr int LDR = T2_LDR(KCH);
    static_assert(KC || EPT == 8 || EPT == 16, "transposed staging: 2 or 4 k per thread");
    float v[EPT];
    // Transposed form, thread -> (group of 4 rows, group of KPT k): lane bits [1:0] walk the rows (four lanes = 64
    // contiguous bytes of one k line of the operand), the next bits the k groups, the rest (the waves) the remaining row
    // groups.  The LDS image is [row][k] with 80- / 144-byte rows, so the bank of a lane's 4-byte store is
    // 16 * (row group & 1) + k group + const (mod 32): with the row groups in the LOW lane bits and 16 of them per wave
    // (the first version: row group = tid % 16) a 32-lane store group hit two banks per k group -- 8-way conflicts on
    // every one of the 12 stores a thread issues per operand and chunk (SQ_LDS_BANK_CONFLICT 78 % of the LDS-active
    // cycles of the weight-gradient products, 66 % of the forward ones); with 8 k groups x 2 row-group parities per
    // store group it is 2-way, which a ds_write_b32 absorbs.
    static __device__ __forceinline__ int kgi(int tid) { return (tid >> 2) % (KCH * 4 / EPT); }
    static __device__ __forceinline__ int rgi(int tid) { return (tid & 3) + 4 * (tid / (KCH * 16 / EPT)); }
    // p(r, k) = base[r * sr + k * sk]; rows >= nrows and k >= kend read as zero
    __device__ __forceinline__ void load(const float* __restrict__ base, long sr, long sk, int row0, int nrows, int k0, int kend, int tid) {
        if (KC) {
            constexpr int TPR = KCH / EPT;
            const int r = row0 + tid / TPR, k = k0 + (tid % TPR) * EPT;
            const float* q = base + (long)r * sr + k;
#pragma unroll
            for (int j = 0; j < EPT; j += 4) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrows && k + j < kend) t = *reinterpret_cast<const float4*>(q + j);     // K % 4 == 0
                v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
            }
        } else {
            constexpr int KPT = EPT / 4;                   // k per thread (KCH / KPT k groups x ROWS / 4 row groups = 256 threads)
            const int r = row0 + rgi(tid) * 4, k = k0 + kgi(tid) * KPT;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrows && k + j < kend) t = *reinterpret_cast<const float4*>(base + (long)(k + j) * sk + r);   // rows % 4 == 0
                v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;   // v[4j + i] = p(r + i, k + j)
            }
        }
    }
    // split and store into the three LDS planes plane[p][row][LDR]
    __device__ __forceinline__ void store(__bf16* __restrict__ lds, int tid) const {
        constexpr int PL = ROWS * LDR;
        if (KC) {
            constexpr int TPR = KCH / EPT;
            __bf16* d = lds + (tid / TPR) * LDR + (tid % TPR) * EPT;
#pragma unroll
            for (int j = 0; j < EPT; j += 8) {
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t2_split2(v[j + 2 * i], v[j + 2 * i + 1], h[i], m[i], l[i]);
                *reinterpret_cast<uint4*>(d + j) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4*>(d + PL + j) = make_uint4(m[0], m[1], m[2], m[3]);
                *reinterpret_cast<uint4*>(d + 2 * PL + j) = make_uint4(l[0], l[1], l[2], l[3]);
            }
        } else {
            constexpr int KPT = EPT / 4;
            __bf16* d = lds + (rgi(tid) * 4) * LDR + kgi(tid) * KPT;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                  // row r + i: KPT consecutive k
                unsigned h[KPT / 2], m[KPT / 2], l[KPT / 2];
#pragma unroll
                for (int j = 0; j < KPT; j += 2) t2_split2(v[4 * j + i], v[4 * (j + 1) + i], h[j / 2], m[j / 2], l[j / 2]);
                if (KPT == 4) {
                    *reinterpret_cast<uint2*>(d + i * LDR) = make_uint2(h[0], h[KPT / 2 - 1]);
                    *reinterpret_cast<uint2*>(d + PL + i * LDR) = make_uint2(m[0], m[KPT / 2 - 1]);
                    *reinterpret_cast<uint2*>(d + 2 * PL + i * LDR) = make_uint2(l[0], l[KPT / 2 - 1]);
                } else {
                    *reinterpret_cast<unsigned*>(d + i * LDR) = h[0];
                    *reinterpret_cast<unsigned*>(d + PL + i * LDR) = m[0];
                    *reinterpret_cast<unsigned*>(d + 2 * PL + i * LDR) = l[0];
                }
            }
        }
    }
};

struct TGemm2 {
    TGemm g;
    float* stat_part;    // [grid.y][2][N] column sums / sums of squares of the tile's rows, or NULL
    int gx, gy, gz;      // the product's workgroup grid (a launch may carry two products: k_tr_gemm2_pair)
};

// one workgroup of one product: tile (bx, by), K slice bz; sA / sB: 3 * BM * LDR and 3 * BN * LDR bfloat16
template <int WM, int WN, bool AKC, bool BKC, int KCH>
__device__ __forceinline__ void t2_gemm_tile(const TGemm2& a2, int bx, int by, int bz, __bf16* __restrict__ sA,
                                             __bf16* __restrict__ sB) {
    const TGemm& g = a2.g;
    constexpr int BM = 64 * WM, BN = 64 * WN, LDR = T2_LDR(KCH);
    constexpr int PLA = BM * LDR, PLB = BN * LDR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, r32 = lane & 31;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.kper, kend = min(g.K, kbeg + g.kper);
    T2Stage<BM, AKC, KCH> ra;
    T2Stage<BN, BKC, KCH> rb;
    // A(m, k): row stride sam / k stride sak;  B(k, n) seen as rows n: row stride sbn, k stride sbk
    tf32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (kbeg < kend) {
        ra.load(g.A, g.sam, g.sak, m0, g.M, kbeg, kend, tid);
        rb.load(g.B, g.sbn, g.sbk, n0, g.N, kbeg, kend, tid);
    }
    const __bf16* fa = sA + (wm * 32 * WM + r32) * LDR + 8 * h;
    const __bf16* fb = sB + (wn * 32 * WN + r32) * LDR + 8 * h;
    for (int k0 = kbeg; k0 < kend; k0 += KCH) {
        __syncthreads();                                   // the previous chunk's fragments have been read
        ra.store(sA, tid);
        rb.store(sB, tid);
        __syncthreads();
        if (k0 + KCH < kend) {
            ra.load(g.A, g.sam, g.sak, m0, g.M, k0 + KCH, kend, tid);
            rb.load(g.B, g.sbn, g.sbk, n0, g.N, k0 + KCH, kend, tid);
        }
#pragma unroll
        for (int ks = 0; ks < KCH / 16; ++ks) {
            tbf16x8 ah[WM], am[WM], al[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const __bf16* q = fa + i * 32 * LDR + ks * 16;
                ah[i] = *reinterpret_cast<const tbf16x8*>(q);
                am[i] = *reinterpret_cast<const tbf16x8*>(q + PLA);
                al[i] = *reinterpret_cast<const tbf16x8*>(q + 2 * PLA);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const __bf16* q = fb + j * 32 * LDR + ks * 16;
                const tbf16x8 bh = *reinterpret_cast<const tbf16x8*>(q);
                const tbf16x8 bm = *reinterpret_cast<const tbf16x8*>(q + PLB);
                const tbf16x8 bl = *reinterpret_cast<const tbf16x8*>(q + 2 * PLB);
#pragma unroll
                for (int i = 0; i < WM; ++i) {             // smallest terms first
                    tf32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, c, 0, 0, 0);
                    acc[i][j] = c;
                }
            }
        }
    }
    // D[row][col]: col = lane & 31 (n), row = (reg & 3) + 8 * (reg >> 2) + 4 * h (m)
    if (a2.stat_part != nullptr) {      // column sums of this workgroup's rows (rows >= M are zero rows of A)
        __syncthreads();
        float* red = reinterpret_cast<float*>(sA);         // [2 row-waves][2][BN]
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float z = acc[i][j][r]; s1 += z; s2 = fmaf(z, z, s2); }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (h == 0) {
                const int col = wn * 32 * WN + j * 32 + r32;
                red[(wm * 2 + 0) * BN + col] = s1;
                red[(wm * 2 + 1) * BN + col] = s2;
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BN; e += 256) {
            const int which = e / BN, col = e % BN;
            if (n0 + col < g.N)
                a2.stat_part[((size_t)by * 2 + which) * g.N + n0 + col] = red[which * BN + col] + red[(2 + which) * BN + col];
        }
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn * 32 * WN + j * 32 + r32;
        if (n >= g.N) continue;
        const float bv = (g.bias != nullptr) ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                if (a2.gz > 1) {
                    g.Cpart[((size_t)bz * g.M + m) * g.N + n] = acc[i][j][r];
                } else {
                    float* c = g.C + (size_t)m * g.ldc + n;
                    const float v = acc[i][j][r] + bv;
                    *c = g.accumulate ? (*c + v) : v;
                }
            }
    }
}

template <int WM, int WN, bool AKC, bool BKC, int KCH>
__global__ __launch_bounds__(256, 2) void k_tr_gemm2(TGemm2 a2) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[3 * 64 * WM * T2_LDR(KCH)];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3 * 64 * WN * T2_LDR(KCH)];
    t2_gemm_tile<WM, WN, AKC, BKC, KCH>(a2, blockIdx.x, blockIdx.y, blockIdx.z, sA, sB);
}

// Two independent products in ONE launch (the weight- and the input-gradient product of a layer read the same dZ and
// depend on nothing else: at the reference's 2-frame batch each is a handful of workgroups living on memory latency,
// and a launch of its own costs as much as the product).  64 x 64 tiles, 64-wide chunks; workgroups [0, n1) run
// product 1 (TN: A and B with unit stride across k), the rest product 2 (A along k; B2KC: B along k).
template <bool B2KC, int KCH>
__global__ __launch_bounds__(256, 2) void k_tr_gemm2_pair(TGemm2 p1, TGemm2 p2) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[3 * 64 * T2_LDR(KCH)];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3 * 64 * T2_LDR(KCH)];
    const int n1 = p1.gx * p1.gy * p1.gz;
    int id = (int)blockIdx.x;
    if (id < n1) {
        const int bx = id % p1.gx; id /= p1.gx;
        t2_gemm_tile<1, 1, false, false, KCH>(p1, bx, id % p1.gy, id / p1.gy, sA, sB);
    } else {
        id -= n1;
        const int bx = id % p2.gx; id /= p2.gx;
        t2_gemm_tile<1, 1, true, B2KC, KCH>(p2, bx, id % p2.gy, id / p2.gy, sA, sB);
    }
}

// out[i] (+)= scale * sum_p part[p][i] in a fixed order (deterministic), two shapes:
//   k_tr_reduce       many outputs (split-K partial tiles): a workgroup owns 64 consecutive outputs and cuts the
//                     partial rows into 4 interleaved slices (thread = (slice, output): a wave reads 256 contiguous
//                     bytes per row); a thread keeps 8 loads in flight and adds them in row order; the 4 slice sums
//                     are added in order through LDS;
//   k_tr_reduce_cols  few outputs, many partial rows (column sums of the persistent reductions): a whole wave per
//                     output, lane l adds rows l, l + 64, ..., then a shuffle tree (the same tree every time).
__device__ __forceinline__ void reduce_store(float s, long i, float* __restrict__ out, long ldo, int ncols, int accumulate,
                                             float scale) {
    s *= scale;
    // optional re-striding of the output ([rows][ncols] with leading dimension ldo)
    float* q = (ncols > 0) ? out + (i / ncols) * ldo + (i % ncols) : out + i;
    *q = accumulate ? (*q + s) : s;
}

__global__ __launch_bounds__(256) void k_tr_reduce(const float* __restrict__ part, int nparts, long n, long pstride,
                                                   float* __restrict__ out, long ldo, int ncols, int accumulate, float scale) {
    __shared__ float ssum[4][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < n) {
        const float* q = part + i;
        int p = sl;
        for (; p + 28 < nparts; p += 32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = q[(size_t)(p + 4 * k) * pstride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; p < nparts; p += 4) s += q[(size_t)p * pstride];
    }
    ssum[sl][o] = s;
    __syncthreads();
    if (sl != 0 || i >= n) return;
    s = ((s + ssum[1][o]) + ssum[2][o]) + ssum[3][o];
    reduce_store(s, i, out, ldo, ncols, accumulate, scale);
}

// dup0 / dup1: a second copy of the sums, outputs [0, dup_split) to dup0 and the rest to dup1 (the BatchNorm backward
// sums ARE the beta / gamma gradients: written where the optimizer reads them, no copy nodes in the graph)
__global__ __launch_bounds__(256) void k_tr_reduce_cols(const float* __restrict__ part, int nparts, long n, long pstride,
                                                        float* __restrict__ out, long ldo, int ncols, int accumulate, float scale,
                                                        float* __restrict__ dup0, float* __restrict__ dup1, int dup_split) {
    const int l = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float s = 0.f;
    if (i < n) {
        int p = l;
        for (; p + 64 * 7 < nparts; p += 64 * 8) {          // eight rows' loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(p + 64 * k) * pstride + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (p + 64 * k < nparts) ? part[(size_t)(p + 64 * k) * pstride + i] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (i >= n || l != 0) return;
    reduce_store(s, i, out, ldo, ncols, accumulate, scale);
    if (dup0 != nullptr) {
        if (i < dup_split) dup0[i] = s * scale;
        else dup1[i - dup_split] = s * scale;
    }
}

// ... and a whole workgroup per output when there are very many partial rows (large per-GPU batches)
__global__ __launch_bounds__(256) void k_tr_reduce_cols_wg(const float* __restrict__ part, int nparts, long n, long pstride,
                                                           float* __restrict__ out, long ldo, int ncols, int accumulate,
                                                           float scale, float* __restrict__ dup0, float* __restrict__ dup1,
                                                           int dup_split) {
    __shared__ float sw[4];
    const long i = blockIdx.x;
    float s = 0.f;
    {
        int p = threadIdx.x;
        for (; p + 256 * 3 < nparts; p += 256 * 4) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = part[(size_t)(p + 256 * k) * pstride + i];
#pragma unroll
            for (int k = 0; k < 4; ++k) s += v[k];
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (p + 256 * k < nparts) ? part[(size_t)(p + 256 * k) * pstride + i] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += v[k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = ((sw[0] + sw[1]) + sw[2]) + sw[3];
    reduce_store(s, i, out, ldo, ncols, accumulate, scale);
    if (dup0 != nullptr) {
        if (i < dup_split) dup0[i] = s * scale;
        else dup1[i - dup_split] = s * scale;
    }
}

// one reduction launch: the shape follows the number of outputs
static void tr_reduce(hipStream_t st, const float* part, int nparts, long n, long pstride, float* out, long ldo, int ncols,
                      int accumulate, float scale) {
    if (n <= 4096 && nparts >= 64)
        PP_LAUNCH("k_tr_reduce", k_tr_reduce_cols, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, part, nparts, n, pstride, out,
                  ldo, ncols, accumulate, scale, (float*)nullptr, (float*)nullptr, 0);
    else
        PP_LAUNCH("k_tr_reduce", k_tr_reduce, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, part, nparts, n, pstride, out, ldo,
                  ncols, accumulate, scale);
}


// Deferred reductions.  The split-K partial tiles of the weight-gradient products and the partial rows of the depthwise
// kernel gradients are needed by nobody before the optimizer: each gets a region of its own in the partial arena
// (cx.gemm_part) and ONE launch at the end of the step adds them all -- at the reference's 2-frame batch the ~35
// reduction launches they replaced were ~0.2 ms of a 2 ms step.
struct ReduceJob {
    const float* part; float* out;
    long n, pstride, ldo;
    int nparts, ncols, accumulate;
    float scale;
};
#define TR_MULTI_MAX 40
struct TMulti {
    ReduceJob job[TR_MULTI_MAX];
    int block_start[TR_MULTI_MAX + 1];
    int njobs;
};
static thread_local std::vector<ReduceJob> g_jobs;     // this step's deferred reductions (train_step resets it)
static thread_local long g_arena_used = 0;             // floats of cx.gemm_part handed out to them

__global__ __launch_bounds__(256) void k_tr_reduce_multi(TMulti m) {
    __shared__ float ssum[4][64];
    int j = 0;
    while (j + 1 < m.njobs && (int)blockIdx.x >= m.block_start[j + 1]) ++j;     // uniform
    const ReduceJob& J = m.job[j];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i = (long)((int)blockIdx.x - m.block_start[j]) * 64 + o;
    float s = 0.f;
    if (i < J.n) {
        const float* q = J.part + i;
        int p = sl;
        for (; p + 28 < J.nparts; p += 32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = q[(size_t)(p + 4 * k) * J.pstride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; p < J.nparts; p += 4) s += q[(size_t)p * J.pstride];
    }
    ssum[sl][o] = s;
    __syncthreads();
    if (sl != 0 || i >= J.n) return;
    s = ((s + ssum[1][o]) + ssum[2][o]) + ssum[3][o];
    reduce_store(s, i, J.out, J.ldo, J.ncols, J.accumulate, J.scale);
}

static void flush_deferred(const TrainCtx& cx) {
    size_t done = 0;
    while (done < g_jobs.size()) {
        TMulti m;
        m.njobs = 0;
        int blocks = 0;
        while (done < g_jobs.size() && m.njobs < TR_MULTI_MAX) {
            m.job[m.njobs] = g_jobs[done++];
            m.block_start[m.njobs] = blocks;
            blocks += (int)((m.job[m.njobs].n + 63) / 64);
            ++m.njobs;
        }
        m.block_start[m.njobs] = blocks;
        for (int k = m.njobs + 1; k <= TR_MULTI_MAX; ++k) m.block_start[k] = blocks;
        PP_LAUNCH("k_tr_reduce_multi", k_tr_reduce_multi, dim3((unsigned)blocks), dim3(256), 0, cx.stream, m);
    }
    g_jobs.clear();
}

// PP_TRAIN_GEMM=f32 keeps every product on the float32 matrix instruction (k_tr_gemm)
static bool train_split_gemm() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PP_TRAIN_GEMM"); v = (e && e[0] == 'f') ? 0 : 1; }
    return v == 1;
}

// profiler name of the next product launches ("k_tr_gemm2:<what>"; the profiler keeps the pointer: interned strings)
static thread_local const char* g_gemm_tag = "k_tr_gemm2";
static void gemm_tag(const std::string& what) {
    static thread_local std::vector<std::string*> pool;
    for (std::string* p : pool) if (*p == what) { g_gemm_tag = p->c_str(); return; }
    pool.push_back(new std::string(what));
    g_gemm_tag = pool.back()->c_str();
}

template <int WM, int WN, int KCH>
static void launch_gemm2(const TGemm2& a2, bool akc, bool bkc, dim3 grid, hipStream_t s) {
    if (akc && bkc) PP_LAUNCH(g_gemm_tag, (k_tr_gemm2<WM, WN, true, true, KCH>), grid, dim3(256), 0, s, a2);
    else if (akc) PP_LAUNCH(g_gemm_tag, (k_tr_gemm2<WM, WN, true, false, KCH>), grid, dim3(256), 0, s, a2);
    else if (bkc) PP_LAUNCH(g_gemm_tag, (k_tr_gemm2<WM, WN, false, true, KCH>), grid, dim3(256), 0, s, a2);
    else PP_LAUNCH(g_gemm_tag, (k_tr_gemm2<WM, WN, false, false, KCH>), grid, dim3(256), 0, s, a2);
}

// rows of the statistics partials a forward product leaves ([tiles][2][N]; 0: the split kernel did not run)
static thread_local int g_last_stat_tiles = 0;
static const int g_num_cus_train = 256;      // (MI355X; only a threshold for the tile choice below)

// a launch with fewer workgroups than this lives on memory latency, not on throughput: 64-wide K chunks (half the
// dependent load -> LDS -> MFMA rounds), and the two gradient products of a layer share one launch
static long tr_latency_wgs() {      // PP_TRAIN_LATENCY_WGS (A/B measurements)
    static long v = -1;
    if (v < 0) { const char* e = getenv("PP_TRAIN_LATENCY_WGS"); v = e ? atol(e) : 512; }
    return v;
}
#define TR_LATENCY_WGS tr_latency_wgs()
// PP_TRAIN_WIDE=0: 32-wide chunks everywhere (A/B measurements)
static bool wide_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PP_TRAIN_WIDE"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}

// C[M][N] (+)= A(m,k) * B(k,n) [+ bias(n)],  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// defer: a split-K product whose result only the optimizer reads (weight gradients) keeps its partial tiles in a
// region of its own and is reduced by the step's one deferred-reduction launch
struct GemmCall {
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    float* C; long ldc;
    int M, N, K;
    const float* bias;
    int accumulate, ksplit;
    float* stat_part;
    bool defer;
};

static bool gemm2_eligible(const GemmCall& c) {
    // the split-precision kernel: unit stride on one axis of each operand, 16-byte aligned rows
    const bool akc = c.sak == 1, bkc = c.sbk == 1;
    const bool a_ok = akc ? (c.sam % 4 == 0) : (c.sam == 1 && c.sak % 4 == 0 && c.M % 4 == 0);
    const bool b_ok = bkc ? (c.sbn % 4 == 0) : (c.sbn == 1 && c.sbk % 4 == 0 && c.N % 4 == 0);
    return train_split_gemm() && a_ok && b_ok && c.K % 4 == 0 && ((uintptr_t)c.A % 16 == 0) && ((uintptr_t)c.B % 16 == 0);
}

static TGemm gemm_args(const TrainCtx& cx, const GemmCall& c, int& ksplit) {
    TGemm g;
    g.A = c.A; g.sam = c.sam; g.sak = c.sak; g.B = c.B; g.sbk = c.sbk; g.sbn = c.sbn; g.C = c.C; g.ldc = c.ldc;
    g.M = c.M; g.N = c.N; g.K = c.K; g.bias = c.bias; g.accumulate = c.accumulate;
    ksplit = std::max(1, c.ksplit);
    // partial tiles go to the free tail of the arena (behind the regions of this step's deferred reductions)
    const long avail = cx.gemm_part_floats - g_arena_used;
    while (ksplit > 1 && (long)ksplit * c.M * c.N > avail) --ksplit;
    g.Cpart = cx.gemm_part + g_arena_used;
    g.kper = c.K;
    return g;
}

static void gemm_finish_split(const TrainCtx& cx, const GemmCall& c, const TGemm& g, int ks) {
    const long n = (long)c.M * c.N;
    if (c.defer) {
        g_jobs.push_back(ReduceJob{g.Cpart, c.C, n, n, c.ldc, ks, c.N, c.accumulate, 1.0f});
        g_arena_used += ((long)ks * n + 63) / 64 * 64;
    } else {
        tr_reduce(cx.stream, (const float*)g.Cpart, ks, n, n, c.C, c.ldc, c.N, c.accumulate, 1.0f);
    }
}

static void tr_gemm(const TrainCtx& cx, const GemmCall& c) {
    int ksplit;
    TGemm g = gemm_args(cx, c, ksplit);
    g_last_stat_tiles = 0;
    const int M = c.M, N = c.N, K = c.K;
    if (gemm2_eligible(c)) {
        const bool akc = c.sak == 1, bkc = c.sbk == 1;
        int kper = ((K + ksplit - 1) / ksplit + 31) / 32 * 32;
        const long small = (long)((M + 63) / 64) * ((N + 63) / 64) * ((K + kper - 1) / kper);
        const bool wide = small < TR_LATENCY_WGS && wide_enabled();        // 64-wide chunks
        if (wide) kper = (kper + 63) / 64 * 64;
        ksplit = (K + kper - 1) / kper;
        g.kper = kper;
        TGemm2 a2{g, (ksplit == 1) ? c.stat_part : nullptr, 0, 0, ksplit};
        // 64 x 64 tiles everywhere by default: five workgroups per CU (88 VGPRs, 30 KB of LDS) whose load -> split ->
        // LDS -> MFMA phases overlap each other's; the 128-row / 128-column instantiations (two or four accumulator
        // tiles per wave sharing their fragments, two / three workgroups per CU) stay selectable for measurements.
        // PP_TRAIN_TILE_THR = workgroups a launch must still have for a larger tile; products of a B=32 step:
        // 128 -> 2.32 ms, 256 -> 2.22, 512 -> 2.16, 1024 -> 2.13, 2048 -> 2.05, 4096 -> 2.01, never -> 2.00
        const long big = (long)((M + 127) / 128) * ((N + 127) / 128) * ksplit;
        const long tall = (long)((M + 127) / 128) * ((N + 63) / 64) * ksplit;
        static long thr = -1;
        if (thr < 0) { const char* e = getenv("PP_TRAIN_TILE_THR"); thr = e ? atol(e) : (1l << 40); }
        // ... except for the products that are bound by the matrix pipe, not by memory: a long K against few bytes per
        // output (the transposed convolutions' forward and input-gradient products at a full-chip batch).  There the
        // 128 x 128 tile reads each LDS fragment for two accumulator tiles (PP_TRAIN_BIG_FLOPB: least FLOPs per byte of
        // operand + result traffic for the large tile; 0 = never)
        static double big_fb = -1.0;
        if (big_fb < 0) { const char* e = getenv("PP_TRAIN_BIG_FLOPB"); big_fb = e ? atof(e) : 0.0; }
        const double flop_per_byte = 2.0 * M * (double)N * K / (4.0 * ((double)M * K + (double)K * N + (double)M * N));
        const bool compute_bound = big_fb > 0.0 && flop_per_byte >= big_fb && big >= 2 * g_num_cus_train && ksplit == 1;
        if (N >= 128 && M > 64 && (big >= thr || compute_bound)) {
            dim3 grid((N + 127) / 128, (M + 127) / 128, ksplit);
            if (a2.stat_part) g_last_stat_tiles = (int)grid.y;
            launch_gemm2<2, 2, 32>(a2, akc, bkc, grid, cx.stream);
        } else if (M > 64 && tall >= thr) {      // (M <= 64: half of a 128-row tile would be zero rows -- the 64-channel weight gradients)
            dim3 grid((N + 63) / 64, (M + 127) / 128, ksplit);
            if (a2.stat_part) g_last_stat_tiles = (int)grid.y;
            launch_gemm2<2, 1, 32>(a2, akc, bkc, grid, cx.stream);
        } else {
            dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
            if (a2.stat_part) g_last_stat_tiles = (int)grid.y;
            if (wide) launch_gemm2<1, 1, 64>(a2, akc, bkc, grid, cx.stream);
            else launch_gemm2<1, 1, 32>(a2, akc, bkc, grid, cx.stream);
        }
        if (ksplit > 1) gemm_finish_split(cx, c, g, ksplit);
        return;
    }
    int kper = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
    ksplit = (K + kper - 1) / kper;
    g.kper = kper;
    dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
    PP_LAUNCH("k_tr_gemm", k_tr_gemm, grid, dim3(256), 0, cx.stream, g);
    if (ksplit > 1) gemm_finish_split(cx, c, g, ksplit);
}

static void tr_gemm(const TrainCtx& cx, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                    long ldc, int M, int N, int K, const float* bias, int accumulate, int ksplit, float* stat_part = nullptr,
                    bool defer = false) {
    tr_gemm(cx, GemmCall{A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias, accumulate, ksplit, stat_part, defer});
}

// The weight-gradient product (w: TN form, split-K, deferred reduction) and the input-gradient product (d) of one
// layer.  While both are a few hundred workgroups they go out as ONE launch (k_tr_gemm2_pair); otherwise one by one.
static void tr_gemm_pair(const TrainCtx& cx, const GemmCall& w, const GemmCall& d) {
    const bool forms = w.sak != 1 && w.sbk != 1 && d.sak == 1;
    if (forms && gemm2_eligible(w) && gemm2_eligible(d) && w.defer && w.stat_part == nullptr && d.stat_part == nullptr) {
        // 64-wide chunks while the launch lives on latency, the 32-wide ones (five workgroups per CU) beyond: two
        // half-filled launches of a large batch -- block3's 640 + 640 workgroups at B=32 -- fill the chip together
        static int pair_big = -1;      // PP_TRAIN_PAIR_BIG=0: pairs only in the latency regime (the round-3 first version)
        if (pair_big < 0) { const char* e = getenv("PP_TRAIN_PAIR_BIG"); pair_big = (e && e[0] == '0') ? 0 : 1; }
        int ks1, ks2, kper1, kper2, n1, n2;
        TGemm g1 = gemm_args(cx, w, ks1);
        int ksd;
        (void)gemm_args(cx, d, ksd);
        const auto tiles = [](const GemmCall& c) { return (long)((c.M + 63) / 64) * ((c.N + 63) / 64); };
        const bool wide = tiles(w) * ks1 + tiles(d) * ksd < 2 * TR_LATENCY_WGS && wide_enabled();
        const int kround = wide ? 64 : 32;
        auto plan = [kround](const GemmCall& c, int ks, int& kper, int& nks) {
            kper = ((c.K + ks - 1) / ks + kround - 1) / kround * kround;
            nks = (c.K + kper - 1) / kper;
            return (long)((c.M + 63) / 64) * ((c.N + 63) / 64) * nks;
        };
        const long wg1 = plan(w, ks1, kper1, n1);
        // (the second product's partial tiles, if it is split, lie behind the first one's region)
        const long region1 = (n1 > 1) ? ((long)n1 * w.M * w.N + 63) / 64 * 64 : 0;
        const long saved = g_arena_used;
        g_arena_used += region1;
        TGemm g2 = gemm_args(cx, d, ks2);
        g_arena_used = saved;
        const long wg2 = plan(d, ks2, kper2, n2);
        if (wide || pair_big) {
            g1.kper = kper1; g2.kper = kper2;
            TGemm2 p1{g1, nullptr, (w.N + 63) / 64, (w.M + 63) / 64, n1};
            TGemm2 p2{g2, nullptr, (d.N + 63) / 64, (d.M + 63) / 64, n2};
            g_last_stat_tiles = 0;
            const dim3 grid((unsigned)(wg1 + wg2));
            if (d.sbk == 1 && wide) PP_LAUNCH(g_gemm_tag, (k_tr_gemm2_pair<true, 64>), grid, dim3(256), 0, cx.stream, p1, p2);
            else if (d.sbk == 1) PP_LAUNCH(g_gemm_tag, (k_tr_gemm2_pair<true, 32>), grid, dim3(256), 0, cx.stream, p1, p2);
            else if (wide) PP_LAUNCH(g_gemm_tag, (k_tr_gemm2_pair<false, 64>), grid, dim3(256), 0, cx.stream, p1, p2);
            else PP_LAUNCH(g_gemm_tag, (k_tr_gemm2_pair<false, 32>), grid, dim3(256), 0, cx.stream, p1, p2);
            if (n1 > 1) gemm_finish_split(cx, w, g1, n1);                 // deferred: its region is now taken
            if (n2 > 1) gemm_finish_split(cx, d, g2, n2);                 // reduced at once (from behind that region)
            return;
        }
    }
    tr_gemm(cx, w);
    tr_gemm(cx, d);
}

// weight gradients: K = rows (pixels); enough slices to fill the chip, bounded by the partial buffer
static int wgrad_split(const TrainCtx& cx, int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    static long wgs = -1;      // PP_TRAIN_WGRAD_WGS: workgroups a weight-gradient product is cut into
    if (wgs < 0) { const char* e = getenv("PP_TRAIN_WGRAD_WGS"); wgs = e ? atol(e) : 1024; }
    long want = std::max<long>(1, wgs / tiles);
    want = std::min<long>(want, (K + 255) / 256);
    while (want > 1 && want * (long)M * N > cx.gemm_part_floats) --want;
    return (int)want;
}

// input gradients with a long K (the transposed convolutions: K = taps * channels) on a map of few tiles: K slices of
// >= 256 until the launch has a workgroup or two per CU (the partial tiles are added at once: the next kernel reads C)
static int dgrad_split(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    // (round 4, measured and dropped: K slices for the long-K input gradients of the transposed convolutions whose 64 x 64
    // tiles fill only half a resident round -- deconv3 at B=32: 640 workgroups x 64 chunks -- pair.deconv3 183.5 -> 173.9 us,
    // and the 4.4 us + 20 MB of the extra reduction launch give it back: the step 3.569 -> 3.571 ms)
    if (tiles >= 256) return 1;
    return (int)std::max<long>(1, std::min<long>(512 / tiles, K / 256));
}

// ------------------------------------------------------------------------------------------------------------
// element kernels
// ------------------------------------------------------------------------------------------------------------
// Index arithmetic of the element kernels in 32 bits: a division by a run-time divisor is a shift when the divisor is
// a power of two (channel counts, taps) and one 32-bit unsigned division otherwise (map widths).  The first versions
// of these kernels did four 64-bit divisions per thread -- several hundred instructions of address arithmetic around
// twenty of work: they were bound by vector issue, not by memory.
struct Div32 {
    unsigned d;
    int shift;      // log2(d) when d is a power of two, else -1
};
static Div32 make_div(unsigned d) {
    Div32 r{d, -1};
    if (d && (d & (d - 1)) == 0) { r.shift = 0; while ((1u << r.shift) < d) ++r.shift; }
    return r;
}
__device__ __forceinline__ void divmod32(unsigned n, const Div32& dv, unsigned& q, unsigned& r) {
    if (dv.shift >= 0) { q = n >> dv.shift; r = n & (dv.d - 1u); }
    else { q = n / dv.d; r = n - q * dv.d; }
}

__global__ __launch_bounds__(256) void k_tr_fill(float* p, long n, float v) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// depthwise 3x3, symmetric padding 1, stride S (ZeroPadding2D(1) + 'valid' == 'same' for stride 1):
// D[b,y,x,c] = sum_t X[b, y*S-1+dy, x*S-1+dx, c] * w[t][c]
// coef != NULL: X is the PRE-BatchNorm map Z of the layer before and the activation relu(z * sc + sh) is evaluated on the
// way in (the activation tensor of an in-block layer is never written: k_tr_bn_relu's pass and its output are gone)
__global__ __launch_bounds__(256) void k_tr_dw_fwd(const float* __restrict__ X, const float* __restrict__ w,
                                                   float* __restrict__ D, unsigned total, int ih, int iw, Div32 doh, Div32 dow,
                                                   Div32 dc4, int C, int S, const float4* __restrict__ coef) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;      // (pixel, channel quad); total < 2^32 (launcher)
    if (i >= total) return;
    unsigned p, cq, x, y, b;
    divmod32(i, dc4, p, cq);
    divmod32(p, dow, p, x);
    divmod32(p, doh, b, y);
    const int c = (int)cq * 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool bn = coef != nullptr;                          // uniform
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (bn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 q = coef[c + j]; sc[j] = q.x; sh[j] = q.y; }
    }
    // Straight-line code: all nine window loads are issued back to back from clamped (always valid) addresses and an
    // out-of-map tap is switched off through its WEIGHT.  (The first version skipped such taps with `continue`: nine
    // basic blocks, each load waited for before the next was issued -- block1's 84 MB took 46 us.)
    const float* xb = X + (size_t)b * ih * iw * C + c;
    float4 v[9], k[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = (int)y * S - 1 + dy;
        const bool yok = (unsigned)yy < (unsigned)ih;
        const int yc = yok ? yy : 0;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = (int)x * S - 1 + dx;
            const bool ok = yok & ((unsigned)xx < (unsigned)iw);
            const int xc = ((unsigned)xx < (unsigned)iw) ? xx : 0;
            v[dy * 3 + dx] = *reinterpret_cast<const float4*>(xb + ((size_t)yc * iw + xc) * C);
            const float4 kk = *reinterpret_cast<const float4*>(w + (size_t)(dy * 3 + dx) * C + c);
            k[dy * 3 + dx] = ok ? kk : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float4 a = v[t];
        if (bn) {
            a.x = fmaxf(fmaf(a.x, sc[0], sh[0]), 0.f); a.y = fmaxf(fmaf(a.y, sc[1], sh[1]), 0.f);
            a.z = fmaxf(fmaf(a.z, sc[2], sh[2]), 0.f); a.w = fmaxf(fmaf(a.w, sc[3], sh[3]), 0.f);
        }
        o.x = fmaf(a.x, k[t].x, o.x); o.y = fmaf(a.y, k[t].y, o.y); o.z = fmaf(a.z, k[t].z, o.z); o.w = fmaf(a.w, k[t].w, o.w);
    }
    *reinterpret_cast<float4*>(D + (size_t)i * 4) = o;        // D is [pixel][C]: element (pixel, quad) is at i * 4
}


// The same layer with the nine tap vectors of a thread's channels in REGISTERS: a workgroup owns a contiguous range of
// output pixels, a thread 4 channels of every (256 / (C / 4))-th pixel of the range (as the backward kernels).  Re-read
// per output, the taps were half of k_tr_dw_fwd's traffic through the CU's vector-memory path (18 loads of 16 bytes per
// 16 output bytes: block1's 84 MB took 30 us beside element kernels at 5-6 TB/s).
__global__ __launch_bounds__(256) void k_tr_dw_fwd_p(const float* __restrict__ X, const float* __restrict__ w,
                                                     float* __restrict__ D, int B, int ih, int iw, int oh, int ow, int C,
                                                     int S, const float4* __restrict__ coef) {
    const int tid = threadIdx.x;
    const int cq = C >> 2;
    const int q = tid % cq, ps = tid / cq, nps = 256 / cq;
    const long npix = (long)B * oh * ow;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = min(npix, p0 + per);
    float4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const float4*>(w + (size_t)t * C + 4 * q);
    const bool bn = coef != nullptr;                          // uniform
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (bn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 cfv = coef[4 * q + j]; sc[j] = cfv.x; sh[j] = cfv.y; }
    }
    int x = 0, y = 0, b = 0;
    if (p0 + ps < p1) {
        const unsigned up = (unsigned)(p0 + ps);             // pixel counts < 2^31 (launcher)
        const unsigned qy = up / (unsigned)ow;
        x = (int)(up - qy * (unsigned)ow);
        b = (int)(qy / (unsigned)oh);
        y = (int)(qy - (unsigned)b * (unsigned)oh);
    }
    for (long p = p0 + ps; p < p1; p += nps) {
        const float* xb = X + (size_t)b * ih * iw * C + 4 * q;
        float4 v[9];
        float m[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y * S - 1 + dy;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x * S - 1 + dx;
                const bool ok = ((unsigned)yy < (unsigned)ih) & ((unsigned)xx < (unsigned)iw);
                v[dy * 3 + dx] = *reinterpret_cast<const float4*>(xb + ((size_t)(ok ? yy : 0) * iw + (ok ? xx : 0)) * C);
                m[dy * 3 + dx] = ok ? 1.f : 0.f;
            }
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float4 a = v[t];
            if (bn) {
                a.x = fmaxf(fmaf(a.x, sc[0], sh[0]), 0.f); a.y = fmaxf(fmaf(a.y, sc[1], sh[1]), 0.f);
                a.z = fmaxf(fmaf(a.z, sc[2], sh[2]), 0.f); a.w = fmaxf(fmaf(a.w, sc[3], sh[3]), 0.f);
            }
            // (an out-of-map tap: its weight times zero -- the same +0 terms as k_tr_dw_fwd's zeroed weights)
            const float4 k = make_float4(wk[t].x * m[t], wk[t].y * m[t], wk[t].z * m[t], wk[t].w * m[t]);
            o.x = fmaf(a.x, k.x, o.x); o.y = fmaf(a.y, k.y, o.y); o.z = fmaf(a.z, k.z, o.z); o.w = fmaf(a.w, k.w, o.w);
        }
        *reinterpret_cast<float4*>(D + (size_t)p * C + 4 * q) = o;
        x += nps;
        while (x >= ow) { x -= ow; if (++y == oh) { y = 0; ++b; } }
    }
}

// gradient of the depthwise convolution with respect to its input:
// dX[b,yy,xx,c] (+)= sum over taps with (yy+1-dy) % S == 0 ... of dD[b,(yy+1-dy)/S,(xx+1-dx)/S,c] * w[t][c]
__global__ __launch_bounds__(256) void k_tr_dw_bwd_in(const float* __restrict__ dD, const float* __restrict__ w,
                                                      float* __restrict__ dX, unsigned total, Div32 dih, Div32 diw, int oh,
                                                      int ow, Div32 dc4, int C, int S, int accumulate) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;      // (input pixel, channel quad)
    if (i >= total) return;
    unsigned p, cq, uxx, uyy, b;
    divmod32(i, dc4, p, cq);
    divmod32(p, diw, p, uxx);
    divmod32(p, dih, b, uyy);
    const int c = (int)cq * 4, xx = (int)uxx, yy = (int)uyy;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    // straight-line: nine unconditional loads from clamped addresses, a tap without an output pixel multiplied by zero
    const float* db = dD + (size_t)b * oh * ow * C + c;
    float4 v[9];
    float m[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ty = yy + 1 - dy;
        const int y = (S == 2) ? (ty >> 1) : ty;             // S is 1 or 2 (train_step checks)
        const bool yok = (ty >= 0) & !(S == 2 && (ty & 1)) & (y < oh);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int tx = xx + 1 - dx;
            const int x = (S == 2) ? (tx >> 1) : tx;
            const bool ok = yok & (tx >= 0) & !(S == 2 && (tx & 1)) & (x < ow);
            v[dy * 3 + dx] = *reinterpret_cast<const float4*>(db + ((size_t)(ok ? y : 0) * ow + (ok ? x : 0)) * C);
            m[dy * 3 + dx] = ok ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 k = *reinterpret_cast<const float4*>(w + (size_t)t * C + c);
        o.x = fmaf(v[t].x * m[t], k.x, o.x); o.y = fmaf(v[t].y * m[t], k.y, o.y);
        o.z = fmaf(v[t].z * m[t], k.z, o.z); o.w = fmaf(v[t].w * m[t], k.w, o.w);
    }
    float4* dst = reinterpret_cast<float4*>(dX + (size_t)i * 4);
    if (accumulate) { const float4 q = *dst; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
    *dst = o;
}

// depthwise kernel gradient: dw[tap][c] = sum over output pixels of X[window position tap][c] * dD[pixel][c].
// A workgroup owns a CONTIGUOUS range of output pixels (the 3x3 windows of neighbouring pixels share their loads in
// L1 / L2); a thread owns 4 channels (16-byte loads) of every (256 / (C / 4))-th pixel of the range; the per-thread
// sums of the 9 taps meet in LDS and leave one partial row part[blk][9][C]; k_tr_reduce adds the rows in order.
__global__ __launch_bounds__(256) void k_tr_dw_bwd_w(const float* __restrict__ X, const float* __restrict__ dD,
                                                     float* __restrict__ part, int B, int ih, int iw, int oh, int ow,
                                                     int C, int S) {
    __shared__ float4 sred[3 * 256];                     // three taps at a time: 12 KB, eight workgroups per CU (the
                                                         // kernel is a few memory round trips per thread: it lives on
                                                         // resident waves; 36 KB for all nine taps allowed four)
    const int tid = threadIdx.x;
    const int cq = C >> 2;                               // channel quads: 8 .. 64 (C in {32, 64, 128, 256})
    const int q = tid % cq, ps = tid / cq, nps = 256 / cq;
    const long npix = (long)B * oh * ow;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = min(npix, p0 + per);
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    // (b, y, x) of the thread's first pixel by division once, then advanced by nps pixels per iteration
    int x = 0, y = 0, b = 0;
    if (p0 + ps < p1) {
        const unsigned up = (unsigned)(p0 + ps);             // pixel counts < 2^31 (launcher)
        const unsigned qy = up / (unsigned)ow;
        x = (int)(up - qy * (unsigned)ow);
        b = (int)(qy / (unsigned)oh);
        y = (int)(qy - (unsigned)b * (unsigned)oh);
    }
    for (long p = p0 + ps; p < p1; p += nps) {
        const float4 g = *reinterpret_cast<const float4*>(dD + (size_t)p * C + 4 * q);
        // straight-line window (as k_tr_dw_fwd): unconditional loads from clamped addresses, out-of-map taps times zero
        const float* xb = X + (size_t)b * ih * iw * C + 4 * q;
        float4 v[9];
        float m[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y * S - 1 + dy;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x * S - 1 + dx;
                const bool ok = ((unsigned)yy < (unsigned)ih) & ((unsigned)xx < (unsigned)iw);
                v[dy * 3 + dx] = *reinterpret_cast<const float4*>(xb + ((size_t)(ok ? yy : 0) * iw + (ok ? xx : 0)) * C);
                m[dy * 3 + dx] = ok ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float4& a = acc[t];
            a.x = fmaf(v[t].x * m[t], g.x, a.x); a.y = fmaf(v[t].y * m[t], g.y, a.y);
            a.z = fmaf(v[t].z * m[t], g.z, a.z); a.w = fmaf(v[t].w * m[t], g.w, a.w);
        }
        x += nps;
        while (x >= ow) { x -= ow; if (++y == oh) { y = 0; ++b; } }
    }
#pragma unroll
    for (int t0 = 0; t0 < 9; t0 += 3) {
        if (t0) __syncthreads();                         // the previous three taps have been read
#pragma unroll
        for (int t = 0; t < 3; ++t) sred[t * 256 + tid] = acc[t0 + t];
        __syncthreads();
        if (ps == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float4 v = acc[t0 + t];
                for (int k = 1; k < nps; ++k) {
                    const float4 o = sred[t * 256 + tid + k * cq];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 9 + t0 + t) * C + 4 * q) = v;
            }
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
        const long step = (long)gridDim.x * nsub;
        long r = (long)blockIdx.x * nsub + rsub;
        for (; r + 3 * step < rows; r += 4 * step) {        // four rows' loads in flight, added in row order
            float z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) z[u] = Z[(size_t)(r + u * step) * C + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += z[u]; q = fmaf(z[u], z[u], q); }
        }
        for (; r < rows; r += step) {
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
// (ntaps > 1: the partial rows are [2][ntaps * C] -- a transposed convolution's GEMM columns, tap-major -- and a
// channel's statistics run over all its taps)
// LPC lanes share a channel (16: a handful of partial rows; 64 / 256: the many row tiles of a large batch)
template <int LPC>
__global__ __launch_bounds__(256) void k_tr_bn_finalize(const float* __restrict__ part, int nparts, int C, float n_rows_arg,
                                                        const float* __restrict__ n_rows_dev, float momentum,
                                                        int unbiased_moving, float* __restrict__ stats,
                                                        float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                        int ntaps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float4* __restrict__ coef) {
    const int l = threadIdx.x & (LPC - 1);
    const int c = blockIdx.x * (256 / LPC) + threadIdx.x / LPC;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        // lane l adds the (partial row, tap) pairs l, l + LPC, ... in that order, EIGHT pairs' loads in flight at a time:
        // written as "load, add, load, add" the loop waited a full memory round trip per pair (the disassembly had a
        // vmcnt(0) per iteration) and this launch -- 20 per step -- took 5.6 us at B=2 / 9 us at B=32 for ~2 us of work
        const size_t N = (size_t)C * ntaps;
        const int total = nparts * ntaps;
        for (int i0 = l; i0 < total; i0 += LPC * 8) {
            float a[8], q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * LPC;
                const bool ok = i < total;
                const int ic = ok ? i : l;                         // (a valid pair; its value is dropped below)
                int pr = ic, t = 0;
                if (ntaps > 1) { pr = ic / ntaps; t = ic - pr * ntaps; }      // (uniform: only the transposed convolutions)
                const float* src = part + ((size_t)pr * 2) * N + (size_t)t * C + c;
                a[k] = src[0];
                q[k] = src[N];
                if (!ok) { a[k] = 0.f; q[k] = 0.f; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1 += a[k]; s2 += q[k]; }
        }
    }
#pragma unroll
    for (int off = (LPC > 64 ? 64 : LPC) / 2; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if (LPC > 64) {                                       // a whole workgroup per channel: the four wave sums in order
        __shared__ float sw[2][4];
        if ((threadIdx.x & 63) == 0) { sw[0][threadIdx.x >> 6] = s1; sw[1][threadIdx.x >> 6] = s2; }
        __syncthreads();
        s1 = ((sw[0][0] + sw[0][1]) + sw[0][2]) + sw[0][3];
        s2 = ((sw[1][0] + sw[1][1]) + sw[1][2]) + sw[1][3];
    }
    if (c >= C || l != 0) return;
    const float n_rows = fmaxf((n_rows_dev != nullptr) ? *n_rows_dev : n_rows_arg, 1.f);   // PFN: P * T, known on the device only
    const float mean = s1 / n_rows;
    float var = s2 / n_rows - mean * mean;
    var = fmaxf(var, 0.f);
    const float inv = 1.0f / sqrtf(var + TR_EPS);
    stats[2 * c] = mean;
    stats[2 * c + 1] = inv;
    if (coef != nullptr) {      // what the consumers of Z evaluate: act = z * sc + sh, zhat = z * inv + nmi
        const float sc = inv * gamma[c];
        coef[c] = make_float4(sc, fmaf(-mean, sc, beta[c]), inv, -mean * inv);
    }
    if (moving_mean != nullptr) {
        const float vm = (unbiased_moving && n_rows > 1.f) ? var * (n_rows / (n_rows - 1.f)) : var;
        moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
        moving_var[c] = moving_var[c] * momentum + vm * (1.f - momentum);
    }
}

// parity tap: the ReLU decisions of a layer as the backward kernels take them (k_tr_bn_bwd_reduce / _apply, k_tr_dw_bwd)
__global__ __launch_bounds__(256) void k_tr_relu_mask(const float* __restrict__ Z, const float4* __restrict__ coef, long n, int C,
                                                      unsigned char* __restrict__ mask) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 q = coef[(int)(i % C)];
    mask[i] = (fmaf(Z[i], q.x, q.y) > 0.f) ? 1 : 0;
}
void launch_relu_mask(const float* Z, const float4* coef, long n, int C, unsigned char* mask, hipStream_t s) {
    if (n <= 0) return;
    k_tr_relu_mask<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(Z, coef, n, C, mask);
}

// row of the [rows][C] matrix -> pixel row of the destination activation (identity, or the pixel shuffle of a
// transposed convolution with kernel == stride k: row = input pixel * k*k + tap)
struct RowMap { int k, in_h, in_w; };
__device__ __forceinline__ long map_row(const RowMap& m, long r) {
    if (m.k <= 1) return r;
    // rows < 2^31 (launcher): 32-bit arithmetic; kernel == stride k is 1, 2 or 4: the tap split is a shift
    const unsigned kk = (unsigned)(m.k * m.k), ur = (unsigned)r;
    unsigned p, tapr;
    if (m.k == 2) { p = ur >> 2; tapr = ur & 3u; }
    else if (m.k == 4) { p = ur >> 4; tapr = ur & 15u; }
    else { p = ur / kk; tapr = ur - p * kk; }
    const unsigned q = p / (unsigned)m.in_w, x = p - q * (unsigned)m.in_w;
    const unsigned b = q / (unsigned)m.in_h, y = q - b * (unsigned)m.in_h;
    const unsigned ti = tapr / (unsigned)m.k, tj = tapr - ti * (unsigned)m.k;
    return ((long)(b * m.in_h * m.k + y * m.k + ti) * ((long)m.in_w * m.k)) + (long)(x * m.k + tj);
}

// A[map(r)][co_off + c] = relu(Z[r][c] * sc + sh)   (coef[c] = (sc, sh, inv, -mean * inv), k_tr_bn_finalize)
__global__ __launch_bounds__(256) void k_tr_bn_relu(const float* __restrict__ Z, long rows, int C, const float4* __restrict__ coef,
                                                    float* __restrict__ A, int ld, int co_off, RowMap rm) {
    const unsigned c4n = (unsigned)C >> 2;                   // C is a power of two here (train_step checks): mask / shift
    const unsigned i = blockIdx.x * 256u + threadIdx.x;      // rows * C / 4 < 2^32 (launcher)
    if ((long)i >= rows * (long)c4n) return;
    const int c = (int)(i & (c4n - 1u)) * 4;
    const long r = (long)(i / c4n);
    const float4 z = *reinterpret_cast<const float4*>(Z + (size_t)i * 4);
    const float zz[4] = {z.x, z.y, z.z, z.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 q = coef[c + j];
        o[j] = fmaxf(fmaf(zz[j], q.x, q.y), 0.f);
    }
    *reinterpret_cast<float4*>(A + (size_t)map_row(rm, r) * ld + co_off + c) = make_float4(o[0], o[1], o[2], o[3]);
}

// backward of BN + ReLU, pass 1: g = dA * (act > 0); part[blk][0][c] = sum g, part[blk][1][c] = sum g * zhat.
// A workgroup owns a contiguous range of rows, a thread 4 channels (16-byte loads) of every (256 / (C / 4))-th row.
__global__ __launch_bounds__(256) void k_tr_bn_bwd_reduce(const float* __restrict__ dA, int ld, int co_off, RowMap rm,
                                                          const float* __restrict__ Z, long rows, int C,
                                                          const float4* __restrict__ coef, float* __restrict__ part) {
    __shared__ float4 s1[256], s2[256];
    const int tid = threadIdx.x;
    const int cq = C >> 2;
    const int q = tid % cq, rs = tid / cq, nrs = 256 / cq;
    float4 cf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cf[j] = coef[4 * q + j];
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    auto add_row = [&](const float4& z4, const float4& d4) {
        const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float act = fmaf(zz[j], cf[j].x, cf[j].y);
            const float zh = fmaf(zz[j], cf[j].z, cf[j].w);
            const float g = (act > 0.f) ? dd[j] : 0.f;
            a[j] += g; b[j] = fmaf(g, zh, b[j]);
        }
    };
    // four rows' loads in flight, added in row order (the same sums as row by row: a thread walked its ~20 rows one
    // memory round trip at a time -- 23 us per launch at B=32 for 8 us of traffic)
    long r = r0 + rs;
    for (; r + 3 * nrs < r1; r += 4 * nrs) {
        float4 z4[4], d4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long rr = r + (long)u * nrs;
            z4[u] = *reinterpret_cast<const float4*>(Z + (size_t)rr * C + 4 * q);
            d4[u] = *reinterpret_cast<const float4*>(dA + (size_t)map_row(rm, rr) * ld + co_off + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add_row(z4[u], d4[u]);
    }
    for (; r < r1; r += nrs) {
        const float4 z4 = *reinterpret_cast<const float4*>(Z + (size_t)r * C + 4 * q);
        const float4 d4 = *reinterpret_cast<const float4*>(dA + (size_t)map_row(rm, r) * ld + co_off + 4 * q);
        add_row(z4, d4);
    }
    s1[tid] = make_float4(a[0], a[1], a[2], a[3]);
    s2[tid] = make_float4(b[0], b[1], b[2], b[3]);
    __syncthreads();
    if (rs == 0) {
        float4 v = s1[tid], w = s2[tid];
        for (int k = 1; k < nrs; ++k) {
            const float4 o = s1[tid + k * cq], u = s2[tid + k * cq];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w;
        }
        *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + 4 * q) = v;
        *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + 4 * q) = w;
    }
}

// pass 2: dZ[r][c] = gamma * inv * (g - Sg / n - zhat * Sgz / n)   (sums[0] = Sg = d beta, sums[1] = Sgz = d gamma)
__global__ __launch_bounds__(256) void k_tr_bn_bwd_apply(const float* __restrict__ dA, int ld, int co_off, RowMap rm,
                                                         const float* __restrict__ Z, long rows, int C,
                                                         const float4* __restrict__ coef, const float* __restrict__ sums,
                                                         float inv_n, float* __restrict__ dZ) {
    const unsigned c4n = (unsigned)C >> 2;                   // C is a power of two (train_step checks)
    const unsigned i = blockIdx.x * 256u + threadIdx.x;      // (row, channel quad)
    if ((long)i >= rows * (long)c4n) return;
    const int c = (int)(i & (c4n - 1u)) * 4;
    const long r = (long)(i / c4n);
    const float4 z4 = *reinterpret_cast<const float4*>(Z + (size_t)i * 4);
    const float4 d4 = *reinterpret_cast<const float4*>(dA + (size_t)map_row(rm, r) * ld + co_off + c);
    const float4 m1 = *reinterpret_cast<const float4*>(sums + c), m2 = *reinterpret_cast<const float4*>(sums + C + c);
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
    const float mm1[4] = {m1.x, m1.y, m1.z, m1.w}, mm2[4] = {m2.x, m2.y, m2.z, m2.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 q = coef[c + j];
        const float act = fmaf(zz[j], q.x, q.y);
        const float zh = fmaf(zz[j], q.z, q.w);
        const float g = (act > 0.f) ? dd[j] : 0.f;
        o[j] = q.x * (g - mm1[j] * inv_n - zh * (mm2[j] * inv_n));
    }
    *reinterpret_cast<float4*>(dZ + (size_t)i * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// Backward of a stride-1 depthwise layer whose input is the BatchNorm + ReLU of the layer before it (Zp, coefp), one
// pass instead of three (k_tr_dw_bwd_w, k_tr_dw_bwd_in and the k_tr_bn_bwd_reduce of the layer before): with the 3x3
// window of dD around input pixel p in registers,
//   dX[p][c]      = sum_t dD[p + 1 - d_t][c] * w[t][c]                  (stored: the layer before's dA)
//   dw[t][c]     += x[p][c] * dD[p + 1 - d_t][c],  x = relu(zp * sc + sh)  (the same window, re-indexed by input pixel)
//   Sg[c]        += g,  Sgz[c] += g * zhat,  g = dX * (x > 0)            (the layer before's BatchNorm-backward sums)
// A workgroup owns a contiguous range of pixels, a thread 4 channels of every (256 / (C / 4))-th pixel; the 11 per-thread
// sums meet in LDS three at a time and leave one partial row part[blk][11][C] (rows 0-8: dw taps, 9: Sg, 10: Sgz).
__global__ __launch_bounds__(256) void k_tr_dw_bwd(const float* __restrict__ dD, const float* __restrict__ w,
                                                   const float* __restrict__ Zp, const float4* __restrict__ coefp,
                                                   float* __restrict__ dX, float* __restrict__ part, int B, int h, int wd, int C) {
    __shared__ float4 sred[3 * 256];
    const int tid = threadIdx.x;
    const int cq = C >> 2;
    const int q = tid % cq, ps = tid / cq, nps = 256 / cq;
    const long npix = (long)B * h * wd;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = min(npix, p0 + per);
    float4 acc[11];
#pragma unroll
    for (int t = 0; t < 11; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cf[j] = coefp[4 * q + j];
    // the nine tap vectors of this thread's channels stay in registers for all its pixels (re-read per pixel they were
    // half of the kernel's traffic through the CU's vector-memory path: 19 loads per 16 output bytes)
    float4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const float4*>(w + (size_t)t * C + 4 * q);
    int x = 0, y = 0;
    if (p0 + ps < p1) {
        const unsigned up = (unsigned)(p0 + ps);             // pixel counts < 2^31 (launcher)
        const unsigned qy = up / (unsigned)wd;
        x = (int)(up - qy * (unsigned)wd);
        y = (int)(qy % (unsigned)h);
    }
    for (long p = p0 + ps; p < p1; p += nps) {
        const float4 z4 = *reinterpret_cast<const float4*>(Zp + (size_t)p * C + 4 * q);
        const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
        float xa[4], zh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xa[j] = fmaxf(fmaf(zz[j], cf[j].x, cf[j].y), 0.f);
            zh[j] = fmaf(zz[j], cf[j].z, cf[j].w);
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        // straight-line window: nine unconditional loads from clamped offsets, an out-of-map tap multiplied by zero
        // (with a branch per tap every load was waited for before the next one was issued)
        float4 v[9];
        float m[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int oy = y + 1 - dy;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ox = x + 1 - dx;
                const bool ok = ((unsigned)oy < (unsigned)h) & ((unsigned)ox < (unsigned)wd);
                // the output pixel (oy, ox) of the same frame: p + (1 - dy) * wd + (1 - dx)
                const int off = ok ? (1 - dy) * wd + (1 - dx) : 0;
                v[dy * 3 + dx] = *reinterpret_cast<const float4*>(dD + (size_t)(p + off) * C + 4 * q);
                m[dy * 3 + dx] = ok ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 k = wk[t];
            const float4 u = make_float4(v[t].x * m[t], v[t].y * m[t], v[t].z * m[t], v[t].w * m[t]);
            o.x = fmaf(u.x, k.x, o.x); o.y = fmaf(u.y, k.y, o.y); o.z = fmaf(u.z, k.z, o.z); o.w = fmaf(u.w, k.w, o.w);
            float4& a = acc[t];
            a.x = fmaf(xa[0], u.x, a.x); a.y = fmaf(xa[1], u.y, a.y); a.z = fmaf(xa[2], u.z, a.z); a.w = fmaf(xa[3], u.w, a.w);
        }
        *reinterpret_cast<float4*>(dX + (size_t)p * C + 4 * q) = o;
        const float g0 = (xa[0] > 0.f) ? o.x : 0.f, g1 = (xa[1] > 0.f) ? o.y : 0.f;
        const float g2 = (xa[2] > 0.f) ? o.z : 0.f, g3 = (xa[3] > 0.f) ? o.w : 0.f;
        acc[9].x += g0; acc[9].y += g1; acc[9].z += g2; acc[9].w += g3;
        acc[10].x = fmaf(g0, zh[0], acc[10].x); acc[10].y = fmaf(g1, zh[1], acc[10].y);
        acc[10].z = fmaf(g2, zh[2], acc[10].z); acc[10].w = fmaf(g3, zh[3], acc[10].w);
        x += nps;
        while (x >= wd) { x -= wd; if (++y == h) y = 0; }
    }
#pragma unroll
    for (int t0 = 0; t0 < 12; t0 += 3) {
        if (t0) __syncthreads();                         // the previous three sums have been read
#pragma unroll
        for (int t = 0; t < 3; ++t) if (t0 + t < 11) sred[t * 256 + tid] = acc[t0 + t];
        __syncthreads();
        if (ps == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t0 + t >= 11) continue;
                float4 v = acc[t0 + t];
                for (int k = 1; k < nps; ++k) {
                    const float4 u = sred[t * 256 + tid + k * cq];
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 11 + t0 + t) * C + 4 * q) = v;
            }
        }
    }
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
    const int* pprefix;        // [batch + 1] exclusive prefix of npillars (k_tr_pfn_rows): the kernels walk the batch's
                               // REAL pillars, not the batch * max_voxels slots (12 000 per frame, a quarter of them used)
    float4* rec;               // [pillars of the batch][2] pillar records (k_tr_pfn_lin writes them, the other kernels read)
};
// global pillar number gp -> (frame, pillar of the frame); bcur: the wave's frame cursor (gp only grows)
__device__ __forceinline__ void pillar_of(const PfnT& p, int gp, int& bcur, int& b, int& pid) {
    while (gp >= p.pprefix[bcur + 1]) ++bcur;
    b = bcur;
    pid = gp - p.pprefix[bcur];
}

__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// ---- round 3: the PFN kernels without the pre-BatchNorm tensor Y[sum N][C] ----
// A pillar row's Dense output is ten FMAs per channel; the tensor Y (268 MB at B=32) cost a write and three reads, and
// every kernel walked a pillar's rows one dependent load at a time.  Now a wave loads its pillar's points ONCE, 64 at
// a time, one per lane (coalesced), broadcasts them with v_readlane and RECOMPUTES y = features . W wherever it is
// needed (the same FMA chain everywhere: the statistics, the max, the gradients all see bit-identical values); no
// memory access inside the per-point loop.
struct PtsBatch { float x, y, z, it; };
__device__ __forceinline__ PtsBatch pfn_load_batch(const PfnT& p, long row0, int n, int j0, int lane) {
    PtsBatch r{0.f, 0.f, 0.f, 0.f};
    const int j = j0 + lane;
    if (j < n) {
        const float* q = p.pts_sorted + (size_t)(row0 + j) * p.F;
        if (p.F > 3) { const float4 v = *reinterpret_cast<const float4*>(q); r.x = v.x; r.y = v.y; r.z = v.z; r.it = v.w; }
        else { r.x = q[0]; r.y = q[1]; r.z = q[2]; }
    }
    return r;
}
__device__ __forceinline__ float pfn_bcast(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// The decorated features of a pillar row from wave-uniform coordinates, in a CANONICAL order that does not depend on
// the configuration: x y z | intensity | xyz - mean | xy - centre | norm.  The configuration lives in the weights: a
// kernel loads row canon_row(c) of the Dense kernel for canonical feature c (none -- zeros -- for the intensity of a
// 3-feature cloud or the norm without with_distance), so the ten-FMA chain visits the real rows in their own order
// with exact no-ops in between: bit-identical to the row-order chain, and no per-point selects.  (The first version
// ordered the features by the configuration with selects and evaluated the norm's sqrt before selecting it away:
// ~95 wave-instructions per point in k_tr_pfn_lin, VALU-bound.)
__device__ __forceinline__ int pfn_canon_row(const PfnT& p, int c) {     // Dense-kernel row of canonical feature c, or -1
    const bool f4 = p.F > 3;
    if (c < 3) return c;
    if (c == 3) return f4 ? 3 : -1;
    const int r = (f4 ? 4 : 3) + (c - 4);                  // c = 4..8: the five offsets; c = 9: the norm
    if (c == 9 && !p.with_distance) return -1;
    return (r < p.FA) ? r : -1;
}
__device__ __forceinline__ void pfn_features(const PfnT& p, float x, float y, float z, float it, float mx, float my, float mz,
                                             float cx, float cy, float (&f)[10]) {
    f[0] = x; f[1] = y; f[2] = z; f[3] = it;
    f[4] = x - mx; f[5] = y - my; f[6] = z - mz; f[7] = x - cx; f[8] = y - cy;
    f[9] = 0.f;
    if (p.with_distance) f[9] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));   // (uniform branch)
}
// w[c][q] = Dense kernel row of canonical feature c, this lane's channels
template <int CPL>
__device__ __forceinline__ void pfn_load_weights(const PfnT& p, int lane, float (&w)[10][CPL]) {
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        const int r = pfn_canon_row(p, c);
#pragma unroll
        for (int q = 0; q < CPL; ++q) w[c][q] = (r >= 0 && lane * CPL + q < p.C) ? p.W[r * p.C + lane * CPL + q] : 0.f;
    }
}
// pillar geometry with the first 64 points already in the lanes: frame, row range, mean (lane-strided partial sums,
// then the xor tree), centre
struct PillarHead { int b, pid, n; long row0; float mx, my, mz, cx, cy; PtsBatch first; };
__device__ __forceinline__ void pillar_head(const PfnT& p, int gp, int lane, int& bcur, PillarHead& o) {
    pillar_of(p, gp, bcur, o.b, o.pid);
    const int* ps = p.pillar_start + (size_t)o.b * (p.max_voxels + 1);
    const int start = ps[o.pid];
    o.n = min(ps[o.pid + 1] - start, p.T);
    o.row0 = (long)p.offsets[o.b] + start;
    o.first = pfn_load_batch(p, o.row0, o.n, 0, lane);
    float sx = o.first.x, sy = o.first.y, sz = o.first.z;        // (lanes >= n hold zeros)
    for (int j0 = 64; j0 < o.n; j0 += 64) {
        const PtsBatch t = pfn_load_batch(p, o.row0, o.n, j0, lane);
        sx += t.x; sy += t.y; sz += t.z;
    }
    sx = wave_sum_f(sx); sy = wave_sum_f(sy); sz = wave_sum_f(sz);
    const float fn = (float)o.n;
    o.mx = sx / fn; o.my = sy / fn; o.mz = sz / fn;
    const int cell = p.pillar_cell[(size_t)o.b * p.max_voxels + o.pid];
    const int xi = cell % p.nx, yi = (cell / p.nx) % p.ny;
    o.cx = __fadd_rn(__fmul_rn((float)xi, p.vx), p.x_off);      // model/pointpillars.py:156-171
    o.cy = __fadd_rn(__fmul_rn((float)yi, p.vy), p.y_off);
}
// The pillar record k_tr_pfn_lin leaves for the three kernels behind it: (row0, n, frame * max_voxels + pillar, cell),
// (mean x, y, z, -).  One 32-byte wave-uniform load instead of the frame search, four dependent index loads, a second
// pass over the points and three wave reductions per pillar and kernel.
__device__ __forceinline__ void pillar_record_store(const PfnT& p, int gp, const PillarHead& h, int lane) {
    if (lane != 0) return;
    const int cell = p.pillar_cell[(size_t)h.b * p.max_voxels + h.pid];
    // (row0, n, frame * max_voxels + pillar, canvas row = frame * ny * nx + y * nx + x), (mean x y z, centre x), (centre y):
    // the readers do no index arithmetic (three integer divisions per pillar and kernel in the first version)
    const int ncanvas = p.nx * p.ny;
    p.rec[3 * (size_t)gp] = make_float4(__int_as_float((int)h.row0), __int_as_float(h.n),
                                        __int_as_float(h.b * p.max_voxels + h.pid), __int_as_float(h.b * ncanvas + cell % ncanvas));
    p.rec[3 * (size_t)gp + 1] = make_float4(h.mx, h.my, h.mz, h.cx);
    p.rec[3 * (size_t)gp + 2] = make_float4(h.cy, 0.f, 0.f, 0.f);
}
struct PillarRec { int n, slot, crow; long row0; float mx, my, mz, cx, cy; PtsBatch first; };   // crow: the pillar's canvas row
// Round 4: the record-based kernels walk their pillars through a two-stage software pipeline -- while pillar i is
// computed the first point batch of pillar i + 1 is in flight (its address needs that pillar's record) and so is the
// record of pillar i + 2.  A wave has ~13 pillars of ~5 points each: processed one by one, every pillar was a chain of
// two dependent memory round trips with a few hundred instructions of work behind it, and the kernels took 70-120 us
// for ~15 us of arithmetic.
struct PillarRecRaw { float4 r0, r1; float cy; };
__device__ __forceinline__ PillarRecRaw pillar_record_issue(const PfnT& p, int gp) {
    PillarRecRaw r;
    r.r0 = p.rec[3 * (size_t)gp];
    r.r1 = p.rec[3 * (size_t)gp + 1];
    r.cy = p.rec[3 * (size_t)gp + 2].x;
    return r;
}
__device__ __forceinline__ void pillar_record_finish(const PfnT& p, const PillarRecRaw& raw, int lane, PillarRec& o);
__device__ __forceinline__ void pillar_record_load(const PfnT& p, int gp, int lane, PillarRec& o) {
    pillar_record_finish(p, pillar_record_issue(p, gp), lane, o);
}
__device__ __forceinline__ void pillar_record_finish(const PfnT& p, const PillarRecRaw& raw, int lane, PillarRec& o) {
    const float4 r0 = raw.r0, r1 = raw.r1;
    o.row0 = (long)__builtin_amdgcn_readfirstlane(__float_as_int(r0.x));
    o.n = __builtin_amdgcn_readfirstlane(__float_as_int(r0.y));
    o.slot = __builtin_amdgcn_readfirstlane(__float_as_int(r0.z));
    o.crow = __builtin_amdgcn_readfirstlane(__float_as_int(r0.w));
    o.first = pfn_load_batch(p, o.row0, o.n, 0, lane);
    o.mx = r1.x; o.my = r1.y; o.mz = r1.z;
    o.cx = r1.w; o.cy = raw.cy;                                 // model/pointpillars.py:156-171, evaluated by k_tr_pfn_lin
}
// for (every pillar of this wave) body(h), h = its record with the first point batch loaded -- pipelined as above.
// prefetch(hn) starts the loads of the NEXT pillar that depend on its record; take() moves what they returned into the
// current pillar's variables (called before the next prefetch overwrites them).
template <class Pre, class Take, class Body>
__device__ __forceinline__ void pfn_for_pillars(const PfnT& p, int total, int lane, int wave, Pre&& prefetch, Take&& take,
                                                Body&& body) {
    const int stride = (int)gridDim.x * 4;
    int gp = (int)blockIdx.x * 4 + wave;
    if (gp >= total) return;
    PillarRec h, hn;
    pillar_record_finish(p, pillar_record_issue(p, gp), lane, hn);
    prefetch(hn);
    PillarRecRaw raw = pillar_record_issue(p, min(gp + stride, total - 1));
    for (; gp < total; gp += stride) {
        h = hn;
        take();
        if (gp + stride < total) {
            pillar_record_finish(p, raw, lane, hn);
            prefetch(hn);
            raw = pillar_record_issue(p, min(gp + 2 * stride, total - 1));
        }
        body(h);
    }
}

// y[q] = features . W[:, lane * CPL + q] (rows >= FA of w are zero)
template <int CPL>
__device__ __forceinline__ void pfn_dense(const float (&f)[10], const float (&w)[10][CPL], float (&y)[CPL]) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) v = fmaf(f[k], w[k][q], v);
        y[q] = v;
    }
}
// Round 4: the Dense row in FOLDED form.  The layer is linear in the decorated features, so with pillar-local
// coordinates x' = x - centre_x, y' = y - centre_y (the reference's f_center features: small numbers)
//   y = (w0 + w4 + w7) x' + (w1 + w5 + w8) y' + (w2 + w6) z + w3 i + w9 |p|  +  K,
//   K = w0 cx + w1 cy - w4 (mean_x - cx) - w5 (mean_y - cy) - w6 mean_z          (one constant per pillar and channel)
// -- five FMAs per point and channel instead of ten (the inference kernel's arithmetic, pfn.hip; canonical feature
// order: x y z | i | xyz - mean | xy - centre | norm).  The SAME chain runs in all four training kernels, so the
// statistics, the max and the gradients see bit-identical values, as before.  The Dense-kernel gradient folds the same
// way: per pillar S0 = sum dy, Sx' = sum x' dy, ... (six FMAs per point and channel instead of ten) and
//   dw0 += Sx' + cx S0, dw4 += Sx' - (mean_x - cx) S0, dw7 += Sx', (y alike), dw2 += Sz, dw6 += Sz - mean_z S0, ...
// with no large-coordinate cancellation (x', y' are within a pillar).
template <int CPL>
struct PfnFold {
    float wsx[CPL], wsy[CPL], wsz[CPL], wi[CPL], wn[CPL];     // per-point weights
    float w0[CPL], w1[CPL], w4[CPL], w5[CPL], w6[CPL];        // the pillar constant's weights
};
template <int CPL>
__device__ __forceinline__ void pfn_fold(const float (&w)[10][CPL], PfnFold<CPL>& o) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        o.wsx[q] = (w[0][q] + w[4][q]) + w[7][q];
        o.wsy[q] = (w[1][q] + w[5][q]) + w[8][q];
        o.wsz[q] = w[2][q] + w[6][q];
        o.wi[q] = w[3][q];
        o.wn[q] = w[9][q];
        o.w0[q] = w[0][q]; o.w1[q] = w[1][q]; o.w4[q] = w[4][q]; o.w5[q] = w[5][q]; o.w6[q] = w[6][q];
    }
}
// K[q] of a pillar (wave-uniform geometry, this lane's channels)
template <int CPL>
__device__ __forceinline__ void pfn_pillar_const(const PfnFold<CPL>& o, float cx, float cy, float mx, float my, float mz,
                                                 float (&K)[CPL]) {
    const float dmx = mx - cx, dmy = my - cy;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        float k = cx * o.w0[q];
        k = fmaf(cy, o.w1[q], k);
        k = fmaf(-dmx, o.w4[q], k);
        k = fmaf(-dmy, o.w5[q], k);
        k = fmaf(-mz, o.w6[q], k);
        K[q] = k;
    }
}
// the point's coordinates in the form the folded chain takes them (wave-uniform): x', y', z, intensity, norm
struct PfnPt { float xl, yl, z, it, nrm; };
__device__ __forceinline__ PfnPt pfn_point(const PfnT& p, float x, float y, float z, float it, float cx, float cy) {
    PfnPt o;
    o.xl = x - cx; o.yl = y - cy; o.z = z; o.it = it;
    o.nrm = 0.f;
    if (p.with_distance) o.nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));   // (uniform branch)
    return o;
}
template <int CPL>
__device__ __forceinline__ void pfn_y(const PfnFold<CPL>& o, const PfnPt& t, const float (&K)[CPL], float (&y)[CPL]) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        float v = K[q];
        v = fmaf(t.xl, o.wsx[q], v);
        v = fmaf(t.yl, o.wsy[q], v);
        v = fmaf(t.z, o.wsz[q], v);
        v = fmaf(t.it, o.wi[q], v);       // (exact no-ops when the cloud has no intensity / the layer no norm feature:
        v = fmaf(t.nrm, o.wn[q], v);      //  those rows of the folded weights are zero)
        y[q] = v;
    }
}
// for (every point j of the pillar) { pt = its folded-form coordinates (wave-uniform); BODY }
#define PFN_FOR_POINTS2(P, H, LANE, BODY)                                                                \
    for (int j0_ = 0; j0_ < (H).n; j0_ += 64) {                                                          \
        const PtsBatch bt_ = (j0_ == 0) ? (H).first : pfn_load_batch((P), (H).row0, (H).n, j0_, (LANE)); \
        const int cnt_ = min(64, (H).n - j0_);                                                           \
        for (int jj_ = 0; jj_ < cnt_; ++jj_) {                                                           \
            const int j = j0_ + jj_;                                                                     \
            const PfnPt pt = pfn_point((P), pfn_bcast(bt_.x, jj_), pfn_bcast(bt_.y, jj_), pfn_bcast(bt_.z, jj_), \
                                       pfn_bcast(bt_.it, jj_), (H).cx, (H).cy);                          \
            BODY                                                                                         \
        }                                                                                                \
    }

// for (every point j of the pillar) { f = its features (wave-uniform); BODY }
#define PFN_FOR_POINTS(P, H, LANE, BODY)                                                                 \
    for (int j0_ = 0; j0_ < (H).n; j0_ += 64) {                                                          \
        const PtsBatch bt_ = (j0_ == 0) ? (H).first : pfn_load_batch((P), (H).row0, (H).n, j0_, (LANE)); \
        const int cnt_ = min(64, (H).n - j0_);                                                           \
        for (int jj_ = 0; jj_ < cnt_; ++jj_) {                                                           \
            const int j = j0_ + jj_;                                                                     \
            float f[10];                                                                                 \
            pfn_features((P), pfn_bcast(bt_.x, jj_), pfn_bcast(bt_.y, jj_), pfn_bcast(bt_.z, jj_),       \
                         pfn_bcast(bt_.it, jj_), (H).mx, (H).my, (H).mz, (H).cx, (H).cy, f);             \
            BODY                                                                                         \
        }                                                                                                \
    }

// part[blk][0/1][c] = sums of y, y^2 over this workgroup's rows
template <int CPL>
__global__ __launch_bounds__(256) void k_tr_pfn_lin(PfnT p, float* __restrict__ part) {
    __shared__ float sp[4][2][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    PfnFold<CPL> fw;
    {
        float w[10][CPL];
        pfn_load_weights<CPL>(p, lane, w);
        pfn_fold<CPL>(w, fw);
    }
    float s1[CPL], s2[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) s1[q] = s2[q] = 0.f;
    const int total = p.pprefix[p.batch];
    int bcur = 0;
    for (int gp = (int)blockIdx.x * 4 + wave; gp < total; gp += (int)gridDim.x * 4) {
        PillarHead h;
        pillar_head(p, gp, lane, bcur, h);
        pillar_record_store(p, gp, h, lane);
        float K[CPL];
        pfn_pillar_const<CPL>(fw, h.cx, h.cy, h.mx, h.my, h.mz, K);
        PFN_FOR_POINTS2(p, h, lane, {
            (void)j;
            float y[CPL];
            pfn_y<CPL>(fw, pt, K, y);
_Pragma("unroll")
            for (int q = 0; q < CPL; ++q) { s1[q] += y[q]; s2[q] = fmaf(y[q], y[q], s2[q]); }
        })
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
__global__ __launch_bounds__(256) void k_tr_pfn_max(PfnT p, const float* __restrict__ stats,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ feat, int* __restrict__ arg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    PfnFold<CPL> fw;
    {
        float w[10][CPL];
        pfn_load_weights<CPL>(p, lane, w);
        pfn_fold<CPL>(w, fw);
    }
    float sc[CPL], sh[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane * CPL + q;
        const float inv = (c < C) ? stats[2 * c + 1] * gamma[c] : 0.f;
        sc[q] = inv;
        sh[q] = (c < C) ? beta[c] - stats[2 * c] * inv : 0.f;
    }
    const int total = p.pprefix[p.batch];
    pfn_for_pillars(p, total, lane, wave, [](const PillarRec&) {}, []() {}, [&](const PillarRec& h) {
        float best[CPL];
        int bi[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) { best[q] = -3.0e38f; bi[q] = -2; }
        float K[CPL];
        pfn_pillar_const<CPL>(fw, h.cx, h.cy, h.mx, h.my, h.mz, K);
        PFN_FOR_POINTS2(p, h, lane, {
            float y[CPL];
            pfn_y<CPL>(fw, pt, K, y);
_Pragma("unroll")
            for (int q = 0; q < CPL; ++q) {
                const float v = fmaf(y[q], sc[q], sh[q]);
                if (v > best[q]) { best[q] = v; bi[q] = j; }
            }
        })
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane * CPL + q;
            if (c >= C) continue;
            if (h.n < p.T && sh[q] > best[q]) { best[q] = sh[q]; bi[q] = -1; }   // a zero-padded row: Dense(0) = 0 -> BN
            if (!(best[q] > 0.f)) { best[q] = 0.f; bi[q] = -2; }
            feat[(size_t)h.slot * C + c] = best[q];                              // rows of feat / arg: frame * max_voxels + pillar
            arg[(size_t)h.slot * C + c] = bi[q];
        }
    });
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
// (y of the winning row recomputed from that row's point: a per-lane gather out of the lines the wave has just read)
template <int CPL>
__global__ __launch_bounds__(256) void k_tr_pfn_bwd_reduce(PfnT p, const float* __restrict__ stats,
                                                           const int* __restrict__ arg, const float* __restrict__ dcanvas,
                                                           float* __restrict__ part) {
    __shared__ float sp[4][2][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    PfnFold<CPL> fw;
    {
        float w[10][CPL];
        pfn_load_weights<CPL>(p, lane, w);
        pfn_fold<CPL>(w, fw);
    }
    float s1[CPL], s2[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) s1[q] = s2[q] = 0.f;
    const int total = p.pprefix[p.batch];
    int an[CPL], ac[CPL];                               // winning rows / canvas gradients of the next and the current pillar
    float gn[CPL], gc[CPL];
    auto red_prefetch = [&](const PillarRec& hn) {
        const size_t crow = (size_t)hn.crow;                   // canvas row of the pillar
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane * CPL + q;
            an[q] = (c < C) ? arg[(size_t)hn.slot * C + c] : -2;
            gn[q] = (c < C) ? dcanvas[crow * C + c] : 0.f;
        }
    };
    auto red_take = [&]() {
#pragma unroll
        for (int q = 0; q < CPL; ++q) { ac[q] = an[q]; gc[q] = gn[q]; }
    };
    pfn_for_pillars(p, total, lane, wave, red_prefetch, red_take, [&](const PillarRec& h) {
        float K[CPL];
        pfn_pillar_const<CPL>(fw, h.cx, h.cy, h.mx, h.my, h.mz, K);
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane * CPL + q;
            if (c >= C) continue;
            const int a = ac[q];
            if (a == -2) continue;
            const float g = gc[q];
            float y = 0.f;
            if (a >= 0) {      // the winning row's y: the same folded chain as the forward kernels (per-lane point)
                const float* pr = p.pts_sorted + (size_t)(h.row0 + a) * p.F;
                const PfnPt pt = pfn_point(p, pr[0], pr[1], pr[2], (p.F > 3) ? pr[3] : 0.f, h.cx, h.cy);
                float v = K[q];
                v = fmaf(pt.xl, fw.wsx[q], v);
                v = fmaf(pt.yl, fw.wsy[q], v);
                v = fmaf(pt.z, fw.wsz[q], v);
                v = fmaf(pt.it, fw.wi[q], v);
                v = fmaf(pt.nrm, fw.wn[q], v);
                y = v;
            }
            const float yh = (y - stats[2 * c]) * stats[2 * c + 1];
            s1[q] += g; s2[q] = fmaf(g, yh, s2[q]);
        }
    });
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
__global__ __launch_bounds__(256) void k_tr_pfn_bwd_apply(PfnT p, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const int* __restrict__ arg,
                                                          const float* __restrict__ dcanvas, const float* __restrict__ sums,
                                                          const float* __restrict__ n_rows_dev, float* __restrict__ part) {
    __shared__ float sp[4][10][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    float dw[10][CPL];                                      // (canonical feature order, as pfn_features)
    PfnFold<CPL> fw;
    {
        float w[10][CPL];
        pfn_load_weights<CPL>(p, lane, w);
        pfn_fold<CPL>(w, fw);
    }
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
    const int total = p.pprefix[p.batch];
    int an[CPL], a[CPL];                                // winning rows / canvas gradients of the next and the current pillar
    float gn[CPL], g[CPL];
    auto app_prefetch = [&](const PillarRec& hn) {
        const size_t crow = (size_t)hn.crow;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = lane * CPL + q;
            an[q] = (c < C) ? arg[(size_t)hn.slot * C + c] : -2;
            gn[q] = (c < C) ? dcanvas[crow * C + c] : 0.f;        // (issued with the index load, masked in app_take)
        }
    };
    auto app_take = [&]() {
#pragma unroll
        for (int q = 0; q < CPL; ++q) { a[q] = an[q]; g[q] = (an[q] >= 0) ? gn[q] : 0.f; }
    };
    pfn_for_pillars(p, total, lane, wave, app_prefetch, app_take, [&](const PillarRec& h) {
        float K[CPL];
        pfn_pillar_const<CPL>(fw, h.cx, h.cy, h.mx, h.my, h.mz, K);
        float S0[CPL], Sx[CPL], Sy[CPL], Sz[CPL], Si[CPL], Sn[CPL];     // the pillar's sums of dy, x' dy, y' dy, z dy, i dy, |p| dy
#pragma unroll
        for (int q = 0; q < CPL; ++q) S0[q] = Sx[q] = Sy[q] = Sz[q] = Si[q] = Sn[q] = 0.f;
        PFN_FOR_POINTS2(p, h, lane, {
            float y[CPL];
            pfn_y<CPL>(fw, pt, K, y);
_Pragma("unroll")
            for (int q = 0; q < CPL; ++q) {
                const float yh = (y[q] - mean[q]) * inv[q];
                const float dy = gi[q] * (((a[q] == j) ? g[q] : 0.f) - m1[q] - yh * m2[q]);
                S0[q] += dy;
                Sx[q] = fmaf(pt.xl, dy, Sx[q]);
                Sy[q] = fmaf(pt.yl, dy, Sy[q]);
                Sz[q] = fmaf(pt.z, dy, Sz[q]);
                Si[q] = fmaf(pt.it, dy, Si[q]);
                Sn[q] = fmaf(pt.nrm, dy, Sn[q]);
            }
        })
        {   // canonical rows: 0 1 2 = x y z, 3 = intensity, 4 5 6 = xyz - mean, 7 8 = xy - centre, 9 = norm
            const float dmx = h.mx - h.cx, dmy = h.my - h.cy;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                dw[0][q] += fmaf(h.cx, S0[q], Sx[q]);
                dw[1][q] += fmaf(h.cy, S0[q], Sy[q]);
                dw[2][q] += Sz[q];
                dw[3][q] += Si[q];
                dw[4][q] += fmaf(-dmx, S0[q], Sx[q]);
                dw[5][q] += fmaf(-dmy, S0[q], Sy[q]);
                dw[6][q] += fmaf(-h.mz, S0[q], Sz[q]);
                dw[7][q] += Sx[q];
                dw[8][q] += Sy[q];
                dw[9][q] += Sn[q];
            }
        }
    });
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int q = 0; q < CPL; ++q) sp[wave][k][lane * CPL + q] = dw[k][q];
    __syncthreads();
    for (int cf = 0; cf < 10; ++cf) {                       // canonical feature -> its row of the Dense-kernel gradient
        const int k = pfn_canon_row(p, cf);
        if (k < 0) continue;
        for (int c = threadIdx.x; c < C; c += 256)
            part[(size_t)blockIdx.x * p.FA * C + (size_t)k * C + c] = ((sp[0][cf][c] + sp[1][cf][c]) + sp[2][cf][c]) + sp[3][cf][c];
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

// This step's forward weights as two float16 pieces each (hi = rne(w), mid = rne(w - hi)), in the operand layout of the
// fused forward kernels (k_sep_u<..., TR = 1 | 2>): [K / 16][piece][N][16] per product -- the pointwise kernels (Keras
// [cin][cout]), the transposed convolutions' kernels as GEMM operands (Keras [k][k][cout][cin]: W(kk, n) = K[n][kk]) --
// from the flat parameter buffer.  One launch for all of them; the packed head matrix follows in a launch of its own
// (it is packed from the three head kernels just before).
struct SplitPwJob { const float* src; long dst; int K, N; long sk, sn; long first; };   // W(k, n) = src[k * sk + n * sn]; dst: 16-bit words; first: global element index
struct SplitPwTable { int n; long total; SplitPwJob job[40]; };
__global__ __launch_bounds__(256) void k_tr_split_pw(unsigned short* __restrict__ out, SplitPwTable t) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // one weight per thread
    if (i >= t.total) return;
    int j = 0;
    while (j + 1 < t.n && i >= t.job[j + 1].first) ++j;
    const SplitPwJob q = t.job[j];
    const long e = i - q.first;
    int k, n;                                                // consecutive threads walk the source's unit-stride axis
    if (q.sn == 1) { n = (int)(e % q.N); k = (int)(e / q.N); } else { k = (int)(e % q.K); n = (int)(e / q.K); }
    const float w = q.src[(long)k * q.sk + (long)n * q.sn];
    const _Float16 hf = (_Float16)w;
    const _Float16 mf = (_Float16)(w - (float)hf);
    const int kc = k >> 4, cc = k & 15;
    unsigned short* d = out + q.dst + (((long)kc * 2) * q.N + n) * 16 + cc;
    d[0] = __builtin_bit_cast(unsigned short, hf);
    d[(long)q.N * 16] = __builtin_bit_cast(unsigned short, mf);
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

// statistics from `nparts` partial rows [2][C * ntaps]; coef (with gamma, beta): the table the consumers of Z evaluate
static void bn_finalize(const TrainCtx& cx, const float* part, int nparts, int C, float n_rows, const float* n_rows_dev,
                        float momentum, int unbiased, float* stats, float* mmean, float* mvar, int ntaps,
                        const float* gamma = nullptr, const float* beta = nullptr, float4* coef = nullptr) {
    // lanes per channel by the number of (partial row, tap) pairs a channel has: a whole workgroup from 128 pairs on, a wave
    // from 32 (PP_TRAIN_FIN_THR="a,b" for A/B measurements; round 4, the 20 launches of a step: thresholds 2048 / 512 ->
    // 154 us at B=32 and 102 us at B=2; 512 / 128 -> 112 / 90; 128 / 32 -> 114 / 87: the launch is a chain of dependent
    // load rounds, and more lanes per channel make it shorter even when most of them carry one pair)
    static long thr256 = -1, thr64 = -1;
    if (thr256 < 0) {
        thr256 = 128; thr64 = 32;
        if (const char* e = getenv("PP_TRAIN_FIN_THR")) { long a = 0, b = 0; if (sscanf(e, "%ld,%ld", &a, &b) == 2 && a > 0 && b > 0) { thr256 = a; thr64 = b; } }
    }
    if ((long)nparts * ntaps >= thr256)
        PP_LAUNCH("k_tr_bn_finalize", (k_tr_bn_finalize<256>), dim3(C), dim3(256), 0, cx.stream, part, nparts, C, n_rows,
                  n_rows_dev, momentum, unbiased, stats, mmean, mvar, ntaps, gamma, beta, coef);
    else if ((long)nparts * ntaps >= thr64)
        PP_LAUNCH("k_tr_bn_finalize", (k_tr_bn_finalize<64>), dim3((C + 3) / 4), dim3(256), 0, cx.stream, part, nparts, C, n_rows,
                  n_rows_dev, momentum, unbiased, stats, mmean, mvar, ntaps, gamma, beta, coef);
    else
        PP_LAUNCH("k_tr_bn_finalize", (k_tr_bn_finalize<16>), dim3((C + 15) / 16), dim3(256), 0, cx.stream, part, nparts, C, n_rows,
                  n_rows_dev, momentum, unbiased, stats, mmean, mvar, ntaps, gamma, beta, coef);
}

// nparts partial rows of [2][C] (row stride pstride floats) -> sums[2][C] (and, when given, row 0 -> dup0[C], row 1 -> dup1[C])
void col_reduce(const TrainCtx& cx, int C, float* sums, float* dup0 = nullptr, float* dup1 = nullptr,
                const float* part = nullptr, long pstride = 0, int nparts = 0) {
    const long n = (long)2 * C;
    if (part == nullptr) { part = cx.part; pstride = n; nparts = TR_NPART; }
    if (nparts >= 1024)
        PP_LAUNCH("k_tr_reduce", k_tr_reduce_cols_wg, dim3((unsigned)n), dim3(256), 0, cx.stream, part,
                  nparts, n, pstride, sums, 0L, 0, 0, 1.0f, dup0, dup1, C);
    else
        PP_LAUNCH("k_tr_reduce", k_tr_reduce_cols, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, cx.stream, part,
                  nparts, n, pstride, sums, 0L, 0, 0, 1.0f, dup0, dup1, C);
}

// BatchNorm (training) statistics of Z[rows][C] -> stats + coefficient table, moving stats updated; with A != NULL the
// activation relu(bn(Z)) is written too (mapped rows) -- only where somebody reads it as a tensor (the block-final
// layers and the transposed convolutions); the in-block consumers evaluate it from Z and the table
// stat_tiles > 0: the product that wrote Z left per-row-tile column sums in cx.stat_part ([stat_tiles][2][C * ntaps])
void bn_relu_forward(const TrainCtx& cx, const float* Z, long rows, int C, const float* gamma, const float* beta,
                     float* stats, float4* coef, float* mmean, float* mvar, float momentum, float* A, int ld, int co_off,
                     RowMap rm, int stat_tiles = 0, int ntaps = 1) {
    if (stat_tiles > 0) {
        bn_finalize(cx, cx.stat_part, stat_tiles, C, (float)rows, nullptr, momentum, 1, stats, mmean, mvar, ntaps, gamma, beta, coef);
    } else {
        PP_LAUNCH("k_tr_colstats", k_tr_colstats, dim3(TR_NPART), dim3(256), 0, cx.stream, Z, rows, C, cx.part);
        bn_finalize(cx, cx.part, TR_NPART, C, (float)rows, nullptr, momentum, 1, stats, mmean, mvar, 1, gamma, beta, coef);
    }
    if (A != nullptr)
        PP_LAUNCH("k_tr_bn_relu", k_tr_bn_relu, dim3(blocks_for(rows * (C / 4))), dim3(256), 0, cx.stream, Z, rows, C,
                  (const float4*)coef, A, ld, co_off, rm);
}

// backward of the same: dA (mapped rows) -> dZ[rows][C]; d gamma, d beta written to the gradient buffer
// sums_part != NULL: the sums of g and g * zhat already lie in TR_NPART partial rows (left by k_tr_dw_bwd)
void bn_relu_backward(const TrainCtx& cx, const float* dA, int ld, int co_off, RowMap rm, const float* Z, long rows, int C,
                      const float4* coef, float* sums, float* dgamma, float* dbeta, float* dZ,
                      const float* sums_part = nullptr, long sums_pstride = 0, int sums_nparts = 0) {
    if (sums_part == nullptr) {
        PP_LAUNCH("k_tr_bn_bwd_reduce", k_tr_bn_bwd_reduce, dim3(TR_NPART), dim3(256), 0, cx.stream, dA, ld, co_off, rm, Z, rows,
                  C, coef, cx.part);
        col_reduce(cx, C, sums, dbeta, dgamma);
    } else {
        col_reduce(cx, C, sums, dbeta, dgamma, sums_part, sums_pstride, sums_nparts);
    }
    PP_LAUNCH("k_tr_bn_bwd_apply", k_tr_bn_bwd_apply, dim3(blocks_for(rows * (C / 4))), dim3(256), 0, cx.stream, dA, ld, co_off, rm,
              Z, rows, C, coef, (const float*)sums, 1.0f / (float)rows, dZ);
}

// n_rows[0] = (sum of the frames' pillar counts) * T: the rows of the reference's padded [P, T, C] tensor
// ... and prefix[b] = pillars of the frames before b (prefix[batch] = all of them)
__global__ void k_tr_pfn_rows(const int* __restrict__ npillars, int batch, int T, float* __restrict__ n_rows,
                              int* __restrict__ prefix) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int tot = 0;
        for (int b = 0; b < batch; ++b) { prefix[b] = tot; tot += npillars[b]; }
        prefix[batch] = tot;
        n_rows[0] = (float)tot * (float)T;
    }
}

// Persistent PFN grids: one resident round of workgroups (a quarter-filled second round costs a whole one).  Resident
// 4-wave workgroups per CU follow the kernels' register budgets: CPL 1 / 2 / 4 -> lin 8 / 7 / 4, max 7 / 5 / 4,
// bwd_reduce 8 / 6 / 4, bwd_apply 5 / 3 / 2.
static int pfn_blocks(int per_cu) { return std::min(TR_NPART, per_cu * 256); }

template <int CPL>
void pfn_forward(const TrainCtx& cx, const PfnT& p, const Lookup& L) {
    PP_LAUNCH("k_tr_pfn_rows", k_tr_pfn_rows, dim3(1), dim3(64), 0, cx.stream, p.npillars, p.batch, p.T, cx.pfn_nrows,
              cx.pfn_prefix);
    const int nlin = pfn_blocks(CPL == 4 ? 4 : 7), nmax = pfn_blocks(CPL == 4 ? 4 : 5);
    PP_LAUNCH("k_tr_pfn_lin", (k_tr_pfn_lin<CPL>), dim3(nlin), dim3(256), 0, cx.stream, p, cx.part);
    bn_finalize(cx, cx.part, nlin, p.C, 0.f, cx.pfn_nrows, 0.01f, 0, cx.pfn_stats, L.s("pfn/bn/moving_mean"),
                L.s("pfn/bn/moving_variance"), 1);
    PP_LAUNCH("k_tr_pfn_max", (k_tr_pfn_max<CPL>), dim3(nmax), dim3(256), 0, cx.stream, p,
              (const float*)cx.pfn_stats, L.p("pfn/bn/gamma"), L.p("pfn/bn/beta"), cx.pfn_feat, cx.pfn_arg);
}

template <int CPL>
void pfn_backward(const TrainCtx& cx, const PfnT& p, const Lookup& L, const float* dcanvas) {
    // (register budgets with the pillar pipeline: bwd_reduce 54 / 79 / 115 VGPRs, bwd_apply 95 / 138 / 214)
    const int nred = pfn_blocks(CPL == 4 ? 4 : 6), nblk = pfn_blocks(CPL == 4 ? 2 : (CPL == 2 ? 3 : 5));
    PP_LAUNCH("k_tr_pfn_bwd_reduce", (k_tr_pfn_bwd_reduce<CPL>), dim3(nred), dim3(256), 0, cx.stream, p,
              (const float*)cx.pfn_stats, (const int*)cx.pfn_arg, dcanvas, cx.part);
    col_reduce(cx, p.C, cx.pfn_sums, L.g("pfn/bn/beta"), L.g("pfn/bn/gamma"), cx.part, (long)2 * p.C, nred);
    PP_LAUNCH("k_tr_pfn_bwd_apply", (k_tr_pfn_bwd_apply<CPL>), dim3(nblk), dim3(256), 0, cx.stream, p,
              (const float*)cx.pfn_stats, L.p("pfn/bn/gamma"), (const int*)cx.pfn_arg, dcanvas, (const float*)cx.pfn_sums,
              (const float*)cx.pfn_nrows, cx.part);
    const long n = (long)p.FA * p.C;
    tr_reduce(cx.stream, (const float*)cx.part, nblk, n, n,
              L.g("pfn/dense/kernel"), 0L, 0, 0, 1.0f);
}

}  // namespace

size_t train_part_floats(const TrainShape& s) {
    size_t m = (size_t)2 * 512;
    m = std::max(m, (size_t)10 * s.C);
    for (const LayerDesc& L : s.layers) m = std::max(m, (size_t)11 * std::max(L.cin, L.cout));
    return (size_t)TR_NPART_MAX * m;
}

int train_step(const TrainCtx& cx, const TrainShape& s, const std::vector<TrainEntry>& layout, const float* params,
               float* grads, float* state, int batch, const LossParams& loss_in, int phase) {
    // phase bit 1: forward (PFN .. head maps); bit 2: loss + backward.  pp_train_step enqueues the two halves one after
    // the other with a wait for the target upload in between (the labels / regression targets travel on the copy
    // stream while the forward pass runs)
    Lookup L{layout, params, grads, state};
    const int B = batch;
    g_jobs.clear();
    g_arena_used = 0;
    {   // partial rows of the persistent reductions: ~128 rows of the largest map per workgroup, one per CU at least
        long want = ((long)B * s.ny * s.nx / 128 + 255) / 256 * 256;
        g_tr_npart = (int)std::min<long>(TR_NPART_MAX, std::max<long>(256, want));
    }
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
    p.npillars = cx.npillars; p.W = L.p("pfn/dense/kernel"); p.pprefix = cx.pfn_prefix; p.rec = cx.pfn_rec;
    const int cpl = (s.C + 63) / 64;
    const RowMap ident{1, 0, 0};
    const size_t HW = (size_t)s.head_h * s.head_w;
    const int nb = s.napl * 7, nc = s.napl * s.ncls, nd = s.use_dir ? s.napl * 2 : 0;
    const long px = (long)B * HW;
    if (phase & 1) {
    if (cpl == 1) pfn_forward<1>(cx, p, L);
    else if (cpl == 2) pfn_forward<2>(cx, p, L);
    else pfn_forward<4>(cx, p, L);
    const int ncanvas = s.nx * s.ny;
    PP_LAUNCH("k_tr_scatter", k_tr_scatter, dim3(blocks_for((long)B * ncanvas * (s.C / 4))), dim3(256), 0, cx.stream, cx.cellmap,
              (const float*)cx.pfn_feat, cx.canvas, B, s.nz, ncanvas, s.C, s.max_voxels);

    // fused forward of the separable layers (depthwise + product + statistics in one launch, launch_sep_train): from
    // PP_TRAIN_FUSED_MIN output pixels on (below that the layers are a handful of workgroups and the split-K product
    // kernels are the shorter chain); PP_TRAIN_FUSED=0 turns it off
    static int fused_on = -1;
    static long fused_min = -1;
    if (fused_on < 0) { const char* e = getenv("PP_TRAIN_FUSED"); fused_on = (e && e[0] == '0') ? 0 : 1; }
    if (fused_min < 0) { const char* e = getenv("PP_TRAIN_FUSED_MIN"); fused_min = e ? atol(e) : 32768; }
    bool fused = fused_on && cx.pw16 != nullptr;
    if (fused) {
        SplitPwTable t;
        memset(&t, 0, sizeof(t));
        int bi_ = 0, li_ = 0;
        for (size_t i = 0; i < s.layers.size() && fused; ++i) {
            const LayerDesc& l = s.layers[i];
            if (l.kind != LAYER_SEP && l.kind != LAYER_DECONV) continue;
            if (t.n >= 40) { fused = false; break; }
            if (l.kind == LAYER_SEP) {
                const std::string pre = "rpn/block" + std::to_string(bi_ + 1) + "/" + std::to_string(li_);
                t.job[t.n] = SplitPwJob{L.p(pre + "/pointwise_kernel"), cx.lbuf[i].pw16_off, l.cin, l.cout, (long)l.cout, 1L, t.total};
                t.total += (long)l.cin * l.cout;
                ++li_;
            } else {
                const int N = l.k * l.k * l.cout;
                t.job[t.n] = SplitPwJob{L.p("rpn/deconv" + std::to_string(bi_ + 1) + "/kernel"), cx.lbuf[i].pw16_off, l.cin, N, 1L, (long)l.cin, t.total};
                t.total += (long)l.cin * N;
                ++bi_; li_ = 0;
            }
            ++t.n;
        }
        if (fused && t.n > 0)
            PP_LAUNCH("k_tr_split_pw", k_tr_split_pw, dim3(blocks_for(t.total)), dim3(256), 0, cx.stream, cx.pw16, t);
    }

    // cur: what the next layer reads -- a tensor (cur_coef == NULL: the canvas, a block-final activation) or the
    // pre-BatchNorm map of an in-block layer with its coefficient table (the activation is evaluated by the reader)
    const float* cur = cx.canvas;
    const float4* cur_coef = nullptr;
    int bi = 0, li = 0, co_off = 0;
    for (size_t i = 0; i < s.layers.size(); ++i) {
        const LayerDesc& l = s.layers[i];
        const TrainLayerBuf& tb = cx.lbuf[i];
        if (l.kind == LAYER_SEP) {
            const std::string pre = "rpn/block" + std::to_string(bi + 1) + "/" + std::to_string(li);
            const long rows = (long)B * l.out_h * l.out_w;
            int stat_rows = 0;
            if (fused && rows >= fused_min && ((rows + 127) / 128 + 8) * 2 * l.cout <= cx.stat_part_floats) {
                SepTrainArgs t;
                t.in = cur; t.coef = cur_coef; t.dw = L.p(pre + "/depthwise_kernel"); t.wt16 = cx.pw16 + tb.pw16_off;
                t.Z = tb.Z; t.D = tb.D; t.stat = cx.stat_part;
                t.batch = B; t.in_h = l.in_h; t.in_w = l.in_w; t.cin = l.cin; t.out_h = l.out_h; t.out_w = l.out_w;
                t.cout = l.cout; t.stride = l.stride;
                static thread_local std::string tags[64];     // (the profiler keeps the pointer)
                std::string& tag = tags[i % 64];
                tag = "k_sep_u_tr:block" + std::to_string(bi + 1) + "." + std::to_string(li);
                t.tag = tag.c_str();
                stat_rows = launch_sep_train(t, cx.stream);
            }
            if (stat_rows > 0) {
                g_last_stat_tiles = stat_rows;
            } else {
            static int dwp = -1;      // PP_TRAIN_DWFWD=0: the thread-per-output kernel everywhere (A/B measurements)
            if (dwp < 0) { const char* e = getenv("PP_TRAIN_DWFWD"); dwp = (e && e[0] == '0') ? 0 : 1; }
            const long nthr = rows * (l.cin / 4);
            static long dwg = -1, dwm = -1;   // PP_TRAIN_DWFWD_GRID / _MINPX: workgroups of the persistent kernel, least pixels per thread
            if (dwg < 0) { const char* e = getenv("PP_TRAIN_DWFWD_GRID"); dwg = e ? atol(e) : 1024; }
            if (dwm < 0) { const char* e = getenv("PP_TRAIN_DWFWD_MINPX"); dwm = e ? atol(e) : 2; }     // (B=32 sweep, grid x pixels: 1024 x 4 0.275 ms, 1280 x 4 0.281, 1280 x 2 0.271, 1024 x 2 0.271, 1280 x 1 0.271, 2048 x 2 0.275; thread-per-output 0.331)
            if (dwp && nthr >= dwm * dwg * 256) {      // at least dwm pixels per thread on a one-round grid
                PP_LAUNCH("k_tr_dw_fwd", k_tr_dw_fwd_p, dim3((unsigned)dwg), dim3(256), 0, cx.stream, cur, L.p(pre + "/depthwise_kernel"),
                          tb.D, B, l.in_h, l.in_w, l.out_h, l.out_w, l.cin, l.stride, cur_coef);
            } else {
                PP_LAUNCH("k_tr_dw_fwd", k_tr_dw_fwd, dim3(blocks_for(nthr)), dim3(256), 0, cx.stream, cur,
                          L.p(pre + "/depthwise_kernel"), tb.D, (unsigned)nthr, l.in_h, l.in_w,
                          make_div((unsigned)l.out_h), make_div((unsigned)l.out_w), make_div((unsigned)(l.cin / 4)), l.cin,
                          l.stride, cur_coef);
            }
            gemm_tag("k_tr_gemm2:fwd.block" + std::to_string(bi + 1) + "." + std::to_string(li));
            tr_gemm(cx, tb.D, l.cin, 1, L.p(pre + "/pointwise_kernel"), l.cout, 1, tb.Z, l.cout, (int)rows, l.cout, l.cin,
                    nullptr, 0, 1, cx.stat_part);
            }
            // tb.A exists for the block-final layers only (the transposed convolution and the next block read it)
            bn_relu_forward(cx, tb.Z, rows, l.cout, L.p(pre + "/bn/gamma"), L.p(pre + "/bn/beta"), tb.stats, tb.coef,
                            L.s(pre + "/bn/moving_mean"), L.s(pre + "/bn/moving_variance"), 0.99f, tb.A, l.cout, 0, ident,
                            g_last_stat_tiles, 1);
            if (tb.A != nullptr) { cur = tb.A; cur_coef = nullptr; }
            else { cur = tb.Z; cur_coef = tb.coef; }
            ++li;
        } else if (l.kind == LAYER_DECONV) {
            const std::string pre = "rpn/deconv" + std::to_string(bi + 1);
            const long m = (long)B * l.in_h * l.in_w;
            const int N = l.k * l.k * l.cout;
            if (cur_coef != nullptr) return PP_ERR_UNSUPPORTED;   // (a block always ends in a layer that keeps its activation)
            // Zs[m][tap * cout + co] = X[m][:] . K[tap][co][:]   (Keras Conv2DTranspose kernel [k, k, Cout, Cin])
            int stat_rows = 0;
            // (a product, not a map walk: what has to be large is the number of workgroup tiles -- deconv3 at B=32 is
            // 10 240 rows x 2 048 columns = 1 280 tiles of 128 x 128)
            const long wg_tiles = ((m + 127) / 128) * (long)(N / ((N % 128 == 0) ? 128 : ((N % 64 == 0) ? 64 : 32)));
            if (fused && (m >= fused_min || (fused_min > 0 && wg_tiles >= 512)) &&
                ((m + 127) / 128 + 8) * 2 * (long)N <= cx.stat_part_floats) {
                static thread_local std::string tags[64];
                std::string& tag = tags[i % 64];
                tag = "k_sep_u_tr:deconv" + std::to_string(bi + 1);
                RowsTrainArgs t;
                t.in = cur; t.wt16 = cx.pw16 + tb.pw16_off; t.bias = nullptr; t.out = tb.Z; t.stat = cx.stat_part;
                t.rows = m; t.K = l.cin; t.N = N; t.ld_out = N; t.tag = tag.c_str();
                stat_rows = launch_rows_train(t, cx.stream);
            }
            if (stat_rows > 0) {
                g_last_stat_tiles = stat_rows;
            } else {
            gemm_tag("k_tr_gemm2:fwd.deconv" + std::to_string(bi + 1));
            tr_gemm(cx, cur, l.cin, 1, L.p(pre + "/kernel"), 1, l.cin, tb.Z, N, (int)m, N, l.cin, nullptr, 0, 1, cx.stat_part);
            }
            const RowMap rm{l.k, l.in_h, l.in_w};
            bn_relu_forward(cx, tb.Z, m * l.k * l.k, l.cout, L.p(pre + "/bn/gamma"), L.p(pre + "/bn/beta"), tb.stats, tb.coef,
                            L.s(pre + "/bn/moving_mean"), L.s(pre + "/bn/moving_variance"), 0.99f, cx.cat, s.CC, co_off, rm,
                            g_last_stat_tiles, l.k * l.k);
            co_off += l.cout;
            ++bi; li = 0;
        }
    }
    // heads: head[px][32] = cat[px][CC] . Wh[CC][32] + bias
    const float* kd = nd ? L.p("rpn/conv_dir_cls/kernel") : L.p("rpn/conv_box/kernel");
    const float* bd = nd ? L.p("rpn/conv_dir_cls/bias") : L.p("rpn/conv_box/bias");
    PP_LAUNCH("k_tr_pack_heads", k_tr_pack_heads, dim3(blocks_for((long)s.CC * PP_HEAD_COLS)), dim3(256), 0, cx.stream,
              L.p("rpn/conv_box/kernel"), L.p("rpn/conv_cls/kernel"), kd, L.p("rpn/conv_box/bias"), L.p("rpn/conv_cls/bias"), bd,
              s.CC, nb, nc, nd, cx.head_w, cx.head_b);
    int heads_done = 0;
    if (fused && px >= fused_min && cx.head_w16 != nullptr) {
        SplitPwTable t;
        memset(&t, 0, sizeof(t));
        t.n = 1; t.total = (long)s.CC * PP_HEAD_COLS;
        t.job[0] = SplitPwJob{cx.head_w, 0L, s.CC, PP_HEAD_COLS, (long)PP_HEAD_COLS, 1L, 0L};
        PP_LAUNCH("k_tr_split_pw", k_tr_split_pw, dim3(blocks_for(t.total)), dim3(256), 0, cx.stream, cx.head_w16, t);
        RowsTrainArgs r;
        r.in = cx.cat; r.wt16 = cx.head_w16; r.bias = cx.head_b; r.out = cx.head; r.stat = nullptr;
        r.rows = px; r.K = s.CC; r.N = PP_HEAD_COLS; r.ld_out = PP_HEAD_COLS; r.tag = "k_sep_u_tr:heads";
        heads_done = launch_rows_train(r, cx.stream);
    }
    if (!heads_done) {
    gemm_tag("k_tr_gemm2:fwd.heads");
    tr_gemm(cx, cx.cat, s.CC, 1, cx.head_w, PP_HEAD_COLS, 1, cx.head, PP_HEAD_COLS, (int)px, PP_HEAD_COLS, s.CC, cx.head_b, 0, 1);
    }
    }   // forward
    if (!(phase & 2)) return PP_OK;

    // ---------------- loss + gradient at the head maps ----------------
    LossParams lp = loss_in;
    lp.head = cx.head;
    lp.head_grad = cx.dhead;
    int st = launch_head_loss(lp, cx.stream);
    if (st) return st;

    // ---------------- backward ----------------
    // heads: dWh = cat^T . dhead, dbias = column sums, dcat = dhead . Wh^T
    gemm_tag("k_tr_gemm2:wgrad.heads");
    tr_gemm(cx, cx.cat, 1, s.CC, cx.dhead, PP_HEAD_COLS, 1, cx.dhead_w, PP_HEAD_COLS, s.CC, PP_HEAD_COLS, (int)px, nullptr, 0,
            wgrad_split(cx, s.CC, PP_HEAD_COLS, (int)px));      // (reduced at once: k_tr_unpack_head_grads reads it next)
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
    gemm_tag("k_tr_gemm2:dgrad.heads");
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
        bn_relu_backward(cx, cx.dcat, s.CC, cat_off[b], rm, db.Z, m * d.k * d.k, d.cout, db.coef, db.sums,
                         L.g(dpre + "/bn/gamma"), L.g(dpre + "/bn/beta"), cx.dZ);
        // dK[n][cin] = dZs^T . X     dX[m][cin] (+)= dZs . K
        float* dAct = cx.lbuf[last].dA;
        gemm_tag("k_tr_gemm2:pair.deconv" + std::to_string(b + 1));
        tr_gemm_pair(cx,
                     GemmCall{cx.dZ, 1, N, Xd, d.cin, 1, L.g(dpre + "/kernel"), d.cin, N, d.cin, (int)m, nullptr, 0,
                              wgrad_split(cx, N, d.cin, (int)m), nullptr, true},
                     GemmCall{cx.dZ, N, 1, L.p(dpre + "/kernel"), d.cin, 1, dAct, d.cin, (int)m, d.cin, N, nullptr,
                              (b + 1 < nblocks) ? 1 : 0, dgrad_split((int)m, d.cin, N), nullptr, false});
        // BatchNorm-backward sums of layer i left by the fused depthwise backward of layer i + 1 (NULL: not yet)
        const float* sums_part = nullptr;
        long sums_pstride = 0;
        int sums_nparts = 0;
        for (int i = last; i >= first_of_block[b]; --i) {
            const LayerDesc& l = s.layers[i];
            const TrainLayerBuf& tb = cx.lbuf[i];
            const std::string pre = "rpn/block" + std::to_string(b + 1) + "/" + std::to_string(i - first_of_block[b]);
            const long rows = (long)B * l.out_h * l.out_w;
            bn_relu_backward(cx, tb.dA, l.cout, 0, ident, tb.Z, rows, l.cout, tb.coef, tb.sums, L.g(pre + "/bn/gamma"),
                             L.g(pre + "/bn/beta"), cx.dZ, sums_part, sums_pstride, sums_nparts);
            sums_part = nullptr;
            // dWp[cin][cout] = D^T . dZ      dD[rows][cin] = dZ . Wp^T
            gemm_tag("k_tr_gemm2:pair.block" + std::to_string(b + 1) + "." + std::to_string(i - first_of_block[b]));
            tr_gemm_pair(cx,
                         GemmCall{tb.D, 1, l.cin, cx.dZ, l.cout, 1, L.g(pre + "/pointwise_kernel"), l.cout, l.cin, l.cout,
                                  (int)rows, nullptr, 0, wgrad_split(cx, l.cin, l.cout, (int)rows), nullptr, true},
                         GemmCall{cx.dZ, l.cout, 1, L.p(pre + "/pointwise_kernel"), 1, l.cout, cx.dD, l.cin, (int)rows, l.cin,
                                  l.cout, nullptr, 0, 1, nullptr, false});
            const bool in_block = i > first_of_block[b];          // the input is the layer before's BatchNorm + ReLU of Z
            if (in_block && l.stride == 1 && cx.lbuf[i - 1].A == nullptr) {
                // one pass: depthwise kernel gradient, input gradient, and the layer before's BatchNorm-backward sums;
                // partial rows [TR_NPART][11][cin] in a region of their own (the kernel-gradient rows are added by the
                // step's deferred-reduction launch, the two sum rows by the layer before's col_reduce)
                const TrainLayerBuf& pb = cx.lbuf[i - 1];
                const long n = (long)11 * l.cin;
                // at most one resident round of workgroups (179 VGPRs with the nine window loads in flight: two per CU);
                // 1 280 workgroups on 1 024 slots ran a quarter-filled second round
                const int nblk = std::min(TR_NPART, 512);
                // (arena exhausted: the partial rows go to the shared scratch and the kernel-gradient rows are added at
                // once, as in the k_tr_dw_bwd_w branch below; the two sum rows are read by the next col_reduce before
                // anything else writes the scratch)
                const bool room = g_arena_used + (long)nblk * n <= cx.gemm_part_floats;
                float* region = room ? cx.gemm_part + g_arena_used : cx.part;
                PP_LAUNCH("k_tr_dw_bwd", k_tr_dw_bwd, dim3(nblk), dim3(256), 0, cx.stream, (const float*)cx.dD,
                          L.p(pre + "/depthwise_kernel"), (const float*)pb.Z, (const float4*)pb.coef, pb.dA, region, B, l.in_h,
                          l.in_w, l.cin);
                if (room) {
                    g_jobs.push_back(ReduceJob{region, L.g(pre + "/depthwise_kernel"), (long)9 * l.cin, n, 0L, nblk, 0, 0, 1.0f});
                    g_arena_used += ((long)nblk * n + 63) / 64 * 64;
                } else {
                    tr_reduce(cx.stream, (const float*)region, nblk, (long)9 * l.cin, n, L.g(pre + "/depthwise_kernel"), 0L, 0, 0, 1.0f);
                }
                sums_part = region + (long)9 * l.cin;
                sums_pstride = n;
                sums_nparts = nblk;
                continue;
            }
            const float* X = (i == 0) ? cx.canvas : cx.lbuf[i - 1 - ((i == first_of_block[b] && b > 0) ? 1 : 0)].A;
            if (X == nullptr) return PP_ERR_UNSUPPORTED;
            {   // partial rows [TR_NPART][9][cin] in a region of their own, added by the step's deferred-reduction launch
                const long n = (long)9 * l.cin;
                const int nbw = std::min(TR_NPART, 1024);       // (98 VGPRs: four workgroups per CU -- one resident round)
                float* region = cx.part;
                const bool room = g_arena_used + (long)nbw * n <= cx.gemm_part_floats;
                if (room) region = cx.gemm_part + g_arena_used;
                PP_LAUNCH("k_tr_dw_bwd_w", k_tr_dw_bwd_w, dim3(nbw), dim3(256), 0, cx.stream, X, (const float*)cx.dD, region, B,
                          l.in_h, l.in_w, l.out_h, l.out_w, l.cin, l.stride);
                if (room) {
                    g_jobs.push_back(ReduceJob{region, L.g(pre + "/depthwise_kernel"), n, n, 0L, nbw, 0, 0, 1.0f});
                    g_arena_used += ((long)nbw * n + 63) / 64 * 64;
                } else {
                    tr_reduce(cx.stream, (const float*)cx.part, nbw, n, n, L.g(pre + "/depthwise_kernel"), 0L, 0, 0, 1.0f);
                }
            }
            // gradient of this layer's input: the previous layer's dA, the previous block's output gradient, or the canvas
            float* dX = (i == 0) ? cx.dcanvas : cx.lbuf[i - 1 - ((i == first_of_block[b] && b > 0) ? 1 : 0)].dA;
            PP_LAUNCH("k_tr_dw_bwd_in", k_tr_dw_bwd_in, dim3(blocks_for((long)B * l.in_h * l.in_w * (l.cin / 4))), dim3(256), 0,
                      cx.stream, (const float*)cx.dD, L.p(pre + "/depthwise_kernel"), dX,
                      (unsigned)((long)B * l.in_h * l.in_w * (l.cin / 4)), make_div((unsigned)l.in_h), make_div((unsigned)l.in_w),
                      l.out_h, l.out_w, make_div((unsigned)(l.cin / 4)), l.cin, l.stride, 0);
        }
    }
    // canvas -> pillar features -> PFN
    if (cpl == 1) pfn_backward<1>(cx, p, L, cx.dcanvas);
    else if (cpl == 2) pfn_backward<2>(cx, p, L, cx.dcanvas);
    else pfn_backward<4>(cx, p, L, cx.dcanvas);
    flush_deferred(cx);     // weight-gradient partial tiles + depthwise-gradient partial rows, one launch
    return PP_OK;
}
