// RPN backbone + SSD heads as fused float32 MFMA GEMM layers.
//
// Replaces RPN.call (reference model/voxelnet.py:695-717; layers :573-691):
//   SEP    ZeroPadding2D(1) + SeparableConv2D(3x3 depthwise [stride s, no bias] -> 1x1
//          pointwise [no bias]) + BatchNorm(eps 1e-3) + ReLU            one launch per layer
//   DECONV Conv2DTranspose(kernel == stride k, no bias) + BN + ReLU, written straight into
//          its channel slice of the concat buffer (tf.concat axis=3 never materialises)
//   HEAD   conv_box | conv_cls | conv_dir_cls (1x1 + bias) as ONE 384 -> 20 GEMM (N padded to 32)
// BN is folded into the pointwise / deconv weights and a bias at weight-finalise time.
//
// The reference computes in float32 and parity is 1e-4 on boxes/scores, so the
// GEMMs use the float32-input matrix instruction v_mfma_f32_32x32x2_f32 (exact
// f32 FMA chain; gfx950 has no TF32/xf32).  One workgroup (4 waves) owns 128
// consecutive output pixels (linear over batch*H*W, so no tile is wasted on odd
// map sizes) x NT output channels; K = Cin is walked in chunks of 32 channels:
//   stage A  the depthwise 3x3 for the chunk is computed by the VALU straight from
//            global/L2 (channels are independent, so chunking K is exact) and
//            written to LDS as the GEMM's A tile [128 px][32 ch (+4 pad)];
//   stage B  the matching [NT][32] slice of the pre-transposed weights -> LDS;
//   MFMA     wave w: rows w*32..+31; lane (r, h) feeds k = 16h + t for t = 0..15, so
//            both operands are four ds_read_b128 of contiguous K (row stride 36
//            floats is bank-conflict free for the 16-lane groups of ds_read_b128).
// The K order inside a chunk is permuted identically for A and B, which a dot
// product does not care about.  Depthwise and pointwise are never written to
// HBM in between.  Workgroup ids are remapped so that consecutive pixel tiles
// (which share halo rows) land on the same XCD / L2.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
// Split-precision operand pieces (PP_SPLIT_MODE, a build-time choice; pp_api.hip splits the weights the same way):
//   0: three bfloat16 pieces per float32 value, six products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid)
//   1: two float16 pieces (11 significand bits each), three products (hi*hi, hi*mid, mid*hi): the dropped
//      mid*mid term and the two-piece representation are each ~2^-22 relative
//   2: two float16 pieces, four products (+ mid*mid)
#if PP_SPLIT_MODE == 0
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define PC_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#else
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));   // (name kept: "one 8-element piece operand")
#define PC_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#endif
// third (lowest) piece of a pre-split operand in LDS: exists only in the three-piece mode
#if PP_SPLIT_MODE == 0
#define PC_LO(PTR) (*reinterpret_cast<const bf16x8*>(PTR))
#else
#define PC_LO(PTR) (bf16x8{})
#endif
// ACC += X * Y with X the first MFMA operand; smallest terms first
#if PP_SPLIT_MODE == 0
#define PC_PRODUCTS(ACC, XH, XM, XL, YH, YM, YL) \
    ACC = PC_MFMA(XL, YH, ACC); ACC = PC_MFMA(XH, YL, ACC); ACC = PC_MFMA(XM, YM, ACC); \
    ACC = PC_MFMA(XM, YH, ACC); ACC = PC_MFMA(XH, YM, ACC); ACC = PC_MFMA(XH, YH, ACC);
#elif PP_SPLIT_MODE == 1
#define PC_PRODUCTS(ACC, XH, XM, XL, YH, YM, YL) \
    ACC = PC_MFMA(XM, YH, ACC); ACC = PC_MFMA(XH, YM, ACC); ACC = PC_MFMA(XH, YH, ACC);
#else
#define PC_PRODUCTS(ACC, XH, XM, XL, YH, YM, YL) \
    ACC = PC_MFMA(XM, YM, ACC); ACC = PC_MFMA(XM, YH, ACC); ACC = PC_MFMA(XH, YM, ACC); ACC = PC_MFMA(XH, YH, ACC);
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer load: address = SRD base + voffset (VGPR, bytes) + soffset (SGPR, bytes).  One VGPR
// per address and no 64-bit vector address arithmetic (a plain pointer + varying uniform base made
// the compiler rebuild 64-bit addresses for every load of every K-chunk).
__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void buf_store16(float4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffe, 0x00020000);
}

#define PX_TILE 128

int g_num_cus = 256;   // set by pp_create from the device properties (persistent launches)

// GEMM arithmetic: split-precision bf16 MFMA (default) or the float32 MFMA.  PP_GEMM_PREC=f32 selects the
// latter; pp_bench_layer's ablation bits 2048 / 4096 force bf16x3 / f32.
static bool split_precision(int ablate) {
    static int dflt = -1;
    if (dflt < 0) {
        const char* e = getenv("PP_GEMM_PREC");
        dflt = (e && e[0] == 'f') ? 0 : 1;
    }
    if (ablate & 2048) return true;
    if (ablate & 4096) return false;
    return dflt == 1;
}
#define KC 32
#define LDS_STRIDE 36

struct GemmArgs {
    const float* in;
    const float* dw;
    const float* wt;
    const unsigned short* wt16;   // split weights: three bf16 pieces, [cin/16][3][n_total][16] (or NULL)
    int n_total;                  // rows of wt / wt16
    const float* bias;
    float* out;               // may be NULL when the layer's only consumer is the fused head GEMM
    float* head;              // fused head map [pixels][PP_HEAD_COLS]
    const float* head_wt;     // [PP_HEAD_COLS][cout] (deconv with fused heads)
    const float* head_wt16;   // the same slice as three bf16 pieces, [cout/16][3][PP_HEAD_COLS][16] 16-bit words
    const float* head_bias;   // [PP_HEAD_COLS]
    int head_mode;            // 0: none, 1: head = partial + bias, 2: head += partial
    int M;                // GEMM rows (pixels) < 2^31 (checked by the launcher)
    int tile_lo;          // k_sep_u: first 128-pixel tile of this launch (a launch over the frames [f0, f1) of a batch walks
                          // the tiles [f0 * hw / 128, ceil(f1 * hw / 128)) of the whole batch's pixel space, M = f1 * hw)
    int dbg;              // tuning aid: ablation bits (pp_bench_layer), 0 in production
    long long* stamps;    // tuning aid (dbg & 64): [block<64][role 2][iter 40][4] shader-clock stamps
    int in_h, in_w, cin;
    int px_h, px_w;       // pixel space of M (sep: output map; deconv/head: input map)
    int stride;
    int ld_out, co_off;
    int epi;              // 0: bias+ReLU rows; 1: deconv pixel-shuffle; 2: heads
    int k, cout;          // deconv
    const int* occ;       // sparse input (k_sep_u<..., OCC = 1> / k_sep_k4): cell -> pillar map, see LayerDesc::d_occ
    int occ_nz;
    const unsigned long long* occbits;   // the same occupancy as a bitmap (LayerDesc::d_occbits), or NULL: k_sep_u<..., OCC = 1>
    int occ_w64;                         // reads three 64-bit windows per pixel pair instead of 3 x WW x nz map entries
    // last fused-head branch only: the finished class logits of every pixel also go to a compact plane
    // [pixels][cls_ncol] (head columns cls_col0 .. cls_col0 + cls_ncol), which is all the post-process reads
    // of the head map for its candidate scan (8 bytes per pixel instead of a 128-byte row); NULL: not written
    float* cls_plane;
    int cls_col0, cls_ncol;
    // training-mode forward of a separable layer (k_sep_u<..., TR = 1>, launch_sep_train):
    const float4* tr_coef;   // [cin] (sc, sh, ., .): the input map is the PRE-BatchNorm map of the layer before and the
                             // activation relu(z * sc + sh) is evaluated on the way in; NULL: the input is a tensor
    float* tr_D;             // depthwise output [M][cin], kept for the weight-gradient product (or NULL)
    float* tr_stat;          // BatchNorm statistics partials [gridDim.x][2][n_total]: column sums / sums of squares
                             // of the output rows each workgroup produced
};

// D[row][col] of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// s_opix (LDS, deconv only): output pixel index of tap (0,0) for each of the tile's 128 input pixels.
template <int NTILES>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[NTILES], int p0, int n0, int wave,
                                              int lane, const int* s_opix) {
    const int h = lane >> 5, col_l = lane & 31;
    if (a.dbg & 4) return;
    if (a.epi == 0) {
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
            const int col = n0 + n * 32 + col_l;
            const float bv = a.bias[col];
            float* dst = a.out + a.co_off + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (pix < a.M) dst[(size_t)pix * a.ld_out] = relu_keep_nan(acc[n][r] + bv);
            }
        }
    } else if (a.epi == 1) {
        // Conv2DTranspose, kernel == stride: out[y*k+i][x*k+j][co] = sum_ci in[y][x][ci] * K[i][j][co][ci].
        // All integer divisions here are wave-uniform (one tap per 32-column tile).
        const int k = a.k, OW = a.px_w * k;
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
            const int col0 = __builtin_amdgcn_readfirstlane(n0 + n * 32);
            const int tap = col0 / a.cout, co = col0 - tap * a.cout + col_l;
            const int i = tap / k, j = tap - i * k;
            const int delta = i * OW + j;
            const float bv = a.bias[co];
            float* dst = a.out + a.co_off + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p0 + row < a.M) dst[(size_t)(s_opix[row] + delta) * a.ld_out] = relu_keep_nan(acc[n][r] + bv);
            }
        }
    } else {
        // heads: one 32-column row per pixel [box | cls | dir | zero pad]; bias, no activation
        const int col = n0 + col_l;
        const float bv = a.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (pix < a.M) a.head[(size_t)pix * PP_HEAD_COLS + col] = acc[0][r] + bv;
        }
    }
}

template <int NT, int MODE>
__global__ __launch_bounds__(256) void k_gemm_layer(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[PX_TILE * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float sB[NT * LDS_STRIDE];
    __shared__ int s_opix[PX_TILE];
    constexpr int NTILES = NT / 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware bijective remap of the pixel-tile index
    int mt;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        mt = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int p0 = mt * PX_TILE;
    const int n0 = blockIdx.y * NT;
    const int cin = a.cin;
    if (a.epi == 1 && tid < PX_TILE) {   // deconv: output pixel of tap (0,0) per input pixel (read after the K loop's barriers)
        const int pix = min(p0 + tid, a.M - 1);
        const int hw = a.px_h * a.px_w;
        const int b = pix / hw, rem = pix - b * hw;
        const int y = rem / a.px_w, x = rem - y * a.px_w;
        s_opix[tid] = (b * a.px_h * a.k + y * a.k) * (a.px_w * a.k) + x * a.k;
    }

    // this thread's 4 staging items: pixel (tid>>3) + 32*it, channel group tid&7
    const int c4 = tid & 7;
    const float* ibase[4];
    int yi0[4], xi0[4];
    bool pvalid[4];
    {
        const int hw = a.px_h * a.px_w;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pix = p0 + (tid >> 3) + 32 * it;
            pvalid[it] = pix < a.M;
            const int pc = pvalid[it] ? pix : 0;
            const int b = pc / hw;
            const int rem = pc - b * hw;
            const int yo = rem / a.px_w, xo = rem - yo * a.px_w;
            if (MODE == 0) {
                yi0[it] = yo * a.stride - 1;
                xi0[it] = xo * a.stride - 1;
                ibase[it] = a.in + (size_t)b * a.in_h * a.in_w * cin;
            } else {
                yi0[it] = 0; xi0[it] = 0;
                ibase[it] = a.in + (size_t)pc * cin;
            }
        }
    }

    f32x16 acc[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    for (int kc = 0; kc < cin; kc += KC) {
        __syncthreads();
        // ---- stage A ----
        if (MODE == 0) {
            float4 d[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) d[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ch = kc + c4 * 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const float4 wv = *reinterpret_cast<const float4*>(a.dw + (size_t)tap * cin + ch);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int yi = yi0[it] + dy, xi = xi0[it] + dx;
                    if (pvalid[it] && yi >= 0 && yi < a.in_h && xi >= 0 && xi < a.in_w) {
                        const float4 v = *reinterpret_cast<const float4*>(
                            ibase[it] + ((size_t)yi * a.in_w + xi) * cin + ch);
                        d[it].x = fmaf(v.x, wv.x, d[it].x);
                        d[it].y = fmaf(v.y, wv.y, d[it].y);
                        d[it].z = fmaf(v.z, wv.z, d[it].z);
                        d[it].w = fmaf(v.w, wv.w, d[it].w);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
                *reinterpret_cast<float4*>(&sA[((tid >> 3) + 32 * it) * LDS_STRIDE + c4 * 4]) = d[it];
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pvalid[it]) v = *reinterpret_cast<const float4*>(ibase[it] + kc + c4 * 4);
                *reinterpret_cast<float4*>(&sA[((tid >> 3) + 32 * it) * LDS_STRIDE + c4 * 4]) = v;
            }
        }
        // ---- stage B ----
#pragma unroll
        for (int r = 0; r < NT / 32; ++r) {
            const int e = tid + 256 * r;
            const int row = e >> 3, cb = e & 7;
            const float4 v = *reinterpret_cast<const float4*>(a.wt + (size_t)(n0 + row) * cin + kc + cb * 4);
            *reinterpret_cast<float4*>(&sB[row * LDS_STRIDE + cb * 4]) = v;
        }
        __syncthreads();
        // ---- MFMA ----
        const int h = lane >> 5, r32 = lane & 31;
        float4 a4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            a4[q] = *reinterpret_cast<const float4*>(&sA[(wave * 32 + r32) * LDS_STRIDE + h * 16 + q * 4]);
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
            float4 b4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                b4[q] = *reinterpret_cast<const float4*>(&sB[(n * 32 + r32) * LDS_STRIDE + h * 16 + q * 4]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].x, b4[q].x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].y, b4[q].y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].z, b4[q].z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].w, b4[q].w, acc[n], 0, 0, 0);
            }
        }
    }

    gemm_epilogue<NTILES>(a, acc, p0, n0, wave, lane, s_opix);
}


// ---------------------------------------------------------------------------------------
// Wave-specialised, software-pipelined variant (the fast path).
//
// 8 wavefronts per workgroup: waves 0-3 are CONSUMERS (one 32-pixel MFMA row block each,
// accumulators + fragments only), waves 4-7 are PRODUCERS (global loads, the depthwise 3x3
// on the VALU, LDS stores).  A workgroup's waves w and w+4 share a SIMD, so every SIMD hosts
// one MFMA wave and one VALU wave: the matrix pipe and the vector pipe run concurrently
// instead of taking turns inside one instruction stream.  LDS tiles are double-buffered:
// while the consumers run the MFMAs of K-chunk k out of buffer k&1, the producers turn the
// (already prefetched) registers of chunk k+1 into buffer (k+1)&1 and issue the global loads
// of chunk k+2; one workgroup barrier per K-chunk.  Register budgets of both roles are kept
// under 128 VGPRs so that two workgroups fit a CU and one's prologue / epilogue overlaps the
// other's MFMA phase.
//   MODE 0  separable layer (stride S = 1 or 2, output width even): K-chunks of 16
//           channels; a producer thread owns 2 consecutive output pixels of one row x 4
//           channels and slides a 3 x (S+3) input window (bounds tests once per row / column).
//   MODE 1  plain GEMM (Conv2DTranspose, heads): K-chunks of 32, producers copy A rows.
// The epilogue stages each consumer wave's 32 x NT tile through LDS and writes whole
// channel rows with 16-byte stores (full 128-B lines).  Odd-width separable layers use
// k_gemm_layer above.
template <int NT, int MODE, int S, int PXB>
__global__ __launch_bounds__(PXB * 4, (MODE == 0 && S == 2) ? 3 : 4) void k_gemm_ws(GemmArgs a) {
    constexpr int NCW = PXB / 32;                    // consumer waves == producer waves
    constexpr int NPT = NCW * 64;                    // producer (and consumer) threads
    constexpr int KCH = (MODE == 0) ? 16 : 32;      // channels per K-chunk
    constexpr int LSTR = KCH + 4;                    // LDS row stride (floats): conflict-free ds_read_b128
    constexpr int G = KCH / 4;                       // float4 channel groups per row
    constexpr int PXT = PXB * G / NPT;               // pixels per producer thread (2 or 4)
    constexpr int WW = (MODE == 0) ? (PXT - 1) * S + 3 : 1;   // input window width
    constexpr int NLD = (MODE == 0) ? 3 * WW : PXT;  // activation float4 loads per thread per chunk
    constexpr int SA = PXB * LSTR, SB = NT * LSTR;
    constexpr int NB4 = (NT * G + NPT - 1) / NPT;    // weight float4 per producer thread per chunk
    constexpr int NTILES = NT / 32;
    constexpr int KQ = KCH / 8;                      // float4 fragment reads per operand per chunk
    // one LDS arena: [sA buf0 | sA buf1 | sB buf0 | sB buf1]; the epilogue re-uses it as the
    // consumers' output staging area
    constexpr int OSTR = NT + 4;                     // output staging row stride (conflict-free b128 reads)
    constexpr int DWMAX = (MODE == 0) ? 9 * 256 : 0;   // depthwise taps [9][cin], cin <= 256 on this path
    constexpr int ARENA = (2 * SA + 2 * SB + DWMAX > PXB * OSTR) ? (2 * SA + 2 * SB + DWMAX) : (PXB * OSTR);
    __shared__ __attribute__((aligned(16))) float smem[ARENA];
    __shared__ int s_opix[PXB];
    float* const sA = smem;
    float* const sB = smem + 2 * SA;
    float* const sDW = smem + 2 * SA + 2 * SB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int mt;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        mt = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int p0 = mt * PXB;
    const int n0 = blockIdx.y * NT;
    const int cin = a.cin;
    const int nchunks = cin / KCH;
    const int dbg = a.dbg;

    if (tid < PXB) {   // output pixel index per tile row (deconv: pixel of tap (0,0))
        const int pix = min(p0 + tid, a.M - 1);
        int o = pix;
        if (a.epi == 1) {
            const int hw = a.px_h * a.px_w;
            const int b = pix / hw, rem = pix - b * hw;
            const int y = rem / a.px_w, x = rem - y * a.px_w;
            o = (b * a.px_h * a.k + y * a.k) * (a.px_w * a.k) + x * a.k;
        }
        s_opix[tid] = o;
    }

    if (MODE == 0) {   // depthwise taps of the whole layer -> LDS once (visible after the first barrier)
        for (int e = tid; e < 9 * cin / 4; e += 2 * NPT)
            reinterpret_cast<float4*>(sDW)[e] = reinterpret_cast<const float4*>(a.dw)[e];
    }

    if (wave >= NCW) {
        // =============================== producers ===============================
        const int pt = tid - NPT;
        const int c4 = pt % G, q = pt / G;            // channel group, pixel group
        const int pix0 = p0 + PXT * q;                // first of this thread's pixels (M % PXT == 0)
        const bool pvalid = pix0 < a.M;
        const int pc = pvalid ? pix0 : 0;
        // addressing: uniform base pointer + 32-bit BYTE offsets (saddr + voffset loads: one VGPR per
        // address).  Every activation buffer is allocated with a zeroed PP_ZPAD_FLOATS header in front
        // of its data (pp_api.hip); an out-of-map window element (zero padding of the convolution, or a
        // pixel past the end of the last tile) simply loads from that header, so no value is ever
        // selected after the load (a select on the LOADED value would serialise the loads, and masking
        // at use costs 4 v_cndmask per element).
        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(reinterpret_cast<const char*>(a.in) - PP_ZPAD_FLOATS * 4);
        const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(a.wt);
        unsigned cbase;
        unsigned okmask = 0;                          // bit dy*WW+dx: that window element is inside the map
        if (MODE == 0) {
            const int hw = a.px_h * a.px_w;
            const int b = pc / hw;
            const int rem = pc - b * hw;
            const int y = rem / a.px_w, x0 = rem - y * a.px_w;
            const int yi = y * S - 1, xi = x0 * S - 1;
            cbase = (unsigned)(((b * a.in_h + y * S) * a.in_w + x0 * S) * cin) * 4u + PP_ZPAD_FLOATS * 4u;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < WW; ++dx)
                    if (pvalid && yi + dy >= 0 && yi + dy < a.in_h && xi + dx >= 0 && xi + dx < a.in_w && !(dbg & 8))
                        okmask |= 1u << (dy * WW + dx);
        } else {
            cbase = (unsigned)(pc * cin) * 4u + PP_ZPAD_FLOATS * 4u;
            if (pvalid && !(dbg & 8)) okmask = 1u;
        }
        const int rs4 = a.in_w * cin * 4, cin4 = cin * 4;   // row / pixel stride in bytes (uniform)
        // per-thread byte offsets of the NLD window elements (incl. this lane's channel group), fixed for
        // the whole K loop: a K-chunk only moves the UNIFORM base pointer, so issuing a chunk's loads
        // costs no vector ALU work at all (the address is saddr + this VGPR)
        unsigned aoff[NLD];
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            unsigned off_;   // out-of-map -> the zero header
            if (MODE == 0) off_ = ((okmask >> e) & 1u) ? cbase + (unsigned)((e / WW - 1) * rs4 + (e % WW - 1) * cin4) : 0u;
            else off_ = okmask ? cbase + (unsigned)(e * cin4) : 0u;
            aoff[e] = off_ + (unsigned)(c4 * 16);
        }
        float4 rin[NLD];
        float4 rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;   // weight prefetch (named scalars: an array ended up in scratch)
        rb0 = rb1 = rb2 = rb3 = rb4 = rb5 = rb6 = rb7 = make_float4(0.f, 0.f, 0.f, 0.f);
        static_assert(NB4 <= 8, "weight prefetch registers");
        float* const dA = sA + (PXT * q) * LSTR + c4 * 4;
        unsigned boff[NB4 > 0 ? NB4 : 1];               // weight-tile byte offsets, fixed likewise
#pragma unroll
        for (int r = 0; r < NB4; ++r) {
            const int e_ = pt + NPT * r;
            boff[r] = (unsigned)(((n0 + e_ / G) * cin + (e_ % G) * 4) * 4);
        }

        // (macros, not lambdas: by-reference captures of rin / rb kept those arrays in scratch memory)
#define WS_LOAD_CHUNK(KCIDX)                                                                             \
        {                                                                                                \
            const unsigned so_ = (unsigned)(KCIDX) * (KCH * 4);   /* uniform: scalar offset operand */  \
            _Pragma("unroll") for (int e = 0; e < NLD; ++e) rin[e] = buf_load16(rs_in, aoff[e], so_);    \
            WS_LOAD_B(0, rb0) WS_LOAD_B(1, rb1) WS_LOAD_B(2, rb2) WS_LOAD_B(3, rb3)                      \
            WS_LOAD_B(4, rb4) WS_LOAD_B(5, rb5) WS_LOAD_B(6, rb6) WS_LOAD_B(7, rb7)                      \
        }
#define WS_LOAD_B(R, REG)                                                                                \
        if (NB4 > (R)) {                                                                                 \
            if ((NT * G) % NPT == 0 || pt + NPT * (R) < NT * G)                                          \
                REG = buf_load16(rs_wt, boff[(R) < NB4 ? (R) : 0], so_);                                 \
        }
#define WS_STORE_B(R, REG)                                                                               \
        if (NB4 > (R)) {                                                                                 \
            const int e_ = pt + NPT * (R);                                                               \
            if ((NT * G) % NPT == 0 || e_ < NT * G)                                                      \
                *reinterpret_cast<float4*>(sB + buf * SB + (e_ / G) * LSTR + (e_ % G) * 4) = REG;        \
        }
        // s = -1 is the pipeline prologue (nothing staged yet)
        {
            // loads of chunk 0 (always issued, from a clamped address when the element is outside the
            // map: a select on the LOADED value would make the compiler wait for each load in turn)
            WS_LOAD_CHUNK(0)
        }
        __syncthreads();   // (A) depthwise taps / s_opix visible; chunk-0 loads are already in flight
#ifdef PP_KERNEL_STAMPS   // diagnostic build (tools/layer_bench.py --ablate 64): in-kernel phase stamps
        const bool stamp = (dbg & 64) && a.stamps != nullptr && blockIdx.x < 64 && blockIdx.y == 0 && pt == 0;
        long long* st = a.stamps ? a.stamps + ((size_t)blockIdx.x * 2 + 1) * 40 * 4 : nullptr;
#else
        constexpr bool stamp = false;
        long long* st = nullptr;
#endif
        for (int s = -1; s < nchunks; ++s) {
            if (stamp && s + 1 < 40) st[(s + 1) * 4 + 0] = clock64();
#ifdef PP_KERNEL_STAMPS
            if ((dbg & 64) && a.stamps != nullptr && blockIdx.x < 64 && blockIdx.y == 0 && wave == NCW) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // isolate the load wait
                if (stamp && s + 1 < 40) st[(s + 1) * 4 + 1] = clock64();
            }
#endif
            if (s + 1 < nchunks) {
                // ---- stage chunk s+1 (registers -> LDS buffer (s+1)&1) ----
                const int kc = s + 1, buf = kc & 1;
                float* dst = dA + buf * SA;
                if (MODE == 0) {
                    const int ch = kc * KCH + c4 * 4;
                    float4 o[PXT];
#pragma unroll
                    for (int j = 0; j < PXT; ++j) o[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!(dbg & 2)) {
                        // all 9 tap vectors of this chunk first (one LDS latency, overlapping the tail of the
                        // global-load wait), then the FMAs
                        float4 wv[9];
#pragma unroll
                        for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(sDW + t * cin + ch);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) {
                                const float4 w4 = wv[dy * 3 + dx];
#pragma unroll
                                for (int j = 0; j < PXT; ++j) {
                                    const float4 v = rin[dy * WW + j * S + dx];
                                    o[j].x = fmaf(v.x, w4.x, o[j].x);
                                    o[j].y = fmaf(v.y, w4.y, o[j].y);
                                    o[j].z = fmaf(v.z, w4.z, o[j].z);
                                    o[j].w = fmaf(v.w, w4.w, o[j].w);
                                }
                            }
                    }
#pragma unroll
                    for (int j = 0; j < PXT; ++j) *reinterpret_cast<float4*>(dst + j * LSTR) = o[j];
                } else {
#pragma unroll
                    for (int j = 0; j < PXT; ++j) {
                        // element-wise rebuild: a whole-struct copy from rin[] becomes a private->LDS
                        // memcpy that keeps the array in scratch memory
                        const float4 t = make_float4(rin[j].x, rin[j].y, rin[j].z, rin[j].w);
                        *reinterpret_cast<float4*>(dst + j * LSTR) = t;
                    }
                }
                WS_STORE_B(0, rb0) WS_STORE_B(1, rb1) WS_STORE_B(2, rb2) WS_STORE_B(3, rb3)
                WS_STORE_B(4, rb4) WS_STORE_B(5, rb5) WS_STORE_B(6, rb6) WS_STORE_B(7, rb7)
                if (stamp && s + 1 < 40) st[(s + 1) * 4 + 2] = clock64();
                // ---- issue the loads of chunk s+2 ----
                if (s + 2 < nchunks) WS_LOAD_CHUNK(s + 2)
            }
            __syncthreads();
            if (stamp && s + 1 < 40) st[(s + 1) * 4 + 3] = clock64();
        }
        return;
#undef WS_LOAD_CHUNK
#undef WS_LOAD_B
#undef WS_STORE_B
    }

    // =============================== consumers ===============================
    f32x16 acc[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int h = lane >> 5, r32 = lane & 31;
    __syncthreads();   // (A)
#ifdef PP_KERNEL_STAMPS
    const bool stamp = (dbg & 64) && a.stamps != nullptr && blockIdx.x < 64 && blockIdx.y == 0 && tid == 0;
    long long* st = a.stamps ? a.stamps + ((size_t)blockIdx.x * 2 + 0) * 40 * 4 : nullptr;
#else
    constexpr bool stamp = false;
    long long* st = nullptr;
#endif
    if (stamp) st[0] = clock64();
    __syncthreads();   // chunk 0 staged
    if (stamp) st[3] = clock64();
    for (int k = 0; k < nchunks; ++k) {
        if (stamp && k + 1 < 40) st[(k + 1) * 4 + 0] = clock64();
        const float* cA = sA + (k & 1) * SA + (wave * 32 + r32) * LSTR + h * (KCH / 2);
        const float* cB = sB + (k & 1) * SB + r32 * LSTR + h * (KCH / 2);
        float4 a4[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) a4[q] = *reinterpret_cast<const float4*>(cA + q * 4);
        if (!(dbg & 1)) {
            if constexpr (NTILES * KQ <= 8) {
                // all B fragments of the chunk first (one LDS latency), then the MFMAs back to back
                float4 bq[NTILES][KQ];
#pragma unroll
                for (int n = 0; n < NTILES; ++n)
#pragma unroll
                    for (int q = 0; q < KQ; ++q) bq[n][q] = *reinterpret_cast<const float4*>(cB + n * 32 * LSTR + q * 4);
#pragma unroll
                for (int n = 0; n < NTILES; ++n)
#pragma unroll
                    for (int q = 0; q < KQ; ++q) {
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].x, bq[n][q].x, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].y, bq[n][q].y, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].z, bq[n][q].z, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].w, bq[n][q].w, acc[n], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int n = 0; n < NTILES; ++n) {
                    float4 b4[KQ];
#pragma unroll
                    for (int q = 0; q < KQ; ++q) b4[q] = *reinterpret_cast<const float4*>(cB + n * 32 * LSTR + q * 4);
#pragma unroll
                    for (int q = 0; q < KQ; ++q) {
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].x, b4[q].x, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].y, b4[q].y, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].z, b4[q].z, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].w, b4[q].w, acc[n], 0, 0, 0);
                    }
                }
            }
        }
        if (stamp && k + 1 < 40) { asm volatile("" :: "v"(acc[0][0])); st[(k + 1) * 4 + 2] = clock64(); }
        __syncthreads();
        if (stamp && k + 1 < 40) st[(k + 1) * 4 + 3] = clock64();
    }
    if (a.epi == 2) {
        gemm_epilogue<NTILES>(a, acc, p0, n0, wave, lane, s_opix);
        return;
    }
    // ---- coalesced epilogue: bias + ReLU -> this wave's LDS rows -> 16-byte stores of whole channel
    // rows (the final barrier of the K loop guarantees nobody still reads the arena) ----
    if (dbg & 4) return;
    float* so = smem + wave * 32 * OSTR;
    int cbase = n0, delta = 0;
    if (a.epi == 1) {
        const int tap = n0 / a.cout, i = tap / a.k;
        cbase = n0 - tap * a.cout;
        delta = i * (a.px_w * a.k) + (tap - i * a.k);
    }
#pragma unroll
    for (int n = 0; n < NTILES; ++n) {
        const float bvn = a.bias[cbase + n * 32 + r32];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            so[((r & 3) + 8 * (r >> 2) + 4 * h) * OSTR + n * 32 + r32] = relu_keep_nan(acc[n][r] + bvn);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int ROW4 = NT / 4;                 // float4 per staged row
    if (a.out != nullptr) {
        const int c4 = lane % ROW4;              // constant per lane: 64 % ROW4 == 0
        float* dst = a.out + a.co_off + cbase + c4 * 4;
        // batches of 4 rows-per-lane: all LDS reads first, then the stores (no per-store wait)
#pragma unroll
        for (int it0 = 0; it0 < NT / 8; it0 += 4) {
            float4 v[4];
            int op[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = ((it0 + u) * 64 + lane) / ROW4;
                v[u] = *reinterpret_cast<const float4*>(so + row * OSTR + c4 * 4);
                op[u] = s_opix[wave * 32 + row];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = ((it0 + u) * 64 + lane) / ROW4;
                if (p0 + wave * 32 + row < a.M)
                    *reinterpret_cast<float4*>(dst + (size_t)(op[u] + delta) * a.ld_out) = v[u];
            }
        }
    }
    if (a.head_mode != 0) {
        // ---- fused SSD heads: this tile's 32 x NT activated outputs (one deconv tap: NT == cout) times
        // the matching [NT x 32] slice of the head kernels, accumulated into the fused head map.  The
        // three deconv launches run in stream order and every (pixel, column) is touched by exactly one
        // workgroup per launch, so the read-modify-write needs no atomics and is bit-reproducible. ----
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
        const float* hA = so + r32 * OSTR + h * (NT / 2);
        const float* hB = a.head_wt + (size_t)r32 * NT + h * (NT / 2);
#pragma unroll
        for (int q = 0; q < NT / 8; ++q) {
            const float4 av = *reinterpret_cast<const float4*>(hA + q * 4);
            const float4 bw = *reinterpret_cast<const float4*>(hB + q * 4);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bw.x, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bw.y, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bw.z, hacc, 0, 0, 0);
            hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bw.w, hacc, 0, 0, 0);
        }
        const float hb = (a.head_mode == 1) ? a.head_bias[r32] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int prow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (p0 + prow < a.M) {
                float* hp = a.head + (size_t)(s_opix[prow] + delta) * PP_HEAD_COLS + r32;
                float v = hacc[r] + hb;
                if (a.head_mode == 2) v += *hp;
                *hp = v;
            }
        }
    }
    if (stamp) st[39 * 4 + 1] = clock64();
}


// 4 x 4 transpose inside each quad of lanes (registers x lanes), two DPP butterfly stages: afterwards
// lane (quad position i) holds, in r0..r3, what the quad's lanes 0..3 held in their register i.
__device__ __forceinline__ float dpp_quad(float v, const int ctrl_xor1) {
    const int x = __float_as_int(v);
    return __int_as_float(ctrl_xor1 ? __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true)
                                    : __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true));
}
#define QUAD_XCH(A, B, BIT, XOR1)                          \
    {                                                      \
        const float t_ = (BIT) ? (A) : (B);                \
        const float u_ = dpp_quad(t_, XOR1);               \
        (A) = (BIT) ? u_ : (A);                            \
        (B) = (BIT) ? (B) : u_;                            \
    }
__device__ __forceinline__ void quad_transpose4(float& r0, float& r1, float& r2, float& r3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    QUAD_XCH(r0, r1, b0, 1)
    QUAD_XCH(r2, r3, b0, 1)
    QUAD_XCH(r0, r2, b1, 0)
    QUAD_XCH(r1, r3, b1, 0)
}

// float32 -> three bfloat16 pieces hi + mid + lo (round-to-nearest-even each; the two remainders are
// exact float32 subtractions), 8 values at a time = one MFMA operand per piece.  a*b is then evaluated
// as the six products whose weight is >= 2^-16 relative (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid)
// on the bf16 matrix pipe with float32 accumulation: float32-equivalent accuracy (the dropped terms
// are < 2^-24 relative) at 6 x 32 instead of 8 x 64 matrix-pipe cycles per 16 channels.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#if PP_SPLIT_MODE == 0
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16x3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    // two values at a time: one v_cvt_pk_bf16_f32 per pair and piece; the bf16 -> f32 widening is a shift / a mask
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2_t x = {v[j], v[j + 1]};
        const bf16x2_t h = __builtin_convertvector(x, bf16x2_t);
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        // (scalar subtractions on purpose: packed fp32 adds are slow beside MFMAs)
        const f32x2_t r1 = {v[j] - __uint_as_float(hu << 16), v[j + 1] - __uint_as_float(hu & 0xffff0000u)};
        const bf16x2_t m = __builtin_convertvector(r1, bf16x2_t);
        const unsigned mu = __builtin_bit_cast(unsigned, m);
        const f32x2_t r2 = {r1.x - __uint_as_float(mu << 16), r1.y - __uint_as_float(mu & 0xffff0000u)};
        const bf16x2_t l = __builtin_convertvector(r2, bf16x2_t);
        hi[j] = h.x; hi[j + 1] = h.y;
        mid[j] = m.x; mid[j + 1] = m.y;
        lo[j] = l.x; lo[j + 1] = l.y;
    }
}
#else
// two float16 pieces: hi = rne(v), mid = rne(v - hi) (the subtraction is exact).  |v| must stay below 65504
// (BN-folded weights and BN-normalised activations are O(1)..O(100)); below ~0.1 the mid piece is a float16
// subnormal, i.e. the pair carries an ABSOLUTE error of at most 2^-25 instead of 2^-22 relative.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16x3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2_t x = {v[j], v[j + 1]};
        const f16x2_t h = __builtin_convertvector(x, f16x2_t);
        const f32x2_t r1 = {v[j] - (float)h.x, v[j + 1] - (float)h.y};
        const f16x2_t m = __builtin_convertvector(r1, f16x2_t);
        hi[j] = h.x; hi[j + 1] = h.y;
        mid[j] = m.x; mid[j + 1] = m.y;
    }
    lo = mid;   // unused by the two-piece product sets
}
#endif

// n / d and n % d for 0 <= n < 2^24 (exact in float32) with a precomputed reciprocal: a multiply, a
// truncation and one correction step each way instead of the ~40-instruction integer division
__device__ __forceinline__ void fast_divmod(int n, int d, float inv_d, int& q, int& r) {
    q = (int)((float)n * inv_d);
    r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

// ---------------------------------------------------------------------------------------
// Uniform-wave separable layer (every wave prepares AND multiplies).
//
// A 4-wave workgroup owns 128 output pixels x NT channels; wave w owns pixels 32w..32w+31 for the
// whole K loop.  Per 16-channel K-chunk a wave (1) turns its prefetched 3 x (S+3) input windows
// into the depthwise result of ITS OWN 32 pixels, written to a wave-private LDS A tile (no
// workgroup synchronisation: a wave's LDS operations execute in order), (2) stages its share of
// the weight chunk (the only shared operand), (3) issues the global loads of the chunk after
// next, (4) runs the 8 x NT/32 MFMAs of the current chunk; one workgroup barrier per chunk (for
// the weight tile).  Compared with the producer/consumer kernel above no wave slot is spent on
// waves that mostly wait for memory: with 3 workgroups per CU every SIMD holds three waves that
// all feed the matrix pipe, each hiding its own load latency behind the other two's MFMAs, and
// the scheduling grain is one wave per 32 pixels (tile quantisation costs less).  The epilogue
// stores straight from the accumulators: for a fixed accumulator register the 32 lanes of a
// half-wave hold 32 consecutive channels of one pixel = one full 128-byte line.
// PREC 0: float32 MFMA (v_mfma_f32_32x32x2_f32); PREC 1: split-precision bf16 MFMA (see split_bf16x3).
// OCC 1: the input is the sparse canvas (only cells that hold a pillar were written): every window position
// looks its cell up in the cell -> pillar map at tile start and reads the zero header when it is empty.
// TR 1: the TRAINING-mode forward of the layer (train.hip, model/voxelnet.py:576-660 with training=True): the input is
// the pre-BatchNorm map of the layer before, activated on the way in with that layer's batch-statistics coefficients
// (two more "tap" vectors per channel group: sc, sh; the padding comes from a NaN header -- v_max_f32(NaN, 0) = 0 -- so
// no window element is ever masked or selected); the depthwise output is also stored (the weight-gradient product's
// operand); the epilogue writes the raw product (this layer's pre-BatchNorm map) and the column sums / sums of squares
// of each wave's 32 rows for the batch statistics.  PREC 1 only.
template <int NT, int S, int WPS, int PREC, int OCC = 0, int TR = 0>
__global__ __launch_bounds__(256, WPS) void k_sep_u(GemmArgs a, int ntiles) {
    static_assert(TR == 0 || (PREC == 1 && OCC == 0), "training forward: split-precision, dense input");
    // TR 2: the same launch WITHOUT the depthwise -- a plain product rows x cin -> rows x n_total of the training forward
    // (transposed convolutions as GEMMs over their input pixels, the heads): the "window" is the pixel pair itself, the
    // A tile is staged unchanged, epilogue and statistics as TR 1 (+ bias when given).  Rows past M read as zeros
    // (out-of-range buffer offsets), so they add nothing to the statistics.
    constexpr bool PW = TR == 2;
    constexpr int NTAP = (TR == 1) ? 11 : 9;         // tap vectors per channel group in LDS (TR 1: + sc, sh)
    constexpr int KCH = 16, LSTR = KCH + 4, G = 4;
    constexpr int WW = S + 3;                        // input window width of 2 adjacent output pixels
    constexpr int NLD = PW ? 2 : 3 * WW;
    constexpr int SAW = 32 * LSTR;                   // one wave-private A buffer (floats)
    // weight tile in LDS: PREC 0 [NT][16 + 4] floats; PREC 1 [3 pieces][NT][16 bf16 = 8 floats], the two
    // 16-byte halves of a row swapped on odd groups of 8 rows (conflict-free ds_read_b128 without padding)
    constexpr int SB = (PREC == 0) ? NT * LSTR : PP_NPIECE * NT * 8;
    constexpr int NB4 = (PREC == 0) ? (NT * G + 255) / 256 : (NT * 2 * PP_NPIECE + 255) / 256;   // 16-byte weight items per thread per chunk
    constexpr int NTILES = NT / 32;
    constexpr int KQ = KCH / 8;
    // (TR: + the workgroup's running column sums / sums of squares, one [2][NT] row per wave)
    __shared__ __attribute__((aligned(16))) float smem[8 * SAW + 2 * SB + (PW ? 0 : NTAP * 256) + (TR ? 8 * NT : 0)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.dbg & 128) return;   // tuning aid: launch + dispatch cost only
#ifdef PP_KERNEL_STAMPS   // diagnostic build: wall-clock phase stamps of every workgroup (wave 0)
    long long* ust = ((a.dbg & 64) && a.stamps && tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096)
                         ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
    if (ust) {
        ust[0] = wall_clock64();
        ust[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        ust[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#define U_STAMP(i) { if (ust) ust[i] = wall_clock64(); }
#else
#define U_STAMP(i) {}
#endif
    float* const sAw = smem + wave * 2 * SAW;
    float* const sB = smem + 8 * SAW;
    float* const sDW = smem + 8 * SAW + 2 * SB;
    [[maybe_unused]] float* const sST = smem + 8 * SAW + 2 * SB + (PW ? 0 : NTAP * 256);

    // ---- this workgroup's tiles: XCD x (= blockIdx.x & 7, the dispatch order) owns the contiguous tile
    // range [x * ntiles / 8, (x + 1) * ntiles / 8); its workgroups walk that range together, so the halo
    // rows two neighbouring tiles share are served by one L2 ----
    const int xcd = blockIdx.x & 7, gl = blockIdx.x >> 3, GL = gridDim.x >> 3;
    const int nt_ = ntiles - a.tile_lo;                // tiles of this launch: [tile_lo, ntiles)
    const int tbase = a.tile_lo + (int)(((long long)xcd * nt_) >> 3), tend = a.tile_lo + (int)(((long long)(xcd + 1) * nt_) >> 3);
    const int first = tbase + gl;
    const int n0 = blockIdx.y * NT;
    if (first >= tend) {                               // uniform for the workgroup
        if (TR && a.tr_stat != nullptr && tid < 2 * NT)   // its (all-zero) row of the statistics partials
            a.tr_stat[((size_t)blockIdx.x * 2 + tid / NT) * a.n_total + n0 + tid % NT] = 0.f;
        return;
    }
    const int ntl = (tend - first + GL - 1) / GL;      // tiles of this workgroup
    const int cin = a.cin;
    const int nchunks = cin / KCH;
#ifdef PP_KERNEL_STAMPS
    const int dbg = a.dbg;                             // phase ablation bits (diagnostic build only)
#else
    constexpr int dbg = 0;                             // production: no ablation branches inside the K loop
#endif
    const int total = ntl * nchunks;                   // K-chunks in this workgroup's stream (>= 2: cin >= 32)

    // ---- staging role: lane (q, c4) owns output pixels pw + 2q, +1 and channels 4*c4..+3 of a chunk ----
    const int c4 = lane & 3, q = lane >> 2;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(reinterpret_cast<const char*>(a.in) - (PW ? 0 : PP_ZPAD_FLOATS * 4));
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(PREC == 0 ? (const void*)a.wt : (const void*)a.wt16);
    const int hw = a.px_h * a.px_w;
    const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)a.px_w;
    const int rs4 = a.in_w * cin * 4, cin4 = cin * 4;
    unsigned aoff[NLD];
    // TR: byte offset of this lane's first output pixel's depthwise row in tr_D (+ its 4 channels); beyond the
    // buffer's range when the pixel pair does not exist -- the buffer store is then dropped by the hardware
    [[maybe_unused]] unsigned doff = 0x80000000u;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_d = make_rsrc(TR ? (const void*)a.tr_D : (const void*)a.in);
    [[maybe_unused]] const bool d_out = TR && a.tr_D != nullptr && blockIdx.y == 0;
    // the input is a pre-BatchNorm map (NaN header) to be activated on the way in, or a tensor (zero header) read as it is
    [[maybe_unused]] const bool tr_act = TR && a.tr_coef != nullptr;
    // byte offsets of the 3 x WW window of this lane's pixel pair in tile `tile` (out-of-map -> zero header)
#define U_TILE_OFFSETS(TILE)                                                                             \
    {                                                                                                    \
        const int pix0_ = (TILE) * 128 + wave * 32 + 2 * q;                                              \
        const bool pvalid_ = pix0_ < a.M;                                                                \
        if (TR) doff = pvalid_ ? (unsigned)pix0_ * (unsigned)cin4 + (unsigned)(c4 * 16) : 0x80000000u;  /* M even */ \
        const int pc_ = pvalid_ ? pix0_ : 0;                                                             \
        if (PW) {   /* the pixel pair's own rows (row stride cin floats); a pair past the end reads zeros */ \
            aoff[0] = pvalid_ ? (unsigned)pc_ * (unsigned)cin4 + (unsigned)(c4 * 16) : 0x80000000u;      \
            aoff[NLD - 1] = pvalid_ ? aoff[0] + (unsigned)cin4 : 0x80000000u;                            \
        } else {                                                                                         \
        int b_, rem_, y_, x0_;                                                                           \
        fast_divmod(pc_, hw, inv_hw, b_, rem_);        /* pixel counts < 2^24 (checked by the launcher) */ \
        fast_divmod(rem_, a.px_w, inv_w, y_, x0_);                                                       \
        const int yi_ = y_ * S - 1, xi_ = x0_ * S - 1;                                                   \
        const unsigned cbase_ =                                                                          \
            (unsigned)(((b_ * a.in_h + y_ * S) * a.in_w + x0_ * S) * cin) * 4u + PP_ZPAD_FLOATS * 4u;    \
        /* row and column validity are separable; non-short-circuit '&' keeps this straight-line code   \
           (the '&&' form compiled to a nest of exec-mask branches per window element) */                \
        bool rowok_[3], colok_[WW];                                                                      \
        unsigned rowoff_[3], coloff_[WW];                                                                \
        _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_) {                                            \
            rowok_[dy_] = pvalid_ & ((unsigned)(yi_ + dy_) < (unsigned)a.in_h) & !(dbg & 8);             \
            rowoff_[dy_] = cbase_ + (unsigned)((dy_ - 1) * rs4);                                         \
        }                                                                                                \
        _Pragma("unroll") for (int dx_ = 0; dx_ < WW; ++dx_) {                                           \
            colok_[dx_] = (unsigned)(xi_ + dx_) < (unsigned)a.in_w;                                      \
            coloff_[dx_] = (unsigned)((dx_ - 1) * cin4);                                                 \
        }                                                                                                \
        /* OCC with the occupancy bitmap: bit (x + 1) of row y; this lane's window starts at bit xi_ + 1 >= 0 of   \
           rows yi_ .. yi_ + 2 -- two 64-bit words per row (the row ends in a spare word) instead of one map entry  \
           per window element and z-cell */                                                               \
        unsigned wbits_[3] = {0u, 0u, 0u};                                                               \
        const bool bits_ = OCC && a.occbits != nullptr;                /* uniform */                     \
        if (bits_) {                                                                                     \
            const unsigned long long* ob_ = a.occbits + (size_t)b_ * a.in_h * a.occ_w64;                 \
            const int pos0_ = xi_ + 1, wi_ = pos0_ >> 6, sh_ = pos0_ & 63;                               \
            _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_) {                                        \
                const int yy_ = min(max(yi_ + dy_, 0), a.in_h - 1);    /* (an out-of-map row is switched off by rowok_) */ \
                const unsigned long long* r_ = ob_ + (size_t)yy_ * a.occ_w64 + wi_;                      \
                const unsigned long long lo_ = r_[0], hi_ = r_[1];                                       \
                wbits_[dy_] = (unsigned)((lo_ >> sh_) | ((hi_ << 1) << (63 - sh_)));                     \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int e = 0; e < NLD; ++e) {                                                \
            const int dy_ = e / WW, dx_ = e % WW;                                                        \
            bool ok_ = rowok_[dy_] & colok_[dx_];                                                        \
            if (OCC) {                                                                                   \
                if (bits_) {                                                                             \
                    ok_ = ok_ & (((wbits_[dy_] >> dx_) & 1u) != 0u);                                     \
                } else {                                                                                 \
                    bool occ_ = false;                                                                   \
                    const int cidx_ = ok_ ? (yi_ + dy_) * a.in_w + xi_ + dx_ : 0;                        \
                    for (int z_ = 0; z_ < a.occ_nz; ++z_)                                                \
                        occ_ = occ_ | (a.occ[(size_t)(b_ * a.occ_nz + z_) * (a.in_h * a.in_w) + cidx_] >= 0); \
                    ok_ = ok_ & occ_;                                                                    \
                }                                                                                        \
            }                                                                                            \
            aoff[e] = (ok_ ? rowoff_[dy_] + coloff_[dx_] : 0u) + (unsigned)(c4 * 16);                    \
        }                                                                                                \
        }                                                                                                \
    }
    // weight staging items (16 bytes each): global byte offset and LDS float offset, fixed for the K loop
    constexpr int NBI = (PREC == 0) ? NT * G : NT * 2 * PP_NPIECE;   // items per chunk
    unsigned boff[NB4];
    int bdst[NB4];
#pragma unroll
    for (int r = 0; r < NB4; ++r) {
        const int e_ = (tid + 256 * r) % NBI;
        if (PREC == 0) {
            boff[r] = (unsigned)(((n0 + e_ / G) * cin + (e_ % G) * 4) * 4);
            bdst[r] = (e_ / G) * LSTR + (e_ % G) * 4;
        } else {
            const int piece = e_ / (NT * 2), rem = e_ % (NT * 2), row = rem >> 1, half = rem & 1;
            boff[r] = (unsigned)(((piece * a.n_total + n0 + row) * 16 + half * 8) * 2);
            bdst[r] = piece * (NT * 8) + row * 8 + ((half ^ ((row >> 3) & 1)) * 4);
        }
    }
    const unsigned bstep = (PREC == 0) ? (unsigned)(KCH * 4) : (unsigned)(PP_NPIECE * a.n_total * 32);   // bytes per K-chunk
    float4 rin[NLD];
    float4 rb0, rb1, rb2;
    rb0 = rb1 = rb2 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(NB4 <= 3, "weight prefetch registers");
#define U_LOAD_ACT(KCIDX, RIN)                                                                           \
    {                                                                                                    \
        const unsigned so_ = (unsigned)(KCIDX) * (KCH * 4);                                              \
        _Pragma("unroll") for (int e = 0; e < NLD; ++e) {                                                \
            /* diagnostic build, bit 256: the two outer window columns are not loaded (half the window   \
               traffic of a stride-1 layer; wrong results, timing only) */                               \
            if ((dbg & 256) && (e % WW == 0 || e % WW == WW - 1)) RIN[e] = make_float4(0.f, 0.f, 0.f, 0.f); \
            else RIN[e] = buf_load16(rs_in, aoff[e], so_);                                               \
        }                                                                                                \
    }
#define U_LOAD_WT(KCIDX)                                                                                 \
    {                                                                                                    \
        const unsigned sb_ = (unsigned)(KCIDX) * bstep;                                                  \
        if (NBI % 256 == 0 || NB4 > 1 || tid < NBI) rb0 = buf_load16(rs_wt, boff[0], sb_);               \
        if (NB4 > 1 && (NBI >= 512 || tid + 256 < NBI)) rb1 = buf_load16(rs_wt, boff[NB4 > 1 ? 1 : 0], sb_);  \
        if (NB4 > 2 && (NBI >= 768 || tid + 512 < NBI)) rb2 = buf_load16(rs_wt, boff[NB4 > 2 ? 2 : 0], sb_);  \
    }
#define U_LOAD_CHUNK(KCIDX) { U_LOAD_ACT(KCIDX, rin) U_LOAD_WT(KCIDX) }
    // three cursors walk the stream of (tile, chunk) positions: loads are issued two positions ahead of
    // the MFMAs, staging runs one ahead
    int ld_tile = first, ld_kc = 0;      // next position whose loads get issued
    int st_kc = 0;                       // chunk index of the next position to stage
    int mm_tile = first, mm_kc = 0;      // position the MFMAs work on
    U_TILE_OFFSETS(ld_tile)
    U_LOAD_CHUNK(0)
    ld_kc = 1;                           // nchunks >= 2 (cin >= 32, checked by the launcher)
    // depthwise taps -> LDS as [channel group of 4][9 taps][4 channels]: the 9 tap vectors of a lane's channel
    // group are 16 bytes apart (immediate offsets on one base address, which moves 576 bytes per K-chunk).
    // Staged AFTER the first window loads have been issued: the two memory round trips of the prologue overlap
    {
        const int ngrp = cin / 4;
        if (!PW) for (int e = tid; e < 9 * ngrp; e += 256) {
            const int t = e / ngrp, g4 = e - t * ngrp;
            reinterpret_cast<float4*>(sDW)[g4 * NTAP + t] = reinterpret_cast<const float4*>(a.dw)[e];
        }
        if (TR) for (int e = tid; e < 8 * NT; e += 256) sST[e] = 0.f;
        if (tr_act) {   // "taps" 9 and 10 of a channel group: the scale and the shift of the input's BatchNorm + ReLU
            for (int g4 = tid; g4 < ngrp; g4 += 256) {
                const float4 q0 = a.tr_coef[4 * g4], q1 = a.tr_coef[4 * g4 + 1], q2 = a.tr_coef[4 * g4 + 2], q3 = a.tr_coef[4 * g4 + 3];
                reinterpret_cast<float4*>(sDW)[g4 * NTAP + 9] = make_float4(q0.x, q1.x, q2.x, q3.x);
                reinterpret_cast<float4*>(sDW)[g4 * NTAP + 10] = make_float4(q0.y, q1.y, q2.y, q3.y);
            }
        }
    }

    f32x16 acc[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int h = lane >> 5, r32 = lane & 31;
    float bias_r[NTILES];   // loaded once: a global load inside the epilogue would sit on its critical path
#pragma unroll
    for (int n = 0; n < NTILES; ++n) bias_r[n] = (TR && a.bias == nullptr) ? 0.f : a.bias[n0 + n * 32 + r32];
    __syncthreads();   // depthwise taps visible
    U_STAMP(1)

    // ---- the parts of one stream position ----
    // MFMAs of position I out of buffer I & 1
#define U_MFMA(I)                                                                                        \
    if (!(dbg & 1) && PREC == 1) {                                                                       \
        const float* cA = cA0 + ((I) & 1) * SAW;                                                         \
        const float* cB = cB0 + ((I) & 1) * SB;                                                          \
        const float4 a0 = *reinterpret_cast<const float4*>(cA), a1 = *reinterpret_cast<const float4*>(cA + 4);  \
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};                            \
        bf16x8 ah, am, al;                                                                               \
        split_bf16x3(av, ah, am, al);                                                                    \
        _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                             \
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(cB + n * 32 * 8);                         \
            const bf16x8 bm = *reinterpret_cast<const bf16x8*>(cB + NT * 8 + n * 32 * 8);                \
            [[maybe_unused]] const bf16x8 bl = PC_LO(cB + 2 * NT * 8 + n * 32 * 8);            \
            PC_PRODUCTS(acc[n], ah, am, al, bh, bm, bl)                                                  \
        }                                                                                                \
    }                                                                                                    \
    if (!(dbg & 1) && PREC == 0) {                                                                       \
        const float* cA = sAw + ((I) & 1) * SAW + r32 * LSTR + h * (KCH / 2);                            \
        const float* cB = sB + ((I) & 1) * SB + r32 * LSTR + h * (KCH / 2);                              \
        float4 a4[KQ];                                                                                   \
        _Pragma("unroll") for (int qq = 0; qq < KQ; ++qq) a4[qq] = *reinterpret_cast<const float4*>(cA + qq * 4);  \
        _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                             \
            float4 b4[KQ];                                                                               \
            _Pragma("unroll") for (int qq = 0; qq < KQ; ++qq) b4[qq] = *reinterpret_cast<const float4*>(cB + n * 32 * LSTR + qq * 4);  \
            _Pragma("unroll") for (int qq = 0; qq < KQ; ++qq) {                                          \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qq].x, b4[qq].x, acc[n], 0, 0, 0);      \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qq].y, b4[qq].y, acc[n], 0, 0, 0);      \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qq].z, b4[qq].z, acc[n], 0, 0, 0);      \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qq].w, b4[qq].w, acc[n], 0, 0, 0);      \
            }                                                                                            \
        }                                                                                                \
    }
    // stage position P (buffer P & 1): depthwise of this wave's pixels -> private A tile; weight share -> sB
#define U_STAGE2(P, RIN)                                                                                 \
    {                                                                                                    \
        constexpr int buf = (P) & 1;   /* the steady loop is unrolled by two: buffer parities are static */ \
        float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;                                            \
        if (tr_act) {   /* relu(z * sc + sh) of every window element; a NaN (the padding header) becomes 0 */ \
            const float4 sc_ = *reinterpret_cast<const float4*>(twp + 36), sh_ = *reinterpret_cast<const float4*>(twp + 40);  \
            _Pragma("unroll") for (int e = 0; e < NLD; ++e) {                                            \
                RIN[e].x = fmaxf(fmaf(RIN[e].x, sc_.x, sh_.x), 0.f); RIN[e].y = fmaxf(fmaf(RIN[e].y, sc_.y, sh_.y), 0.f);  \
                RIN[e].z = fmaxf(fmaf(RIN[e].z, sc_.z, sh_.z), 0.f); RIN[e].w = fmaxf(fmaf(RIN[e].w, sc_.w, sh_.w), 0.f);  \
            }                                                                                            \
        }                                                                                                \
        if (PW) { o0 = RIN[0]; o1 = RIN[NLD - 1]; }                                                      \
        if (!PW && !(dbg & 2)) {                                                                         \
            const float* tw = twp;                                                                       \
            _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                             \
                _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                       \
                    const float4 w4 = *reinterpret_cast<const float4*>(tw + (dy * 3 + dx) * 4);          \
                    const float4 v0 = RIN[PW ? 0 : dy * WW + dx], v1 = RIN[PW ? 0 : dy * WW + S + dx];   \
                    o0.x = fmaf(v0.x, w4.x, o0.x); o0.y = fmaf(v0.y, w4.y, o0.y);                        \
                    o0.z = fmaf(v0.z, w4.z, o0.z); o0.w = fmaf(v0.w, w4.w, o0.w);                        \
                    o1.x = fmaf(v1.x, w4.x, o1.x); o1.y = fmaf(v1.y, w4.y, o1.y);                        \
                    o1.z = fmaf(v1.z, w4.z, o1.z); o1.w = fmaf(v1.w, w4.w, o1.w);                        \
                }                                                                                        \
        }                                                                                                \
        float* dA = sAw + buf * SAW + (2 * q) * LSTR + c4 * 4;                                           \
        *reinterpret_cast<float4*>(dA) = o0;                                                             \
        *reinterpret_cast<float4*>(dA + LSTR) = o1;                                                      \
        if (d_out) {   /* the depthwise output, kept for the weight-gradient product: [pixel][cin] */     \
            /* the chunk offset goes into the VECTOR offset, soffset stays the constant 0: a 16-byte buffer store \
               with an SGPR soffset was seen to pick up the NEXT value written to its data registers (the voffset \
               of the second store landed in D) -- the compiler only keeps the store-data wait state of gfx9 for \
               an immediate soffset */                                                                   \
            const unsigned dv_ = doff + (unsigned)st_kc * (KCH * 4);                                     \
            buf_store16(o0, rs_d, dv_, 0u);                                                              \
            buf_store16(o1, rs_d, dv_ + (unsigned)cin4, 0u);                                             \
        }                                                                                                \
        if (NBI % 256 == 0 || NB4 > 1 || tid < NBI) *reinterpret_cast<float4*>(sB + buf * SB + bdst[0]) = rb0;  \
        if (NB4 > 1 && (NBI >= 512 || tid + 256 < NBI)) *reinterpret_cast<float4*>(sB + buf * SB + bdst[NB4 > 1 ? 1 : 0]) = rb1;  \
        if (NB4 > 2 && (NBI >= 768 || tid + 512 < NBI)) *reinterpret_cast<float4*>(sB + buf * SB + bdst[NB4 > 2 ? 2 : 0]) = rb2;  \
        twp += 4 * (4 * NTAP);                                                                           \
        if (++st_kc == nchunks) { st_kc = 0; twp = sDW + c4 * (4 * NTAP); }                              \
    }
#define U_STAGE(P) U_STAGE2(P, rin)
    // global loads of the next position of the load cursor
#define U_ISSUE()                                                                                        \
    {                                                                                                    \
        if (ld_kc == 0) U_TILE_OFFSETS(ld_tile)                                                          \
        U_LOAD_CHUNK(ld_kc)                                                                              \
        if (++ld_kc == nchunks) { ld_kc = 0; ld_tile += GL; }                                            \
    }
    // tile finished: bias + ReLU, then a 4 x 4 register/lane transpose inside each lane quad turns "lane =
    // channel, register = pixel row" into "lane holds 4 consecutive channels of one pixel": 16-byte stores, 8
    // full 128-byte lines per instruction (few, large stores: the store path is bound by requests in flight,
    // not bytes).  Issued AFTER the iteration's prefetch loads: the memory counter retires in issue order, so
    // the next wait for those loads does not also wait for the stores' acknowledgements; the stores drain
    // while the next tile's chunks are multiplied.
    // (inference: folded bias + ReLU; training: the raw product is this layer's pre-BatchNorm map)
#define U_ACT(X) (TR ? (X) : relu_keep_nan(X))
#define U_EPILOGUE()                                                                                     \
    {                                                                                                    \
        const int pw = mm_tile * 128 + wave * 32;                                                        \
        if (TR) {   /* column sums of this lane's 16 rows (rows >= M are exact zeros), kept over the workgroup's tiles */ \
            _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                         \
                float s1 = 0.f, s2 = 0.f;                                                                \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) { const float z = acc[n][r]; s1 += z; s2 = fmaf(z, z, s2); }  \
                s1 += __shfl_xor(s1, 32);                                                                \
                s2 += __shfl_xor(s2, 32);                                                                \
                if (h == 0) {   /* wave-private LDS row: no synchronisation (a wave's LDS operations execute in order) */ \
                    sST[(wave * 2 + 0) * NT + n * 32 + r32] += s1;                                       \
                    sST[(wave * 2 + 1) * NT + n * 32 + r32] += s2;                                       \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        if (!(dbg & 4) && pw < a.M) {                                                                    \
            const int qi = lane & 3, qj = r32 >> 2;                                                      \
            float* dst = a.out + (size_t)(pw + 4 * h + qi) * a.ld_out + a.co_off + n0 + qj * 4;          \
            const bool full = pw + 32 <= a.M;             /* wave-uniform: one branch, not one per store */ \
            const int ldo = a.ld_out;                                                                    \
            if (full) {                                                                                  \
                _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                     \
                    const float bvn = bias_r[n];                                                         \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                      \
                        float x0 = U_ACT(acc[n][4 * g + 0] + bvn), x1 = U_ACT(acc[n][4 * g + 1] + bvn);          \
                        float x2 = U_ACT(acc[n][4 * g + 2] + bvn), x3 = U_ACT(acc[n][4 * g + 3] + bvn);          \
                        quad_transpose4(x0, x1, x2, x3, lane);                                           \
                        *reinterpret_cast<float4*>(dst + (8 * g) * ldo + n * 32) = make_float4(x0, x1, x2, x3);  \
                    }                                                                                    \
                }                                                                                        \
            } else {                                                                                     \
                _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                     \
                    const float bvn = bias_r[n];                                                         \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                      \
                        float x0 = U_ACT(acc[n][4 * g + 0] + bvn), x1 = U_ACT(acc[n][4 * g + 1] + bvn);          \
                        float x2 = U_ACT(acc[n][4 * g + 2] + bvn), x3 = U_ACT(acc[n][4 * g + 3] + bvn);          \
                        quad_transpose4(x0, x1, x2, x3, lane);                                           \
                        if (pw + 8 * g + 4 * h + qi < a.M)                                               \
                            *reinterpret_cast<float4*>(dst + (8 * g) * ldo + n * 32) = make_float4(x0, x1, x2, x3);  \
                    }                                                                                    \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int n = 0; n < NTILES; ++n)                                               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;                              \
        mm_kc = 0;                                                                                       \
        mm_tile += GL;                                                                                   \
    }

    const float* twp = sDW + c4 * (4 * NTAP);          // this lane's tap vectors of the next chunk to stage
    const float* const cA0 = sAw + r32 * LSTR + h * (KCH / 2);                       // fragment read bases (PREC 1)
    const float* const cB0 = sB + r32 * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
    // prologue: stage position 0, issue the loads of position 1 (total >= 2)
    U_STAGE(0)
    U_ISSUE()
    __syncthreads();
    U_STAMP(2)
    // steady state: positions i (MFMA), i+1 (staging) and i+2 (loads) all exist -- no branch between the
    // MFMA block and the staging block, so the scheduler may interleave matrix and vector work
    // (unrolled by two -- total is even -- so that the LDS buffer of every access is a compile-time constant)
    int i = 0;
    for (; i + 3 < total; i += 2) {
#ifdef PP_KERNEL_STAMPS
        if (ust && blockIdx.x < 64 && i + 1 < 64) a.stamps[4096 * 8 + blockIdx.x * 64 + i + 1] = wall_clock64();
#endif
#ifdef PP_KERNEL_STAMPS   // phase boundaries of ONE iteration (i == 4), shader clock: tools/phase_stamps.py
#define U_PH(k) { if (ust && blockIdx.x < 64 && i == 4) a.stamps[4096 * 8 + blockIdx.x * 64 + 40 + (k)] = clock64(); }
#else
#define U_PH(k) {}
#endif
        U_PH(0)
        U_MFMA(0)
        U_PH(1)
        U_STAGE(1)
        U_PH(2)
        U_ISSUE()
        U_PH(3)
        if (++mm_kc == nchunks) U_EPILOGUE()
        __syncthreads();
        U_PH(4)
        U_MFMA(1)
        U_STAGE(0)
        U_ISSUE()
        if (++mm_kc == nchunks) U_EPILOGUE()
        __syncthreads();
    }
    // last two positions (i even): nothing left to load, then nothing left to stage
    U_MFMA(0)
    U_STAGE(1)
    if (++mm_kc == nchunks) U_EPILOGUE()
    __syncthreads();
    U_MFMA(1)
    ++mm_kc;
    U_EPILOGUE()
    if (TR && a.tr_stat != nullptr) {   // one statistics row per workgroup: [gridDim.x][2][n_total]; the four waves' sums added in a fixed order
        __syncthreads();                               // every wave's sums are in its row
        const float* red = sST;                        // [4 waves][2][NT]
        if (tid < 2 * NT) {
            const int which = tid / NT, col = tid % NT;
            const float v = ((red[(0 * 2 + which) * NT + col] + red[(1 * 2 + which) * NT + col]) +
                             red[(2 * 2 + which) * NT + col]) + red[(3 * 2 + which) * NT + col];
            a.tr_stat[((size_t)blockIdx.x * 2 + which) * a.n_total + n0 + col] = v;
        }
    }
#undef U_MFMA
#undef U_STAGE
#undef U_ISSUE
#undef U_PH
#undef U_EPILOGUE
#undef U_ACT
#undef U_LOAD_CHUNK
#undef U_TILE_OFFSETS
    U_STAMP(3)
#ifdef PP_KERNEL_STAMPS
    if (a.dbg & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    U_STAMP(4)
#endif
}

// persistent launch: WPS workgroups per CU (one wave per SIMD each), a multiple of 8 so that every XCD
// gets the same number
// WPS / WPB: workgroups per CU of the float32 / split-bf16 instantiation (register budgets differ)
template <int NT, int S, int WPS, int WPB>
static void launch_u(const GemmArgs& a, int n_total, hipStream_t s) {
    const int ntiles = (a.M + 127) / 128;
    const int ny = n_total / NT;
    const bool bf = a.wt16 != nullptr && split_precision(a.dbg);
    int slots = (g_num_cus * (bf ? WPB : WPS)) / ny;
    const int mine = ntiles - a.tile_lo;               // (a.tile_lo > 0: a launch over a sub-range of the batch's frames)
    int gx = mine < slots ? mine : slots;
    gx = (gx + 7) & ~7;
    dim3 grid((unsigned)gx, ny);
    if (bf && a.occ != nullptr)
        PP_LAUNCH("k_sep_u", (k_sep_u<NT, S, WPB, 1, 1>), grid, dim3(256), 0, s, a, ntiles);
    else if (bf)
        PP_LAUNCH("k_sep_u", (k_sep_u<NT, S, WPB, 1>), grid, dim3(256), 0, s, a, ntiles);
    else
        PP_LAUNCH("k_sep_u", (k_sep_u<NT, S, WPS, 0>), grid, dim3(256), 0, s, a, ntiles);
}

// Training-mode forward of one separable layer as ONE launch (SepTrainArgs, pp_common.h): depthwise (with the input's
// BatchNorm + ReLU evaluated on the way in) -> A tile in LDS -> pointwise product with this step's weights as two
// float16 pieces -> pre-BatchNorm map + statistics partials; the depthwise output is stored for the backward pass.
// Returns the number of statistics rows written ([rows][2][cout]), or 0 when the shape is not one the kernel takes
// (the caller then runs the separate depthwise and product kernels).
int launch_sep_train(const SepTrainArgs& t, hipStream_t s) {
#if PP_SPLIT_MODE != 1
    (void)t; (void)s;
    return 0;
#else
    const long long M = (long long)t.batch * t.out_h * t.out_w;
    const long long in_bytes = (long long)t.batch * t.in_h * t.in_w * t.cin * 4;
    if (t.cin % KC != 0 || t.cout % 32 != 0 || (t.stride != 1 && t.stride != 2) || t.out_w % 2 != 0 || M >= (1 << 24) ||
        in_bytes >= (1ll << 31) - 4096 || M * t.cin * 4 >= (1ll << 31) - 4096 || t.cin > 256)
        return 0;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = t.in; a.dw = t.dw; a.wt16 = t.wt16; a.n_total = t.cout; a.out = t.Z;
    a.M = (int)M; a.in_h = t.in_h; a.in_w = t.in_w; a.cin = t.cin; a.px_h = t.out_h; a.px_w = t.out_w;
    a.stride = t.stride; a.ld_out = t.cout; a.co_off = 0; a.cout = t.cout;
    a.tr_coef = t.coef; a.tr_D = t.D; a.tr_stat = t.stat;
    const int ntiles = (a.M + 127) / 128;
    const int nt = (t.cout % 128 == 0) ? 128 : (t.cout % 64 == 0 ? 64 : 32);
    const int ny = t.cout / nt;
    const int wpb = (nt == 128) ? 2 : 3;
    int slots = (g_num_cus * wpb) / ny;
    int gx = ntiles < slots ? ntiles : slots;
    gx = (gx + 7) & ~7;
    const dim3 grid((unsigned)gx, ny);
    const char* tg = t.tag ? t.tag : "k_sep_u_tr";
    {   // measurement switches (wrong results): PP_TR_NOD=1 no depthwise-output store, PP_TR_NOACT=1 no activation on the way in
        static int nod = -1, noact = -1;
        if (nod < 0) { const char* e = getenv("PP_TR_NOD"); nod = (e && e[0] == '1') ? 1 : 0; }
        if (noact < 0) { const char* e = getenv("PP_TR_NOACT"); noact = (e && e[0] == '1') ? 1 : 0; }
        if (nod) a.tr_D = nullptr;
        if (noact) a.tr_coef = nullptr;
    }
    if (t.stride == 1) {
        if (nt == 128) PP_LAUNCH(tg, (k_sep_u<128, 1, 2, 1, 0, 1>), grid, dim3(256), 0, s, a, ntiles);
        else if (nt == 64) PP_LAUNCH(tg, (k_sep_u<64, 1, 3, 1, 0, 1>), grid, dim3(256), 0, s, a, ntiles);
        else PP_LAUNCH(tg, (k_sep_u<32, 1, 3, 1, 0, 1>), grid, dim3(256), 0, s, a, ntiles);
    } else {
        if (nt == 128) PP_LAUNCH(tg, (k_sep_u<128, 2, 2, 1, 0, 1>), grid, dim3(256), 0, s, a, ntiles);
        else if (nt == 64) PP_LAUNCH(tg, (k_sep_u<64, 2, 3, 1, 0, 1>), grid, dim3(256), 0, s, a, ntiles);
        else PP_LAUNCH(tg, (k_sep_u<32, 2, 3, 1, 0, 1>), grid, dim3(256), 0, s, a, ntiles);
    }
    return gx;          // one statistics row per workgroup of a channel column
#endif
}

// Training-mode forward of a plain product rows x K -> rows x N (a transposed convolution as a GEMM over its input
// pixels, the heads): k_sep_u<..., TR = 2> -- no depthwise, the rest as launch_sep_train.  `in` needs no header.
// Returns the statistics rows written (t.stat != NULL) or 1, 0 when the shape is not one the kernel takes.
int launch_rows_train(const RowsTrainArgs& t, hipStream_t s) {
#if PP_SPLIT_MODE != 1
    (void)t; (void)s;
    return 0;
#else
    const long long M = t.rows;
    if (t.K % KC != 0 || t.N % 32 != 0 || M % 2 != 0 || M >= (1 << 24) || M * t.K * 4 >= (1ll << 31) - 4096 || t.ld_out % 4 != 0)
        return 0;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = t.in; a.wt16 = t.wt16; a.n_total = t.N; a.out = t.out; a.bias = t.bias;
    a.M = (int)M; a.in_h = 1; a.in_w = (int)M; a.cin = t.K; a.px_h = 1; a.px_w = (int)M;
    a.stride = 1; a.ld_out = t.ld_out; a.co_off = 0; a.cout = t.N;
    a.tr_stat = t.stat;
    const int ntiles = (a.M + 127) / 128;
    const int nt = (t.N % 128 == 0) ? 128 : (t.N % 64 == 0 ? 64 : 32);
    const int ny = t.N / nt;
    const int wpb = 3;          // (no depthwise: 159 / 105 / 76 VGPRs -- three workgroups per CU for every tile width)
    int slots = (g_num_cus * wpb) / ny;
    if (slots < 8) slots = 8;
    int gx = ntiles < slots ? ntiles : slots;
    gx = (gx + 7) & ~7;
    const dim3 grid((unsigned)gx, ny);
    const char* tg = t.tag ? t.tag : "k_sep_u_tr";
    if (nt == 128) PP_LAUNCH(tg, (k_sep_u<128, 1, 3, 1, 0, 2>), grid, dim3(256), 0, s, a, ntiles);
    else if (nt == 64) PP_LAUNCH(tg, (k_sep_u<64, 1, 3, 1, 0, 2>), grid, dim3(256), 0, s, a, ntiles);
    else PP_LAUNCH(tg, (k_sep_u<32, 1, 3, 1, 0, 2>), grid, dim3(256), 0, s, a, ntiles);
    return gx;
#endif
}

// ---------------------------------------------------------------------------------------
// Separable stride-1 layer with 256 output channels, depthwise computed ONCE per pixel (two-piece builds).
//
// k_sep_u runs a 256-channel layer as two columns of 128-channel workgroups, each computing the depthwise of its
// pixels again (window loads and FMAs twice per pixel).  Here one 8-wave workgroup owns 128 pixels x all 256
// channels: wave (pg, grp) belongs to pixel group pg = wave & 3 (32 pixels) and channel half grp = wave >> 2.  A
// stream position is a PAIR of K-chunks: wave (pg, grp) stages chunk 2p + grp of its pixel group (loads, depthwise,
// A tile -- exactly k_sep_u's staging), both waves of a pixel group then multiply BOTH chunks' A tiles with their own
// 128 output channels.  Per output: half the window loads and half the depthwise FMAs, the same MFMAs, half as many
// iterations (each with twice the matrix work to overlap).  113 KB of LDS: one workgroup (two waves per SIMD) per CU,
// the occupancy of the 200-register k_sep_u<128>.  Skeleton (persistent XCD-aware tile walk, load / stage / multiply
// cursors, one barrier per position, transposing epilogue) as k_sep_u.  cin % 64 == 0.
#if PP_SPLIT_MODE != 0
template <int CO>   // output channels of the layer (256)
__global__ __launch_bounds__(512, 1) void k_sep_p(GemmArgs a, int ntiles) {
    constexpr int KCH = 16, LSTR = KCH + 4, WW = 4, NLD = 12;
    constexpr int SAW = 32 * LSTR;                   // one A tile (floats)
    constexpr int SB1 = PP_NPIECE * CO * 8;          // weights of one chunk: [pieces][CO][16 f16]
    constexpr int NTILES = CO / 64;                  // channel tiles of a wave (half the channels)
    constexpr int NB4 = (2 * CO * 2 * PP_NPIECE) / 512;   // 16-byte weight items per thread per position
    __shared__ __attribute__((aligned(16))) float smem[16 * SAW + 4 * SB1 + 9 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pg = wave & 3, grp = wave >> 2;
    float* const sA = smem;                          // [pg][chunk of the pair][buffer][SAW]
    float* const sB = smem + 16 * SAW;               // [buffer][chunk of the pair][SB1]
    float* const sDW = smem + 16 * SAW + 4 * SB1;

    const int xcd = blockIdx.x & 7, gl = blockIdx.x >> 3, GL = gridDim.x >> 3;
    const int tbase = (int)(((long long)xcd * ntiles) >> 3), tend = (int)(((long long)(xcd + 1) * ntiles) >> 3);
    const int first = tbase + gl;
    if (first >= tend) return;
    const int ntl = (tend - first + GL - 1) / GL;
    const int cin = a.cin;
    const int npos = cin / (2 * KCH);                // positions (chunk pairs) per tile: even (cin % 64 == 0)
    const int total = ntl * npos;

    const int c4 = lane & 3, q = lane >> 2;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(reinterpret_cast<const char*>(a.in) - PP_ZPAD_FLOATS * 4);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc((const void*)a.wt16);
    const int hw = a.px_h * a.px_w;
    const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)a.px_w;
    const int rs4 = a.in_w * cin * 4, cin4 = cin * 4;
    unsigned aoff[NLD];
#define P_TILE_OFFSETS(TILE)                                                                             \
    {                                                                                                    \
        const int pix0_ = (TILE) * 128 + pg * 32 + 2 * q;                                                \
        const bool pvalid_ = pix0_ < a.M;                                                                \
        const int pc_ = pvalid_ ? pix0_ : 0;                                                             \
        int b_, rem_, y_, x0_;                                                                           \
        fast_divmod(pc_, hw, inv_hw, b_, rem_);                                                          \
        fast_divmod(rem_, a.px_w, inv_w, y_, x0_);                                                       \
        const unsigned cbase_ = (unsigned)(((b_ * a.in_h + y_) * a.in_w + x0_) * cin) * 4u + PP_ZPAD_FLOATS * 4u; \
        bool rowok_[3], colok_[WW];                                                                      \
        _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_) rowok_[dy_] = pvalid_ & ((unsigned)(y_ - 1 + dy_) < (unsigned)a.in_h); \
        _Pragma("unroll") for (int dx_ = 0; dx_ < WW; ++dx_) colok_[dx_] = (unsigned)(x0_ - 1 + dx_) < (unsigned)a.in_w; \
        _Pragma("unroll") for (int e = 0; e < NLD; ++e) {                                                \
            const int dy_ = e / WW, dx_ = e % WW;                                                        \
            const bool ok_ = rowok_[dy_] & colok_[dx_];                                                  \
            aoff[e] = (ok_ ? cbase_ + (unsigned)((dy_ - 1) * rs4) + (unsigned)((dx_ - 1) * cin4) : 0u) + (unsigned)(c4 * 16); \
        }                                                                                                \
    }
    // weight items of a position: [chunk of the pair][piece][row][half]
    unsigned boff[NB4];
    int bdst[NB4];
    const unsigned bstep = (unsigned)(PP_NPIECE * a.n_total * 32);       // bytes per K-chunk
#pragma unroll
    for (int r = 0; r < NB4; ++r) {
        const int e_ = tid + 512 * r;
        const int cc = e_ / (CO * 2 * PP_NPIECE), e2 = e_ % (CO * 2 * PP_NPIECE);
        const int piece = e2 / (CO * 2), rem = e2 % (CO * 2), row = rem >> 1, half = rem & 1;
        boff[r] = (unsigned)(((piece * a.n_total + row) * 16 + half * 8) * 2) + (unsigned)cc * bstep;
        bdst[r] = cc * SB1 + piece * (CO * 8) + row * 8 + ((half ^ ((row >> 3) & 1)) * 4);
    }
    float4 rin[NLD];
    float4 rb[NB4];
#define P_LOAD(POS)                                                                                      \
    {                                                                                                    \
        const unsigned so_ = (unsigned)(2 * (POS) + grp) * (KCH * 4);                                    \
        _Pragma("unroll") for (int e = 0; e < NLD; ++e) rin[e] = buf_load16(rs_in, aoff[e], so_);        \
        const unsigned sb_ = (unsigned)(2 * (POS)) * bstep;                                              \
        _Pragma("unroll") for (int r = 0; r < NB4; ++r) rb[r] = buf_load16(rs_wt, boff[r], sb_);         \
    }
    int ld_tile = first, ld_pos = 0;
    int st_pos = 0;
    int mm_tile = first, mm_pos = 0;
    P_TILE_OFFSETS(ld_tile)
    P_LOAD(0)
    ld_pos = 1;
    {
        const int ngrp = cin / 4;
        for (int e = tid; e < 9 * ngrp; e += 512) {
            const int t = e / ngrp, g4 = e - t * ngrp;
            reinterpret_cast<float4*>(sDW)[g4 * 9 + t] = reinterpret_cast<const float4*>(a.dw)[e];
        }
    }
    f32x16 acc[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int h = lane >> 5, r32 = lane & 31;
    const int n0 = grp * (CO / 2);
    float bias_r[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n) bias_r[n] = a.bias[n0 + n * 32 + r32];
    __syncthreads();

    const float* const cA0 = sA + pg * (4 * SAW) + r32 * LSTR + h * (KCH / 2);
    const float* const cB0 = sB + (n0 + r32) * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
#define P_MFMA(I)                                                                                        \
    {                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                  \
            const float* cA = cA0 + (c * 2 + ((I) & 1)) * SAW;                                           \
            const float* cB = cB0 + (((I) & 1) * 2 + c) * SB1;                                           \
            const float4 a0 = *reinterpret_cast<const float4*>(cA), a1 = *reinterpret_cast<const float4*>(cA + 4);  \
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};                        \
            bf16x8 ah, am, al;                                                                           \
            split_bf16x3(av, ah, am, al);                                                                \
            _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                         \
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(cB + n * 32 * 8);                     \
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(cB + CO * 8 + n * 32 * 8);           \
                [[maybe_unused]] const bf16x8 bl = PC_LO(cB);                                            \
                PC_PRODUCTS(acc[n], ah, am, al, bh, bm, bl)                                              \
            }                                                                                            \
        }                                                                                                \
    }
    // staging of position st_pos into buffer P & 1: this wave's chunk (2 st_pos + grp) of its pixel group, its
    // share of the position's weights
#define P_STAGE(P)                                                                                       \
    {                                                                                                    \
        constexpr int buf = (P) & 1;                                                                     \
        const float* tw = sDW + ((2 * st_pos + grp) * 4 + c4) * 36;                                      \
        float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;                                            \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                 \
            _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                           \
                const float4 w4 = *reinterpret_cast<const float4*>(tw + (dy * 3 + dx) * 4);              \
                const float4 v0 = rin[dy * WW + dx], v1 = rin[dy * WW + 1 + dx];                         \
                o0.x = fmaf(v0.x, w4.x, o0.x); o0.y = fmaf(v0.y, w4.y, o0.y);                            \
                o0.z = fmaf(v0.z, w4.z, o0.z); o0.w = fmaf(v0.w, w4.w, o0.w);                            \
                o1.x = fmaf(v1.x, w4.x, o1.x); o1.y = fmaf(v1.y, w4.y, o1.y);                            \
                o1.z = fmaf(v1.z, w4.z, o1.z); o1.w = fmaf(v1.w, w4.w, o1.w);                            \
            }                                                                                            \
        float* dA = sA + pg * (4 * SAW) + (grp * 2 + buf) * SAW + (2 * q) * LSTR + c4 * 4;               \
        *reinterpret_cast<float4*>(dA) = o0;                                                             \
        *reinterpret_cast<float4*>(dA + LSTR) = o1;                                                      \
        _Pragma("unroll") for (int r = 0; r < NB4; ++r)                                                  \
            *reinterpret_cast<float4*>(sB + buf * (2 * SB1) + bdst[r]) = rb[r];                          \
        if (++st_pos == npos) st_pos = 0;                                                                \
    }
#define P_ISSUE()                                                                                        \
    {                                                                                                    \
        if (ld_pos == 0) P_TILE_OFFSETS(ld_tile)                                                         \
        P_LOAD(ld_pos)                                                                                   \
        if (++ld_pos == npos) { ld_pos = 0; ld_tile += GL; }                                             \
    }
#define P_EPILOGUE()                                                                                     \
    {                                                                                                    \
        const int pw = mm_tile * 128 + pg * 32;                                                          \
        if (pw < a.M) {                                                                                  \
            const int qi = lane & 3, qj = r32 >> 2;                                                      \
            float* dst = a.out + (size_t)(pw + 4 * h + qi) * a.ld_out + a.co_off + n0 + qj * 4;          \
            const int ldo = a.ld_out;                                                                    \
            _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                         \
                const float bvn = bias_r[n];                                                             \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                          \
                    float x0 = relu_keep_nan(acc[n][4 * g + 0] + bvn), x1 = relu_keep_nan(acc[n][4 * g + 1] + bvn);  \
                    float x2 = relu_keep_nan(acc[n][4 * g + 2] + bvn), x3 = relu_keep_nan(acc[n][4 * g + 3] + bvn);  \
                    quad_transpose4(x0, x1, x2, x3, lane);                                               \
                    if (pw + 8 * g + 4 * h + qi < a.M)                                                   \
                        *reinterpret_cast<float4*>(dst + (8 * g) * ldo + n * 32) = make_float4(x0, x1, x2, x3);  \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int n = 0; n < NTILES; ++n)                                               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;                              \
        mm_pos = 0;                                                                                      \
        mm_tile += GL;                                                                                   \
    }
    P_STAGE(0)
    P_ISSUE()
    __syncthreads();
    int i = 0;
    for (; i + 3 < total; i += 2) {
        P_MFMA(0)
        P_STAGE(1)
        P_ISSUE()
        if (++mm_pos == npos) P_EPILOGUE()
        __syncthreads();
        P_MFMA(1)
        P_STAGE(0)
        P_ISSUE()
        if (++mm_pos == npos) P_EPILOGUE()
        __syncthreads();
    }
    P_MFMA(0)
    P_STAGE(1)
    if (++mm_pos == npos) P_EPILOGUE()
    __syncthreads();
    P_MFMA(1)
    ++mm_pos;
    P_EPILOGUE()
#undef P_MFMA
#undef P_STAGE
#undef P_ISSUE
#undef P_EPILOGUE
#undef P_LOAD
#undef P_TILE_OFFSETS
}

static void launch_p(const GemmArgs& a, hipStream_t s) {
    const int ntiles = (a.M + 127) / 128;
    int gx = ntiles < g_num_cus ? ntiles : g_num_cus;
    gx = (gx + 7) & ~7;
    PP_LAUNCH("k_sep_p", k_sep_p<256>, dim3((unsigned)gx), dim3(512), 0, s, a, ntiles);
}
#endif   // PP_SPLIT_MODE != 0

// does k_sep_p run this separable layer?  (256 output channels, stride 1, dense input, two-piece builds, large enough
// for the persistent kernels; PP_SEP_P=0: never)
static bool sep_p_runs(const void* wt16, int stride, int cin, int cout, int n_total, long long M, const int* occ, int dbg) {
#if PP_SPLIT_MODE != 0
    static int v = -1;
    if (v < 0) { const char* e = getenv("PP_SEP_P"); v = (e && e[0] == '0') ? 0 : 1; }
    // (the 128-channel instantiation, where k_sep_u already computes the depthwise once, measured 32 us against 28.5)
    return v != 0 && wt16 != nullptr && split_precision(dbg) && stride == 1 && occ == nullptr && cout == 256 &&
           n_total == 256 && cin % 64 == 0 && cin <= 256 && M < (1 << 24);
#else
    return false;
#endif
}

// ---------------------------------------------------------------------------------------
// Split-K separable layer for small maps (few frames in flight: the latency case).
//
// With a handful of 32-pixel wave tiles the chip is empty and a layer's time is the length of ONE
// wave's dependent chain: cin / 16 K-chunks of load -> depthwise -> LDS -> MFMA.  Here the four waves
// of a workgroup share one 32-pixel x NT tile and split the K range instead: wave w takes chunks w,
// w + 4, ...; each wave has its own A and weight tiles in LDS (no workgroup barrier inside the K loop),
// and the four partial accumulators are added through LDS in a fixed order (deterministic), wave n
// finishing channel tile n (bias, ReLU, 16-byte stores).  The chain is a quarter as long.
// Split-precision bf16 only; cin % 64 == 0.
// NW: waves per workgroup = ways the K range is split (4, or 8 when even the 4-way kernel leaves the chip to less
// than one workgroup per CU: the chain halves again, one workgroup per CU by LDS)
template <int NT, int S, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_sep_k4(GemmArgs a) {
    constexpr int KCH = 16, LSTR = KCH + 4;
    constexpr int WW = S + 3, NLD = 3 * WW;
    constexpr int SAW = 32 * LSTR;              // one A buffer (floats)
    constexpr int SB = PP_NPIECE * NT * 8;              // one weight buffer: [3 pieces][NT][16 bf16]
    constexpr int NTILES = NT / 32;
    constexpr int NBL = NT * 2 * PP_NPIECE / 64;            // 16-byte weight items per lane per chunk
    constexpr int WV = 2 * SAW + 2 * SB;        // floats per wave
    static_assert(NW * NTILES * 16 * 64 <= NW * WV, "reduction scratch overlays the staging tiles");
    __shared__ __attribute__((aligned(16))) float smem[NW * WV + 9 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const sAw = smem + wave * WV;
    float* const sBw = sAw + 2 * SAW;
    float* const sDW = smem + NW * WV;
    const int n0 = blockIdx.y * NT;
    const int cin = a.cin;
    const int niter = cin / (KCH * NW);          // chunks per wave (>= 1, checked by the launcher)

    const int c4 = lane & 3, q = lane >> 2;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(reinterpret_cast<const char*>(a.in) - PP_ZPAD_FLOATS * 4);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc((const void*)a.wt16);
    unsigned aoff[NLD];
    {
        const int hw = a.px_h * a.px_w;
        const int pix0 = blockIdx.x * 32 + 2 * q;
        const bool pvalid = pix0 < a.M;
        const int pc = pvalid ? pix0 : 0;
        const int b = pc / hw, rem = pc - b * hw;
        const int y = rem / a.px_w, x0 = rem - y * a.px_w;
        const int yi = y * S - 1, xi = x0 * S - 1;
        const int rs4 = a.in_w * cin * 4, cin4 = cin * 4;
        const unsigned cbase = (unsigned)(((b * a.in_h + y * S) * a.in_w + x0 * S) * cin) * 4u + PP_ZPAD_FLOATS * 4u;
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int dy = e / WW, dx = e % WW;
            bool ok = pvalid & ((unsigned)(yi + dy) < (unsigned)a.in_h) & ((unsigned)(xi + dx) < (unsigned)a.in_w);
            if (a.occ != nullptr) {   // sparse canvas: empty cells were not written
                bool occ = false;
                const int cidx = ok ? (yi + dy) * a.in_w + xi + dx : 0;
                for (int z = 0; z < a.occ_nz; ++z)
                    occ = occ | (a.occ[(size_t)(b * a.occ_nz + z) * (a.in_h * a.in_w) + cidx] >= 0);
                ok = ok & occ;
            }
            aoff[e] = (ok ? cbase + (unsigned)((dy - 1) * rs4 + (dx - 1) * cin4) : 0u) + (unsigned)(c4 * 16);
        }
    }
    unsigned boff[NBL];
    int bdst[NBL];
#pragma unroll
    for (int r = 0; r < NBL; ++r) {
        const int e = lane + 64 * r;
        const int piece = e / (NT * 2), rem = e % (NT * 2), row = rem >> 1, half = rem & 1;
        boff[r] = (unsigned)(((piece * a.n_total + n0 + row) * 16 + half * 8) * 2);
        bdst[r] = piece * (NT * 8) + row * 8 + ((half ^ ((row >> 3) & 1)) * 4);
    }
    const unsigned bstep = (unsigned)(PP_NPIECE * a.n_total * 32);   // bytes per K-chunk of the split weights
    float4 rin[NLD];
    float4 rb[NBL];
#define K4_LOAD(KC)                                                                                      \
    {                                                                                                    \
        const unsigned so_ = (unsigned)(KC) * (KCH * 4), sb_ = (unsigned)(KC) * bstep;                   \
        _Pragma("unroll") for (int e = 0; e < NLD; ++e) rin[e] = buf_load16(rs_in, aoff[e], so_);        \
        _Pragma("unroll") for (int r = 0; r < NBL; ++r) rb[r] = buf_load16(rs_wt, boff[r], sb_);         \
    }
#define K4_STAGE(KC, BUF)                                                                                \
    {                                                                                                    \
        const float* tw = sDW + ((KC) * 4 + c4) * 36;                                                    \
        float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;                                            \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                 \
            _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                           \
                const float4 w4 = *reinterpret_cast<const float4*>(tw + (dy * 3 + dx) * 4);              \
                const float4 v0 = rin[dy * WW + dx], v1 = rin[dy * WW + S + dx];                         \
                o0.x = fmaf(v0.x, w4.x, o0.x); o0.y = fmaf(v0.y, w4.y, o0.y);                            \
                o0.z = fmaf(v0.z, w4.z, o0.z); o0.w = fmaf(v0.w, w4.w, o0.w);                            \
                o1.x = fmaf(v1.x, w4.x, o1.x); o1.y = fmaf(v1.y, w4.y, o1.y);                            \
                o1.z = fmaf(v1.z, w4.z, o1.z); o1.w = fmaf(v1.w, w4.w, o1.w);                            \
            }                                                                                            \
        float* dA = sAw + (BUF) * SAW + (2 * q) * LSTR + c4 * 4;                                         \
        *reinterpret_cast<float4*>(dA) = o0;                                                             \
        *reinterpret_cast<float4*>(dA + LSTR) = o1;                                                      \
        _Pragma("unroll") for (int r = 0; r < NBL; ++r)                                                  \
            *reinterpret_cast<float4*>(sBw + (BUF) * SB + bdst[r]) = rb[r];                              \
    }
    K4_LOAD(wave)
    {   // depthwise taps -> LDS, after the first window loads are in flight (one prologue round trip, not two)
        const int ngrp = cin / 4;
        for (int e = tid; e < 9 * ngrp; e += NW * 64) {
            const int t = e / ngrp, g4 = e - t * ngrp;
            reinterpret_cast<float4*>(sDW)[g4 * 9 + t] = reinterpret_cast<const float4*>(a.dw)[e];
        }
    }
    f32x16 acc[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int h = lane >> 5, r32 = lane & 31;
    const float bias_w = (wave < NTILES) ? a.bias[n0 + wave * 32 + r32] : 0.f;
    __syncthreads();   // depthwise taps visible
    K4_STAGE(wave, 0)
    if (niter > 1) K4_LOAD(wave + NW)
    const float* const cA0 = sAw + r32 * LSTR + h * (KCH / 2);
    const float* const cB0 = sBw + r32 * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
#pragma unroll 1
    for (int j = 0; j < niter; ++j) {
        const int buf = j & 1;
        {
            const float* cA = cA0 + buf * SAW;
            const float* cB = cB0 + buf * SB;
            const float4 a0 = *reinterpret_cast<const float4*>(cA), a1 = *reinterpret_cast<const float4*>(cA + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            bf16x8 ah, am, al;
            split_bf16x3(av, ah, am, al);
#pragma unroll
            for (int n = 0; n < NTILES; ++n) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(cB + n * 32 * 8);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(cB + NT * 8 + n * 32 * 8);
                [[maybe_unused]] const bf16x8 bl = PC_LO(cB + 2 * NT * 8 + n * 32 * 8);
                PC_PRODUCTS(acc[n], ah, am, al, bh, bm, bl)
            }
        }
        if (j + 1 < niter) {
            K4_STAGE(wave + NW * (j + 1), buf ^ 1)
            if (j + 2 < niter) K4_LOAD(wave + NW * (j + 2))
        }
    }
#undef K4_LOAD
#undef K4_STAGE
    // ---- add the NW partial sums: [wave][channel tile][register group][lane][4] through LDS ----
    __syncthreads();   // every wave is done with its staging tiles
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(smem + (((wave * NTILES + n) * 4 + g) * 64 + lane) * 4) =
                make_float4(acc[n][4 * g], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]);
    __syncthreads();
    const int pw = blockIdx.x * 32;
    if (wave < NTILES) {
        const int qi = lane & 3, qj = r32 >> 2;
        float* dst = a.out + (size_t)(pw + 4 * h + qi) * a.ld_out + a.co_off + n0 + wave * 32 + qj * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = *reinterpret_cast<const float4*>(smem + (((0 * NTILES + wave) * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float4 u = *reinterpret_cast<const float4*>(smem + (((w * NTILES + wave) * 4 + g) * 64 + lane) * 4);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            float x0 = relu_keep_nan(t.x + bias_w), x1 = relu_keep_nan(t.y + bias_w);
            float x2 = relu_keep_nan(t.z + bias_w), x3 = relu_keep_nan(t.w + bias_w);
            quad_transpose4(x0, x1, x2, x3, lane);
            if (pw + 8 * g + 4 * h + qi < a.M)
                *reinterpret_cast<float4*>(dst + (size_t)(8 * g) * a.ld_out) = make_float4(x0, x1, x2, x3);
        }
    }
}

// the split-K kernel runs a layer while the map is too small to fill the chip with 32-pixel wave tiles.
// Measured crossover against k_sep_u (tools/k4_sweep.sh, cfg-A, B = 2..16): 1.5 workgroups per CU for
// cin = 64, 2.5 for cin = 128, 3 for cin = 256 -- the longer the K chain, the later k_sep_u catches up.
// PP_SEP_K4=0 turns it off, PP_SEP_K4=n sets the limit to n workgroups per CU for every layer.
static bool sep_k4_runs(const void* wt16, int cin, int n_total, long long M, int dbg) {
    static int force = -2;
    if (force == -2) { const char* e = getenv("PP_SEP_K4"); force = e ? atoi(e) : -1; }
    if (force == 0 || wt16 == nullptr || !split_precision(dbg) || cin % 64 != 0 || cin > 256 || n_total % 64 != 0)
        return false;
    const int half_cus = (force > 0) ? 2 * force : (cin <= 64 ? 3 : (cin <= 128 ? 5 : 6));
    return 2 * ((M + 31) / 32 * (n_total / 64)) <= (long long)half_cus * g_num_cus;
}
// eight-way split: cin = 256 and at most one workgroup per CU (the latency case proper: a frame or two).  Measured
// at B=1 (tools/latency_b1.py): block3.1-.5 8.1 -> 7.2 us; the 128-channel layers gain nothing (block2.x 5.8 -> 5.6,
// block3.0 6.0 -> 7.0: one iteration per wave leaves nothing to pipeline) and keep four waves
static bool sep_k8_runs(int cin, int n_total, long long M) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PP_SEP_K8"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0 && cin % 256 == 0 && (M + 31) / 32 * (n_total / 64) <= (long long)g_num_cus;
}
template <int S>
static void launch_k4(const GemmArgs& a, int n_total, hipStream_t s) {
    dim3 grid((unsigned)((a.M + 31) / 32), n_total / 64);
    if (sep_k8_runs(a.cin, n_total, a.M)) {
        PP_LAUNCH("k_sep_k4", (k_sep_k4<64, S, 8>), grid, dim3(512), 0, s, a);
        return;
    }
    PP_LAUNCH("k_sep_k4", (k_sep_k4<64, S>), grid, dim3(256), 0, s, a);
}

// Epilogue of one 128-pixel deconv tile (see k_deconv_u).  Accumulator layout (operands swapped in the
// MFMAs): lane & 31 = pixel of the wave's 32, register r of tile n = channel n*32 + DCH(r, h).
#define DCH(R, H) (((R) & 3) + 8 * ((R) >> 2) + 4 * (H))
template <int NT>
__device__ __forceinline__ void deconv_tile_epilogue(const GemmArgs& a, f32x16 (&acc)[NT / 32],
                                                     int tile, int wave, int lane, int cbase, int delta,
                                                     const int* opix_tab, const float* sHW, const float* s_hbias) {
    constexpr int NTILES = NT / 32;
    const int h = lane >> 5, r32 = lane & 31;
    const int pix = tile * 128 + wave * 32 + r32;
    if ((a.dbg & 4) || tile * 128 + wave * 32 >= a.M) return;
    const bool ok = pix < a.M;
    const size_t orow = (size_t)(opix_tab[wave * 32 + r32] + delta);   // this lane's output pixel
    const bool heads = a.head_mode != 0;
    // head map row of this pixel: this lane owns columns 4h + {0..3, 8..11, 16..19, 24..27} (4 x 16 bytes);
    // the earlier branches' partial sums (mode 2) are fetched now and used at the end
    // the head accumulator starts at the head bias (first branch) or at the partial sums the earlier branches
    // left in the head map (its 16 registers are exactly this lane's 4 x 16 bytes of the row)
    float* hrow = a.head + orow * PP_HEAD_COLS + 4 * h;
    f32x16 hacc;
    if (a.head_mode == 2 && ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(hrow + 8 * g);
            hacc[4 * g] = t.x; hacc[4 * g + 1] = t.y; hacc[4 * g + 2] = t.z; hacc[4 * g + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = (a.head_mode == 1) ? s_hbias[DCH(r, h)] : 0.f;
    }
    float* dst = (a.out != nullptr) ? a.out + orow * a.ld_out + a.co_off + cbase + 4 * h : nullptr;
#pragma unroll
    for (int n = 0; n < NTILES; ++n) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = relu_keep_nan(acc[n][r]);   // the folded bias is the accumulator's initial value
        if (heads) {
            // the activated values are already a B operand (k = channel, column = pixel) up to a fixed
            // permutation of k, which the head kernels carry as well (pp_api.hip: head_k_permutation)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float av[8] = {v[8 * g], v[8 * g + 1], v[8 * g + 2], v[8 * g + 3],
                                     v[8 * g + 4], v[8 * g + 5], v[8 * g + 6], v[8 * g + 7]};
                bf16x8 xh, xm, xl;
                split_bf16x3(av, xh, xm, xl);
                const float* hW = sHW + ((n * 2 + g) * PP_NPIECE * 32 + r32) * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(hW);
                const bf16x8 wm = *reinterpret_cast<const bf16x8*>(hW + 32 * 8);
                [[maybe_unused]] const bf16x8 wl = PC_LO(hW + 2 * 32 * 8);
                PC_PRODUCTS(hacc, wh, wm, wl, xh, xm, xl)
            }
        }
        if (dst != nullptr && ok) {   // concat slice (only when the heads are not fused): 4 consecutive channels per store
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(dst + n * 32 + 8 * g) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        }
    }
    if (heads && ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(hrow + 8 * g) = make_float4(hacc[4 * g], hacc[4 * g + 1], hacc[4 * g + 2], hacc[4 * g + 3]);
        if (a.cls_plane != nullptr) {   // register r of this lane is head column 8*(r/4) + 4h + r%4
            float* cp = a.cls_plane + orow * a.cls_ncol;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = 8 * (r >> 2) + 4 * h + (r & 3) - a.cls_col0;
                if ((unsigned)col < (unsigned)a.cls_ncol) cp[col] = hacc[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Uniform-wave Conv2DTranspose (kernel == stride) + BN + ReLU [+ fused SSD heads], split-precision bf16.
//
// Same skeleton as k_sep_u (persistent 4-wave workgroups, wave w owns input pixels 32w..32w+31 of a
// 128-pixel tile, one barrier per 16-channel K-chunk for the shared weight tile), but the input operand
// needs no LDS at all: lane (r, h) owns pixel r, channels 8h..8h+7 of the chunk, i.e. 32 contiguous
// bytes of the input row, loaded straight into registers two K-chunks ahead (two register sets
// alternate; every layer has an even number of chunks), split into three bf16 pieces and multiplied.
// The MFMAs take the WEIGHT fragment as the A operand and the input fragment as B, so the accumulator
// holds out^T: lane = pixel, registers = 16 channels of that pixel.  That layout is, up to a fixed
// permutation of k, the B operand of the head GEMM head^T[32 x px] = Wh^T[32 x ch] . act^T[ch x px]:
// the fused heads run straight out of the accumulator registers (ReLU, split, 6 MFMAs per 16 channels)
// with no transposition through LDS, and every lane ends up with 4 x 4 consecutive head columns of its
// own pixel, which it adds to the head map with four 16-byte read-modify-writes (race-free: the three
// deconv launches run in stream order and a pixel's row is touched by one half-wave pair per launch).
// blockIdx.y selects NT of the k*k*cout GEMM columns (one tap when NT == cout).  The folded bias is the
// accumulators' initial value (per-channel, from LDS).  Output pixel index per input pixel comes from a
// small double-buffered LDS table filled one tile ahead.
template <int NT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_deconv_u(GemmArgs a, int ntiles) {
    constexpr int KCH = 16;
    constexpr int NTILES = NT / 32;
    constexpr int SB = PP_NPIECE * NT * 8;                       // one weight buffer: [3 pieces][NT][8 floats]
    constexpr int NBI = NT * 2 * PP_NPIECE;                          // 16-byte weight items per chunk
    constexpr int NB4 = (NBI + 255) / 256;
    constexpr int SHW = (NT / 16) * PP_NPIECE * 32 * 8;          // head weights: [NT/16][3][32 cols][8 floats]
    __shared__ __attribute__((aligned(16))) float sB[2 * SB];
    __shared__ __attribute__((aligned(16))) float sHW[SHW];
    __shared__ float s_bias[NT];
    __shared__ float s_hbias[PP_HEAD_COLS];
    __shared__ int s_opix[2][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, r32 = lane & 31;
    const int dbg = a.dbg;

    const int xcd = blockIdx.x & 7, gl = blockIdx.x >> 3, GL = gridDim.x >> 3;
    const int tbase = (int)(((long long)xcd * ntiles) >> 3), tend = (int)(((long long)(xcd + 1) * ntiles) >> 3);
    const int first = tbase + gl;
    if (first >= tend) return;
    const int ntl = (tend - first + GL - 1) / GL;
    const int n0 = blockIdx.y * NT;
    const int cin = a.cin;
    const int nchunks = cin / KCH;                       // even (cin % 32 == 0, checked by the launcher)
    const int total = ntl * nchunks;
    const int tap = n0 / a.cout, cbase = n0 - tap * a.cout;
    const int ti = tap / a.k;
    const int delta = ti * (a.px_w * a.k) + (tap - ti * a.k);   // output-pixel offset of this tap
    const bool heads = a.head_mode != 0;

    // output pixel (tap (0,0)) of every input pixel of a tile -> s_opix[slot]
    const int hwpx = a.px_h * a.px_w;
#define D_FILL_OPIX(TILE, SLOT)                                                         \
    if (tid < 128) {                                                                    \
        const int pix_ = min((TILE) * 128 + tid, a.M - 1);                              \
        const int b_ = pix_ / hwpx, rem_ = pix_ - b_ * hwpx;                            \
        const int y_ = rem_ / a.px_w, x_ = rem_ - y_ * a.px_w;                          \
        s_opix[SLOT][tid] = (b_ * a.px_h * a.k + y_ * a.k) * (a.px_w * a.k) + x_ * a.k; \
    }
    D_FILL_OPIX(first, 0)

    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(reinterpret_cast<const char*>(a.in) - PP_ZPAD_FLOATS * 4);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(a.wt16);
    unsigned boff[NB4];
    int bdst[NB4];
#pragma unroll
    for (int r = 0; r < NB4; ++r) {
        const int e_ = (tid + 256 * r) % NBI;
        const int piece = e_ / (NT * 2), rem = e_ % (NT * 2), row = rem >> 1, half = rem & 1;
        boff[r] = (unsigned)(((piece * a.n_total + n0 + row) * 16 + half * 8) * 2);
        bdst[r] = piece * (NT * 8) + row * 8 + ((half ^ ((row >> 3) & 1)) * 4);
    }
    const unsigned bstep = (unsigned)(PP_NPIECE * a.n_total * 32);
    float4 rb0, rb1, rb2;
    rb0 = rb1 = rb2 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(NB4 <= 3, "weight prefetch registers");
#define D_LOAD_B(KCIDX)                                                                                  \
    {                                                                                                    \
        const unsigned sb_ = (unsigned)(KCIDX) * bstep;                                                  \
        if (NBI % 256 == 0 || NB4 > 1 || tid < NBI) rb0 = buf_load16(rs_wt, boff[0], sb_);               \
        if (NB4 > 1 && (NBI >= 512 || tid + 256 < NBI)) rb1 = buf_load16(rs_wt, boff[NB4 > 1 ? 1 : 0], sb_);  \
        if (NB4 > 2 && (NBI >= 768 || tid + 512 < NBI)) rb2 = buf_load16(rs_wt, boff[NB4 > 2 ? 2 : 0], sb_);  \
    }
#define D_STORE_B(BUF)                                                                                   \
    {                                                                                                    \
        if (NBI % 256 == 0 || NB4 > 1 || tid < NBI) *reinterpret_cast<float4*>(sB + (BUF) * SB + bdst[0]) = rb0;  \
        if (NB4 > 1 && (NBI >= 512 || tid + 256 < NBI)) *reinterpret_cast<float4*>(sB + (BUF) * SB + bdst[NB4 > 1 ? 1 : 0]) = rb1;  \
        if (NB4 > 2 && (NBI >= 768 || tid + 512 < NBI)) *reinterpret_cast<float4*>(sB + (BUF) * SB + bdst[NB4 > 2 ? 2 : 0]) = rb2;  \
    }
    // input operand: this lane's 32 bytes of a chunk; byte offset of the tile's row (0 -> zero header when
    // the pixel is past the end)
    int ld_tile = first, ld_kc = 0;
    unsigned avo = 0;
#define D_TILE_AOFF(TILE)                                                                                \
    {                                                                                                    \
        const int pix_ = (TILE) * 128 + wave * 32 + r32;                                                 \
        avo = (pix_ < a.M && !(dbg & 8)) ? (unsigned)(pix_ * cin + h * 8) * 4u + PP_ZPAD_FLOATS * 4u : (unsigned)(h * 32); \
    }
#define D_LOAD_A(R0, R1)                                                                                 \
    {                                                                                                    \
        if (ld_kc == 0) D_TILE_AOFF(ld_tile)                                                             \
        const unsigned so_ = (unsigned)ld_kc * (KCH * 4);                                                \
        R0 = buf_load16(rs_in, avo, so_);                                                                \
        R1 = buf_load16(rs_in, avo + 16u, so_);                                                          \
        if (++ld_kc == nchunks) { ld_kc = 0; ld_tile += GL; }                                            \
    }
    float4 ra0, ra1, rc0, rc1;           // raw input of even / odd stream positions
    D_LOAD_A(ra0, ra1)                   // position 0
    D_LOAD_A(rc0, rc1)                   // position 1 (total >= 2)
    D_LOAD_B(0)
    int lb_kc = 1;                       // chunk index of the next weight tile to load
    // bias and this branch's head-kernel slice -> LDS, after the first operand loads are in flight (the
    // prologue's memory round trips overlap)
    if (tid < NT) s_bias[tid] = a.bias[cbase + tid];
    if (heads) {   // three bf16 pieces; 16-byte halves swapped on odd groups of 8 columns (bank conflicts)
        for (int e = tid; e < SHW / 4; e += 256)
            reinterpret_cast<float4*>(sHW)[e ^ ((e >> 4) & 1)] = reinterpret_cast<const float4*>(a.head_wt16)[e];
        if (tid < PP_HEAD_COLS) s_hbias[tid] = a.head_bias[tid];
    }

    int mm_tile = first, mm_kc = 0, mm_slot = 0;
    __syncthreads();                     // bias / head weights / opix table visible
    f32x16 acc[NTILES];                  // accumulators start at the (BN-folded) bias of their channel
#define D_INIT_ACC()                                                                                     \
    _Pragma("unroll") for (int n = 0; n < NTILES; ++n)                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[n][r] = s_bias[n * 32 + DCH(r, h)];
    D_INIT_ACC()
    D_STORE_B(0)
    D_LOAD_B(lb_kc)
    if (++lb_kc == nchunks) lb_kc = 0;
    __syncthreads();

    // one stream position: multiply chunk i (raw input in RA0/RA1), prefetch position i+2 into the same registers
#define D_STEP(I, RA0, RA1)                                                                              \
    {                                                                                                    \
        const int i_ = (I);                                                                              \
        const float av_[8] = {RA0.x, RA0.y, RA0.z, RA0.w, RA1.x, RA1.y, RA1.z, RA1.w};                   \
        bf16x8 ah_, am_, al_;                                                                            \
        split_bf16x3(av_, ah_, am_, al_);                                                                \
        if (i_ + 2 < total) D_LOAD_A(RA0, RA1)                                                           \
        if (!(dbg & 1)) {                                                                                \
            const float* cB_ = sB + (i_ & 1) * SB + r32 * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);              \
            _Pragma("unroll") for (int n = 0; n < NTILES; ++n) {                                         \
                const bf16x8 bh_ = *reinterpret_cast<const bf16x8*>(cB_ + n * 32 * 8);                   \
                const bf16x8 bm_ = *reinterpret_cast<const bf16x8*>(cB_ + NT * 8 + n * 32 * 8);          \
                [[maybe_unused]] const bf16x8 bl_ = PC_LO(cB_ + 2 * NT * 8 + n * 32 * 8);      \
                PC_PRODUCTS(acc[n], bh_, bm_, bl_, ah_, am_, al_)                                        \
            }                                                                                            \
        }                                                                                                \
        if (i_ + 1 < total && !(dbg & 512)) D_STORE_B((i_ + 1) & 1)   /* weight tile of position i+1 -> LDS */ \
        if (++mm_kc == nchunks) {                                                                        \
            deconv_tile_epilogue<NT>(a, acc, mm_tile, wave, lane, cbase, delta, s_opix[mm_slot], sHW, s_hbias);  \
            D_INIT_ACC()                                                                                 \
            mm_kc = 0;                                                                                   \
            mm_tile += GL;                                                                               \
            mm_slot ^= 1;                                                                                \
        } else if (mm_kc == 1 && mm_tile + GL < tend) {                                                  \
            D_FILL_OPIX(mm_tile + GL, mm_slot ^ 1)   /* table of the next tile, one tile ahead */        \
        }                                                                                                \
        if (i_ + 1 < total) {   /* weight loads of position i+2 (after the epilogue: not live across it) */ \
            if (i_ + 2 < total && !(dbg & 512)) D_LOAD_B(lb_kc)                                          \
            if (++lb_kc == nchunks) lb_kc = 0;                                                           \
        }                                                                                                \
        /* tuning aid, bit 512: no weight staging and no barrier (wrong results): what resident weights could give */ \
        if (!(dbg & 512)) __syncthreads();                                                               \
    }
    for (int i = 0; i < total; i += 2) {
        D_STEP(i, ra0, ra1)
        D_STEP(i + 1, rc0, rc1)
    }
#undef D_STEP
#undef D_INIT_ACC
#undef D_LOAD_A
#undef D_TILE_AOFF
#undef D_STORE_B
#undef D_LOAD_B
#undef D_FILL_OPIX
}

#if PP_SPLIT_MODE != 0
// ---------------------------------------------------------------------------------------
// Conv2DTranspose (kernel == stride) + BN + ReLU [+ fused SSD heads] with the INPUT operand resident in
// registers (round 3).  k_deconv_u above gives every (pixel tile, tap) pair to a workgroup of its own tap
// column: an input fragment is fetched (through L2) and split into its two float16 pieces once per tap -- 16
// times for the 4x4 transposed convolution -- and the per-chunk split, address arithmetic and barrier share
// the SIMD with the matrix work (measured: 54 of deconv3's 88 us remain with neither weights nor MFMAs).
// Here the work is the flat list of (tile, tap) units, tile-major, cut into equal contiguous runs, one per
// persistent workgroup (2 per CU): a wave loads the CIN channels of its 32 pixels ONCE per tile, splits them
// once, and keeps the pieces in CIN/2 registers for all the taps of its run.  The K loop is then matrix
// instructions and LDS fragment reads only:
//   * weights stream through LDS as 16 KB panels = 8 steps of (32-channel n-tile, 16-channel K-chunk) x 2
//     pieces, double-buffered, ONE workgroup barrier per panel (24 MFMAs per wave);
//   * one 16-register accumulator (out^T of a 32-pixel x 32-channel tile, as k_deconv_u: lane = pixel) walks
//     the tap's four n-tiles in turn; after an n-tile's last chunk: ReLU, optional concat-slice store, split,
//     6 MFMAs of the fused head GEMM straight from the registers;
//   * after the tap's last n-tile the lane's 4 x 16 bytes of its output pixel's head row are written (first
//     branch: + head bias; later branches: + the partial sums fetched at the start of the unit).
// cout == 128 (one tap = four n-tiles); CIN 64 / 128 / 256.  Same summation order per output as k_deconv_u
// (chunks ascending, three products per chunk smallest first), so the two kernels agree bit for bit.
#ifdef PP_DECONV_ABLATE   // tuning build: pp_bench_layer's ablation bits switch phases of k_deconv_r off (wrong results)
#define R_ABL(BIT) (a.dbg & (BIT))
// bit 64: shader-clock stamps of workgroups 0..63 (wave 0): [0] start, [1] after the prologue, per unit u of the run
// [2 + 8u] unit start, [3 + 8u] input tile in registers, [4 + 8u .. 7 + 8u] n-tiles done, [8 + 8u] head row stored
#define R_STAMP(IDX) { if ((a.dbg & 64) && a.stamps && blockIdx.x < 64 && tid == 0 && (IDX) < 64) a.stamps[4096 * 8 + blockIdx.x * 64 + (IDX)] = clock64(); }
#else
#define R_ABL(BIT) false
#define R_STAMP(IDX) {}
#endif
template <int CIN>
__global__ __launch_bounds__(256, 2) void k_deconv_r(GemmArgs a, int nunits, int upw) {
    constexpr int NCH = CIN / 16;                        // K-chunks
    constexpr int STEPS = 4 * NCH;                       // (n-tile, chunk) steps of one unit
    constexpr int NPU = STEPS / 8;                       // weight panels per unit (8 steps each): 8 / 4 / 2
    constexpr int PANEL = 8 * 2 * 256;                   // floats: 8 steps x 2 pieces x 1 KB
    constexpr int NB = 3;                                // LDS ring of weight panels
    constexpr int PD = 2;                                // fragment reads run PD steps ahead of their products
    constexpr int SHW = (128 / 16) * 2 * 32 * 8;         // head weights: [cout/16][2 pieces][32 cols][8 floats]
    static_assert(PP_NPIECE == 2, "two-piece operands");
    static_assert(STEPS % 4 == 0 && PD < 4, "fragment ring of four");
    __shared__ __attribute__((aligned(16))) float sW[NB * PANEL];
    __shared__ __attribute__((aligned(16))) float sHW[SHW];
    __shared__ __attribute__((aligned(16))) float s_bias[128];
    __shared__ float s_hbias[PP_HEAD_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;

    // workgroup -> run of units; workgroups that share an XCD (equal blockIdx.x % 8) take neighbouring runs
    int wg;
    {
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = blockIdx.x & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int u0 = wg * upw, u1 = min(u0 + upw, nunits);
    if (u0 >= u1) return;
    R_STAMP(0)
    const int ntaps = a.k * a.k;
    const bool heads = a.head_mode != 0;
    const int hwpx = a.px_h * a.px_w, OW = a.px_w * a.k;

    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(reinterpret_cast<const char*>(a.in) - PP_ZPAD_FLOATS * 4);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(a.wt16);
    // weight staging: item j of this thread = block (wave + 4j) of the panel = step (wave >> 1) + 2j, piece wave & 1;
    // inside the 1 KB block: row lane >> 1, 16-byte half lane & 1 (halves swapped on odd groups of 8 rows)
    const int srow = lane >> 1, shalf = lane & 1;
    const unsigned wvoff = (unsigned)((((wave & 1) * a.n_total + srow) * 16 + shalf * 8) * 2);
    const int wdst = wave * 256 + srow * 8 + ((shalf ^ ((srow >> 3) & 1)) * 4);
    const unsigned cstep = (unsigned)(2 * a.n_total * 32);       // bytes per K-chunk of the split weights
    float4 rw[4];
    rw[0] = rw[1] = rw[2] = rw[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    int tile = u0 / ntaps, tap = u0 - tile * ntaps;
    // the flat sequence of weight panels of this run: (unit, panel) -> next to fetch
    int ld_left = (u1 - u0) * NPU, ld_tap = tap, ld_p = 0;
#define R_LOAD_NEXT()                                                                                    \
    {                                                                                                    \
        if (ld_left > 0 && !R_ABL(512)) {                                                                \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                           \
                const int fs_ = ld_p * 8 + (wave >> 1) + 2 * j_;                                         \
                const int nt_ = fs_ / NCH, c_ = fs_ % NCH;                                               \
                rw[j_] = buf_load16(rs_wt, wvoff, (unsigned)c_ * cstep + (unsigned)((ld_tap * 128 + nt_ * 32) * 32)); \
            }                                                                                            \
        }                                                                                                \
        --ld_left;                                                                                       \
        if (++ld_p == NPU) { ld_p = 0; if (++ld_tap == ntaps) ld_tap = 0; }                              \
    }
#define R_STORE_W(SLOT)                                                                                  \
    if (!R_ABL(512)) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                    \
        *reinterpret_cast<float4*>(sW + (SLOT) * PANEL + wdst + j_ * 1024) = rw[j_];

    // the first input tile's loads go out before anything else: they are the longest round trip of the prologue (every
    // workgroup of the launch asks for its tile at once), and the weight / bias / head-kernel staging below overlaps it
    float4 ra0[NCH][2];
    const int pix0_ = tile * 128 + wave * 32 + r32;
    const bool ok0_ = pix0_ < a.M;
    {
        const unsigned avo = (ok0_ && !R_ABL(8)) ? (unsigned)(pix0_ * CIN + h * 8) * 4u + PP_ZPAD_FLOATS * 4u : (unsigned)(h * 32);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            ra0[c][0] = buf_load16(rs_in, avo, (unsigned)c * 64u);
            ra0[c][1] = buf_load16(rs_in, avo + 16u, (unsigned)c * 64u);
        }
    }
    R_LOAD_NEXT()                        // panel 0
    for (int e = tid; e < 128; e += 256) s_bias[e] = a.bias[e];
    if (heads) {
        for (int e = tid; e < SHW / 4; e += 256)
            reinterpret_cast<float4*>(sHW)[e ^ ((e >> 4) & 1)] = reinterpret_cast<const float4*>(a.head_wt16)[e];
        if (tid < PP_HEAD_COLS) s_hbias[tid] = a.head_bias[tid];
    }
    R_STORE_W(0)
    R_LOAD_NEXT()                        // panel 1: written at step 0 of panel 0
    __syncthreads();
    R_STAMP(1)

    bf16x8 xh[NCH], xm[NCH];             // the wave's 32 pixels x CIN channels, two float16 pieces
    int tile_cur = tile, opix0;
    bool ok = ok0_;
    {
        const int pc = ok ? pix0_ : 0;
        const int pb = pc / hwpx, prem = pc - pb * hwpx;
        const int py = prem / a.px_w, pxx = prem - py * a.px_w;
        opix0 = (pb * a.px_h * a.k + py * a.k) * OW + pxx * a.k;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float av[8] = {ra0[c][0].x, ra0[c][0].y, ra0[c][0].z, ra0[c][0].w, ra0[c][1].x, ra0[c][1].y, ra0[c][1].z, ra0[c][1].w};
            bf16x8 lo_;
            split_bf16x3(av, xh[c], xm[c], lo_);
        }
    }
    const float* const cW = sW + r32 * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
    const float* const cHW = sHW + r32 * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
    int slot = 0;                        // ring slot of the panel being multiplied
    const float* pcur = cW;              // this lane's fragment base in that slot / in the next one
    const float* pnxt = cW + PANEL;
    // fragment ring: step fs uses entry fs % 4, requested PD steps earlier (also across panels and units)
    bf16x8 fh[4], fm[4];
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        fh[i] = *reinterpret_cast<const bf16x8*>(pcur + i * 512);
        fm[i] = *reinterpret_cast<const bf16x8*>(pcur + i * 512 + 256);
    }

    // CIN <= 128: the raw input of the NEXT tile is fetched during the last unit of the current one (CIN / 4 more
    // registers): the single-tap deconv1 changes tile with every unit, and the round trip was exposed every time
    constexpr bool PREF = CIN <= 128;
    float4 rn[PREF ? NCH : 1][2];
    int pre_tile = -1;
    for (int u = u0; u < u1; ++u) {
        R_STAMP(2 + 8 * (u - u0))
        if (tile != tile_cur) {          // uniform: fetch and split this tile's input once
            tile_cur = tile;
            const int pix = tile * 128 + wave * 32 + r32;
            ok = pix < a.M;
            const unsigned avo = (ok && !R_ABL(8)) ? (unsigned)(pix * CIN + h * 8) * 4u + PP_ZPAD_FLOATS * 4u : (unsigned)(h * 32);
            float4 ra[NCH][2];
            if (PREF && pre_tile == tile) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) { ra[c][0] = rn[PREF ? c : 0][0]; ra[c][1] = rn[PREF ? c : 0][1]; }
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    ra[c][0] = buf_load16(rs_in, avo, (unsigned)c * 64u);
                    ra[c][1] = buf_load16(rs_in, avo + 16u, (unsigned)c * 64u);
                }
            }
            const int pc = ok ? pix : 0;
            const int pb = pc / hwpx, prem = pc - pb * hwpx;
            const int py = prem / a.px_w, pxx = prem - py * a.px_w;
            opix0 = (pb * a.px_h * a.k + py * a.k) * OW + pxx * a.k;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float av[8] = {ra[c][0].x, ra[c][0].y, ra[c][0].z, ra[c][0].w, ra[c][1].x, ra[c][1].y, ra[c][1].z, ra[c][1].w};
                bf16x8 lo_;
                split_bf16x3(av, xh[c], xm[c], lo_);
            }
        }
        R_STAMP(3 + 8 * (u - u0))
        const int ti = tap / a.k;
        const size_t orow = (size_t)(opix0 + ti * OW + (tap - ti * a.k));      // this lane's output pixel
        float* const hrow = a.head + orow * PP_HEAD_COLS + 4 * h;              // its columns 4h + {0..3, 8.., 16.., 24..}
        f32x16 hacc;
        if (a.head_mode == 2 && ok && !R_ABL(4)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(hrow + 8 * g);
                hacc[4 * g] = t.x; hacc[4 * g + 1] = t.y; hacc[4 * g + 2] = t.z; hacc[4 * g + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[r] = (a.head_mode == 1) ? s_hbias[DCH(r, h)] : 0.f;
        }
        float* const dst = (a.out != nullptr) ? a.out + orow * a.ld_out + a.co_off + 4 * h : nullptr;
        f32x16 acc;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fs = 0; fs < STEPS; ++fs) {
            const int s = fs & 7, nt = fs / NCH, c = fs % NCH;       // compile-time after unrolling
            if (c == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = *reinterpret_cast<const float4*>(s_bias + nt * 32 + 8 * g + 4 * h);
                    acc[4 * g] = t.x; acc[4 * g + 1] = t.y; acc[4 * g + 2] = t.z; acc[4 * g + 3] = t.w;
                }
            }
            if (s == 4) {                // publishes the panel written at step 0 (read from step 8 - PD on)
                if (!R_ABL(512)) __syncthreads();
            }
            {   // fragments of step fs + PD
                const float* w_ = (s + PD < 8) ? pcur + (s + PD) * 512 : pnxt + (s + PD - 8) * 512;
                fh[(fs + PD) & 3] = *reinterpret_cast<const bf16x8*>(w_);
                fm[(fs + PD) & 3] = *reinterpret_cast<const bf16x8*>(w_ + 256);
            }
            if (!R_ABL(1)) { PC_PRODUCTS(acc, fh[fs & 3], fm[fs & 3], fm[fs & 3], xh[c], xm[c], xm[c]) }
            if (s == 0) {                // the panel after this one -> its ring slot (fetched one panel ago)
                const int ws = (slot + 1 == NB) ? 0 : slot + 1;
                R_STORE_W(ws)
            }
            if (s == 1) { R_LOAD_NEXT() }    // and the one after that -> registers
            if (PREF && fs == 1 && tap + 1 == ntaps && u + 1 < u1) {
                // the next unit starts a new tile: its loads go out now -- BEHIND this panel's weight-tile store (the
                // memory counter retires in order: issued in front of it, the store's wait also waited for these)
                pre_tile = tile + 1;
                const int pixn = pre_tile * 128 + wave * 32 + r32;
                const unsigned avn = (pixn < a.M && !R_ABL(8)) ? (unsigned)(pixn * CIN + h * 8) * 4u + PP_ZPAD_FLOATS * 4u : (unsigned)(h * 32);
#pragma unroll
                for (int c = 0; c < (PREF ? NCH : 0); ++c) {
                    rn[c][0] = buf_load16(rs_in, avn, (unsigned)c * 64u);
                    rn[c][1] = buf_load16(rs_in, avn + 16u, (unsigned)c * 64u);
                }
            }
            if (s == 7) {
                slot = (slot + 1 == NB) ? 0 : slot + 1;
                pcur = pnxt;
                pnxt = cW + ((slot + 1 == NB) ? 0 : slot + 1) * PANEL;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (c == NCH - 1) {          // n-tile nt complete: ReLU [+ concat slice] [+ its share of the head GEMM]
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = relu_keep_nan(acc[r]);
                if (heads && !R_ABL(2)) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const float av[8] = {v[8 * g], v[8 * g + 1], v[8 * g + 2], v[8 * g + 3],
                                             v[8 * g + 4], v[8 * g + 5], v[8 * g + 6], v[8 * g + 7]};
                        bf16x8 yh, ym, yl;
                        split_bf16x3(av, yh, ym, yl);
                        const float* hW = cHW + (nt * 2 + g) * 2 * 32 * 8;
                        const bf16x8 gh = *reinterpret_cast<const bf16x8*>(hW);
                        const bf16x8 gm = *reinterpret_cast<const bf16x8*>(hW + 32 * 8);
                        PC_PRODUCTS(hacc, gh, gm, gm, yh, ym, yl)
                    }
                }
                if (dst != nullptr && ok) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(dst + nt * 32 + 8 * g) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
                }
                R_STAMP(4 + nt + 8 * (u - u0))
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (heads && ok && !R_ABL(4)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(hrow + 8 * g) = make_float4(hacc[4 * g], hacc[4 * g + 1], hacc[4 * g + 2], hacc[4 * g + 3]);
            if (a.cls_plane != nullptr) {   // register r of this lane is head column 8*(r/4) + 4h + r%4
                float* cp = a.cls_plane + orow * a.cls_ncol;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = 8 * (r >> 2) + 4 * h + (r & 3) - a.cls_col0;
                    if ((unsigned)col < (unsigned)a.cls_ncol) cp[col] = hacc[r];
                }
            }
        }
        R_STAMP(8 + 8 * (u - u0))
        if (++tap == ntaps) { tap = 0; ++tile; }
    }
#undef R_STORE_W
#undef R_LOAD_NEXT
#undef R_STAMP
#undef R_ABL
}

// runs where k_deconv_u would and the shape fits (cout == 128, cin 64 / 128 / 256); PP_DECONV_R=0 turns it off
static bool deconv_r_runs(const LayerDesc& L, int ablate) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PP_DECONV_R"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || (ablate & 8192)) return false;
    // kernel == stride 1 has one tap per tile: nothing is re-read or re-split, and k_deconv_u's two-chunk register
    // prefetch hides the tile changes better (deconv1 at B=64: 35.4 us against 37.1)
    return L.k > 1 && L.cout == 128 && (L.cin == 64 || L.cin == 128 || L.cin == 256);
}
template <int CIN>
static void launch_deconv_r(const GemmArgs& a, hipStream_t s) {
    const int ntiles = (a.M + 127) / 128;
    const long long U = (long long)ntiles * a.k * a.k;
    const int slots = 2 * g_num_cus;
    const int upw = (int)((U + slots - 1) / slots);                 // units per workgroup
    const int G = (int)((U + upw - 1) / upw);
    PP_LAUNCH("k_deconv_r", (k_deconv_r<CIN>), dim3((unsigned)G), dim3(256), 0, s, a, (int)U, upw);
}
#else
static bool deconv_r_runs(const LayerDesc&, int) { return false; }
#endif

// ---------------------------------------------------------------------------------------
// Split-K Conv2DTranspose (+ fused heads) for small maps: the four waves of a workgroup share one
// 32-pixel x NT tile and take every fourth K-chunk each (private weight tiles, no barrier inside the K
// loop; the input fragments of all of a wave's chunks are loaded up front).  The partial accumulators meet
// in LDS (fixed order of addition), wave n finishes channel tile n -- bias, ReLU, concat slice if it is
// kept, its 32 channels' share of the head GEMM -- and wave 0 adds the head shares to the head bias / the
// earlier branches' sums.  Same operand layout as k_deconv_u (accumulator = out^T, heads from registers).
template <int NT>
__global__ __launch_bounds__(256, 2) void k_deconv_k4(GemmArgs a) {
    constexpr int KCH = 16;
    constexpr int NTILES = NT / 32;
    constexpr int SB = PP_NPIECE * NT * 8;                       // one weight tile: [3 pieces][NT][8 floats]
    constexpr int NBL = NT * 2 * PP_NPIECE / 64;                     // 16-byte weight items per lane per chunk
    constexpr int SHW = (NT / 16) * PP_NPIECE * 32 * 8;
    constexpr int SLOT = 16 * 64;                        // one partial accumulator tile (floats)
    constexpr int OWNERS = NTILES < 4 ? NTILES : 4;
    // the partial sums overlay the weight tiles (and need more room than two-piece weight tiles give)
    constexpr int SR = (4 * SB > (4 * NTILES - OWNERS) * SLOT) ? 4 * SB : (4 * NTILES - OWNERS) * SLOT;
    __shared__ __attribute__((aligned(16))) float sR[SR];
    __shared__ __attribute__((aligned(16))) float sHW[SHW];
    __shared__ float s_bias[NT];
    __shared__ float s_hbias[PP_HEAD_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, r32 = lane & 31;
    const int n0 = blockIdx.y * NT;
    const int cin = a.cin;
    const int niter = cin / (KCH * 4);                   // chunks per wave: 1..4 (checked by the launcher)
    const int tap = n0 / a.cout, cbase = n0 - tap * a.cout;
    const int ti = tap / a.k;
    const int delta = ti * (a.px_w * a.k) + (tap - ti * a.k);
    const bool heads = a.head_mode != 0;
    float* const sBw = sR + wave * SB;

    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(reinterpret_cast<const char*>(a.in) - PP_ZPAD_FLOATS * 4);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(a.wt16);
    const int pix = blockIdx.x * 32 + r32;
    const bool ok = pix < a.M;
    const unsigned avo = ok ? (unsigned)(pix * cin + h * 8) * 4u + PP_ZPAD_FLOATS * 4u : (unsigned)(h * 32);
    // all input fragments of this wave's chunks: one memory round trip
    float4 ra[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ra[j][0] = ra[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < niter) {
            const unsigned so = (unsigned)(wave + 4 * j) * (KCH * 4);
            ra[j][0] = buf_load16(rs_in, avo, so);
            ra[j][1] = buf_load16(rs_in, avo + 16u, so);
        }
    }
    unsigned boff[NBL];
    int bdst[NBL];
#pragma unroll
    for (int r = 0; r < NBL; ++r) {
        const int e = lane + 64 * r;
        const int piece = e / (NT * 2), rem = e % (NT * 2), row = rem >> 1, half = rem & 1;
        boff[r] = (unsigned)(((piece * a.n_total + n0 + row) * 16 + half * 8) * 2);
        bdst[r] = piece * (NT * 8) + row * 8 + ((half ^ ((row >> 3) & 1)) * 4);
    }
    const unsigned bstep = (unsigned)(PP_NPIECE * a.n_total * 32);
    float4 rb[NBL];
#pragma unroll
    for (int r = 0; r < NBL; ++r) rb[r] = buf_load16(rs_wt, boff[r], (unsigned)wave * bstep);
    // bias / head weights -> LDS after the operand loads are in flight (visible at the first barrier below)
    if (tid < NT) s_bias[tid] = a.bias[cbase + tid];
    if (heads) {
        for (int e = tid; e < SHW / 4; e += 256)
            reinterpret_cast<float4*>(sHW)[e ^ ((e >> 4) & 1)] = reinterpret_cast<const float4*>(a.head_wt16)[e];
        if (tid < PP_HEAD_COLS) s_hbias[tid] = a.head_bias[tid];
    }
    f32x16 acc[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const float* const cB = sBw + r32 * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < niter) {                                 // uniform
            // weight tile of chunk j -> this wave's LDS tile (a wave's LDS operations execute in order: the
            // previous chunk's fragment reads were issued before these writes)
#pragma unroll
            for (int r = 0; r < NBL; ++r) *reinterpret_cast<float4*>(sBw + bdst[r]) = rb[r];
            if (j + 1 < niter) {
#pragma unroll
                for (int r = 0; r < NBL; ++r) rb[r] = buf_load16(rs_wt, boff[r], (unsigned)(wave + 4 * (j + 1)) * bstep);
            }
            const float av[8] = {ra[j][0].x, ra[j][0].y, ra[j][0].z, ra[j][0].w, ra[j][1].x, ra[j][1].y, ra[j][1].z, ra[j][1].w};
            bf16x8 ah, am, al;
            split_bf16x3(av, ah, am, al);
#pragma unroll
            for (int n = 0; n < NTILES; ++n) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(cB + n * 32 * 8);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(cB + NT * 8 + n * 32 * 8);
                [[maybe_unused]] const bf16x8 bl = PC_LO(cB + 2 * NT * 8 + n * 32 * 8);
                PC_PRODUCTS(acc[n], bh, bm, bl, ah, am, al)
            }
        }
    }
    // ---- partial sums: every tile a wave does not own goes to LDS ----
    // slot of (source wave w, tile n), n != w, in row-major order with the owners' own tiles left out
    auto slot = [](int w, int n) -> int {
        return w * NTILES + n - (w < OWNERS ? w : OWNERS) - ((w < OWNERS && n > w) ? 1 : 0);
    };
    __syncthreads();   // all waves are done with their weight tiles (and bias / head weights are visible)
#pragma unroll
    for (int n = 0; n < NTILES; ++n)
        if (n != wave) {
            float* d = sR + slot(wave, n) * SLOT + lane * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(d + g * 256) = make_float4(acc[n][4 * g], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]);
        }
    __syncthreads();
    // output pixel of this lane's input pixel
    const int pc = ok ? pix : 0;
    const int hwpx = a.px_h * a.px_w;
    const int pb = pc / hwpx, prem = pc - pb * hwpx;
    const int py = prem / a.px_w, pxx = prem - py * a.px_w;
    const size_t orow = (size_t)((pb * a.px_h * a.k + py * a.k) * (a.px_w * a.k) + pxx * a.k + delta);
    f32x16 hpart;
#pragma unroll
    for (int r = 0; r < 16; ++r) hpart[r] = 0.f;
    if (wave < NTILES) {
        // own partial first, then the other waves' in ascending order: a fixed order of addition
        float v[16];
#pragma unroll
        for (int n = 0; n < NTILES; ++n)
            if (n == wave) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[n][r];
            }
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w != wave) {
                const float* sp = sR + slot(w, wave) * SLOT + lane * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = *reinterpret_cast<const float4*>(sp + g * 256);
                    v[4 * g] += t.x; v[4 * g + 1] += t.y; v[4 * g + 2] += t.z; v[4 * g + 3] += t.w;
                }
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = relu_keep_nan(v[r] + s_bias[wave * 32 + DCH(r, h)]);
        if (a.out != nullptr && ok) {
            float* dst = a.out + orow * a.ld_out + a.co_off + cbase + 4 * h + wave * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        }
        if (heads) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float av[8] = {v[8 * g], v[8 * g + 1], v[8 * g + 2], v[8 * g + 3],
                                     v[8 * g + 4], v[8 * g + 5], v[8 * g + 6], v[8 * g + 7]};
                bf16x8 xh, xm, xl;
                split_bf16x3(av, xh, xm, xl);
                const float* hW = sHW + ((wave * 2 + g) * PP_NPIECE * 32 + r32) * 8 + ((h ^ ((r32 >> 3) & 1)) * 4);
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(hW);
                const bf16x8 wm = *reinterpret_cast<const bf16x8*>(hW + 32 * 8);
                [[maybe_unused]] const bf16x8 wl = PC_LO(hW + 2 * 32 * 8);
                PC_PRODUCTS(hpart, wh, wm, wl, xh, xm, xl)
            }
        }
    }
    if (!heads) return;                                  // uniform
    __syncthreads();   // the partial-sum slots have been read
    if (wave >= 1 && wave < NTILES) {
        float* d = sR + (wave - 1) * SLOT + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(d + g * 256) = make_float4(hpart[4 * g], hpart[4 * g + 1], hpart[4 * g + 2], hpart[4 * g + 3]);
    }
    __syncthreads();
    if (wave == 0 && ok) {
        // head row of this pixel: this lane owns columns 4h + {0..3, 8..11, 16..19, 24..27}
        float* hrow = a.head + orow * PP_HEAD_COLS + 4 * h;
        float hv[16];
        if (a.head_mode == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(hrow + 8 * g);
                hv[4 * g] = t.x; hv[4 * g + 1] = t.y; hv[4 * g + 2] = t.z; hv[4 * g + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = s_hbias[DCH(r, h)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] += hpart[r];
#pragma unroll
        for (int w = 1; w < NTILES; ++w) {
            const float* sp = sR + (w - 1) * SLOT + lane * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(sp + g * 256);
                hv[4 * g] += t.x; hv[4 * g + 1] += t.y; hv[4 * g + 2] += t.z; hv[4 * g + 3] += t.w;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(hrow + 8 * g) = make_float4(hv[4 * g], hv[4 * g + 1], hv[4 * g + 2], hv[4 * g + 3]);
        if (a.cls_plane != nullptr) {   // compact class-logit plane (see GemmArgs::cls_plane)
            float* cp = a.cls_plane + orow * a.cls_ncol;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = 8 * (r >> 2) + 4 * h + (r & 3) - a.cls_col0;
                if ((unsigned)col < (unsigned)a.cls_ncol) cp[col] = hv[r];
            }
        }
    }
}

// runs while the layer has at most 1.5 workgroups per CU (measured crossover against k_deconv_u on cfg-A,
// tools/k4_sweep.sh: ahead at B = 1, 2 (160 / 320 workgroups), behind from B = 4); PP_DECONV_K4=0: never,
// =n: up to n per CU
static bool deconv_k4_runs(const LayerDesc& L, long long M, int ablate) {
    static int force = -2;
    if (force == -2) { const char* e = getenv("PP_DECONV_K4"); force = e ? atoi(e) : -1; }
    if (force == 0 || (ablate & 16) || L.cin % 64 != 0 || L.cin > 256) return false;
    const int nt = L.cout % 128 == 0 ? 128 : (L.cout % 64 == 0 ? 64 : 32);
    const int half_cus = force > 0 ? 2 * force : 3;
    return 2 * ((M + 31) / 32 * (L.n_total / nt)) <= (long long)half_cus * g_num_cus;
}
template <int NT>
static void launch_deconv_k4(const GemmArgs& a, int n_total, hipStream_t s) {
    dim3 grid((unsigned)((a.M + 31) / 32), n_total / NT);
    PP_LAUNCH("k_deconv_k4", (k_deconv_k4<NT>), grid, dim3(256), 0, s, a);
}

template <int NT>
static void launch_deconv_u(const GemmArgs& a, int n_total, hipStream_t s) {
    constexpr int WPS = 3;
    const int ntiles = (a.M + 127) / 128;
    const int ny = n_total / NT;
    int slots = (g_num_cus * WPS) / ny;
    if (slots < 8) slots = 8;
    int gx = ntiles < slots ? ntiles : slots;
    gx = (gx + 7) & ~7;
    dim3 grid((unsigned)gx, ny);
    PP_LAUNCH("k_deconv_u", (k_deconv_u<NT, WPS>), grid, dim3(256), 0, s, a, ntiles);
}

template <int NT, int MODE>
static void launch_t(const GemmArgs& a, int n_total, hipStream_t s) {
    const unsigned mt = (unsigned)((a.M + PX_TILE - 1) / PX_TILE);
    dim3 grid(mt, n_total / NT);
    PP_LAUNCH("k_gemm_layer", (k_gemm_layer<NT, MODE>), grid, dim3(256), 0, s, a);
}

// 128-pixel tiles (8-wave workgroups: every SIMD hosts one consumer and one producer wave) are the
// default.  64-pixel tiles (4-wave workgroups) are used only when a launch would otherwise not even
// cover the chip's CUs once (small batches): measured at B=64 they are slower -- a 4-wave
// workgroup's two consumer waves land on two of the four SIMDs, so the matrix pipes are unevenly fed.
static bool ws_small_tile(long long M, int n_total, int NT) {
    const long long tiles128 = ((M + 127) / 128) * (n_total / NT);
    return tiles128 < 256;
}

template <int NT, int MODE, int S>
static void launch_ws(const GemmArgs& a, int n_total, hipStream_t s) {
    if (ws_small_tile(a.M, n_total, NT)) {
        dim3 grid((unsigned)((a.M + 63) / 64), n_total / NT);
        PP_LAUNCH("k_gemm_ws", (k_gemm_ws<NT, MODE, S, 64>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((unsigned)((a.M + 127) / 128), n_total / NT);
        PP_LAUNCH("k_gemm_ws", (k_gemm_ws<NT, MODE, S, 128>), grid, dim3(512), 0, s, a);
    }
}

// separable layers: uniform-wave kernel (k_sep_u) or the producer/consumer kernel (k_gemm_ws).
// k_sep_u is the default; PP_SEP_KERNEL=ws selects the other; pp_bench_layer's ablation bits 16 / 32
// force u / ws.
static bool sep_uniform(int ablate) {
    static int dflt = -1;
    if (dflt < 0) {
        const char* e = getenv("PP_SEP_KERNEL");
        dflt = (e && e[0] == 'w') ? 0 : 1;
    }
    if (ablate & 16) return true;
    if (ablate & 32) return false;
    return dflt == 1;
}

// channel-tile width of the uniform-wave kernel: the widest that divides cout, narrowed to 64 when
// the layer would otherwise put fewer than ~2.5 waves on a SIMD (small maps: a wave per 32 pixels x NT
// channels is the scheduling grain; the depthwise is recomputed per channel tile, which the idle
// vector pipe absorbs)
static int sep_u_nt(const LayerDesc& L, int batch) {
    if (L.cout % 64 != 0) return 32;
    if (L.cout % 128 != 0) return 64;
    const long long waves128 = ((long long)batch * L.out_h * L.out_w + 31) / 32 * (L.cout / 128);
    static int force = -1;
    if (force < 0) { const char* e = getenv("PP_SEP_NT"); force = e ? atoi(e) : 0; }
    if (force == 64 || force == 128) return force;
    // split-precision path: the depthwise (VALU) is the larger half of a chunk, so recomputing it per
    // channel tile costs more than the thinner wave supply (measured: block3 at B=64 36 vs 40 us)
    if (L.d_wt16 != nullptr && split_precision(0)) return 128;
    return (waves128 * 2 < 5ll * g_num_cus * 4) ? 64 : 128;
}

long long* g_stamps = nullptr;   // tuning aid: device buffer for in-kernel stamps (pp_bench_layer, ablate & 64)

// workgroups per CU of the split-precision k_sep_u<128, 1, ...> instantiation (its register budget: 256 / 168)
#ifndef PP_SEP128_WPB
#define PP_SEP128_WPB 2
#endif
#ifndef PP_SEP64_WPB
#define PP_SEP64_WPB 3
#endif

// which kernel runs layer L (shared by the launcher and the profiler tags)
static bool use_ws(const LayerDesc& L) {
    // MODE 1 producers own 4 consecutive pixels, MODE 0 producers 2 of one row: the frame's pixel
    // count (hence every batch's) must be divisible accordingly, else the generic kernel runs
    if (L.kind != LAYER_SEP) return (L.in_h * L.in_w) % 4 == 0;
    return (L.stride == 1 || L.stride == 2) && (L.out_w % 2 == 0);
}

// the split-precision uniform-wave deconv kernel runs when its operands exist (cin % 32 == 0: an even
// number of 16-channel chunks) and, with fused heads, when one column tile covers the tap (NT == cout)
static bool deconv_uniform(const LayerDesc& L, int ablate) {
    if (L.kind != LAYER_DECONV || L.d_wt16 == nullptr || L.cin % 32 != 0 || !split_precision(ablate)) return false;
    if (ablate & 32) return false;
    if (L.head_mode != 0 && (L.d_head_wt16 == nullptr || !(L.cout == 32 || L.cout == 64 || L.cout == 128))) return false;
    return true;
}

// can layer L (a separable layer) read a sparse canvas at this batch size?  (the kernels with the cell-map lookup)
bool sparse_input_supported(const LayerDesc& L, int batch) {
    return L.kind == LAYER_SEP && use_ws(L) && sep_uniform(0) && L.d_wt16 != nullptr && split_precision(0) &&
           (long long)batch * L.out_h * L.out_w < (1 << 24);
}

// a deconv can carry the fused head GEMM when one workgroup column covers the tap's whole channel
// range (NT == cout) and the wave-specialised kernel runs it
bool deconv_can_fuse_heads(const LayerDesc& L) {
    return L.kind == LAYER_DECONV && (L.cout == 32 || L.cout == 64 || L.cout == 128) && use_ws(L);
}

static long long layer_rows(const LayerDesc& L, int batch) {
    return (L.kind == LAYER_SEP) ? (long long)batch * L.out_h * L.out_w : (long long)batch * L.in_h * L.in_w;
}

// name of the template instantiation that runs layer L at this batch size (profiler tags)
std::string layer_kernel_name(const LayerDesc& L, int batch) {
    const int nt = (L.kind == LAYER_HEAD) ? 32 : (L.cout % 128 == 0 ? 128 : (L.cout % 64 == 0 ? 64 : 32));
    const int mode = (L.kind == LAYER_SEP) ? 0 : 1;
    char buf[64];
    if (L.kind == LAYER_SEP && use_ws(L) && sep_uniform(0) && layer_rows(L, batch) < (1 << 24)) {
        if (sep_k4_runs(L.d_wt16, L.cin, L.n_total, layer_rows(L, batch), 0)) {
            if (sep_k8_runs(L.cin, L.n_total, layer_rows(L, batch))) snprintf(buf, sizeof(buf), "k_sep_k4<64,%d,8>", L.stride);
            else snprintf(buf, sizeof(buf), "k_sep_k4<64,%d>", L.stride);
            return std::string(buf);
        }
        if (sep_p_runs(L.d_wt16, L.stride, L.cin, L.cout, L.n_total, layer_rows(L, batch), L.d_occ, 0)) return std::string("k_sep_p");
        const int unt = sep_u_nt(L, batch);
        const bool bf = L.d_wt16 != nullptr && split_precision(0);
        int wps;   // workgroups per CU of the instantiation launch_layer picks (launch_u<NT, S, WPS, WPB>)
        if (L.stride == 1) wps = bf ? (unt == 128 ? PP_SEP128_WPB : (unt == 64 ? PP_SEP64_WPB : 3)) : (unt == 128 ? 3 : 4);
        else wps = bf ? (unt == 128 ? 2 : 3) : (unt == 128 ? 2 : (unt == 64 ? 3 : 4));
        snprintf(buf, sizeof(buf), "k_sep_u<%d,%d,%d,%d,%d>", unt, L.stride, wps, bf ? 1 : 0, (bf && L.d_occ) ? 1 : 0);
    } else if (deconv_uniform(L, 0) && deconv_k4_runs(L, layer_rows(L, batch), 0)) {
        snprintf(buf, sizeof(buf), "k_deconv_k4<%d>", nt);
    } else if (deconv_uniform(L, 0) && deconv_r_runs(L, 0)) {
        snprintf(buf, sizeof(buf), "k_deconv_r<%d>", L.cin);
    } else if (deconv_uniform(L, 0)) {
        snprintf(buf, sizeof(buf), "k_deconv_u<%d,3>", nt);
    } else if (use_ws(L)) {
        const int pxb = ws_small_tile(layer_rows(L, batch), L.n_total, nt) ? 64 : 128;
        snprintf(buf, sizeof(buf), "k_gemm_ws<%d,%d,%d,%d>", nt, mode, mode == 0 ? L.stride : 1, pxb);
    } else {
        snprintf(buf, sizeof(buf), "k_gemm_layer<%d,%d>", nt, mode);
    }
    return std::string(buf);
}

// does layer L (the last fused-head deconv) leave the compact class-logit plane?  (the uniform-wave and
// split-K deconv kernels do; the fallback generations do not and the post-process then scans the head rows)
bool layer_writes_cls_plane(const LayerDesc& L) {
    return L.kind == LAYER_DECONV && L.head_mode == 2 && L.d_cls_plane != nullptr && deconv_uniform(L, 0);
}

// frame0 > 0 (separable layers on k_sep_u only, launch_layer_subrange_ok): the launch covers the frames [frame0, frame0 +
// batch) of a larger batch -- same buffers, same pixel numbering, a sub-range of the tiles
bool launch_layer_subrange_ok(const LayerDesc& L, int frame0, int batch, int total_batch) {
    if (L.kind != LAYER_SEP) return false;
    const long long hw = (long long)L.out_h * L.out_w;
    if ((frame0 * hw) % PX_TILE != 0) return false;
    return layer_kernel_name(L, batch).compare(0, 8, "k_sep_u<") == 0 && layer_kernel_name(L, total_batch).compare(0, 8, "k_sep_u<") == 0;
}

int launch_layer(const LayerDesc& L, int batch, float* d_head, hipStream_t s, int ablate, int frame0) {
    if (batch <= 0) return 0;
    if (L.cin % KC != 0) return PP_ERR_UNSUPPORTED;
    {
        const long long m_sep = (long long)batch * L.out_h * L.out_w, m_in = (long long)batch * L.in_h * L.in_w;
        if (m_sep >= (1ll << 31) - PX_TILE || m_in >= (1ll << 31) - PX_TILE ||
            (long long)batch * L.out_h * L.out_w * L.k * L.k >= (1ll << 31)) return PP_ERR_UNSUPPORTED;
    }
    GemmArgs a;
    a.dbg = ablate;
    a.stamps = g_stamps;
    a.tile_lo = 0;
    a.in = L.in; a.dw = L.d_dw; a.wt = L.d_wt; a.bias = L.d_bias; a.out = L.out;
    a.wt16 = reinterpret_cast<const unsigned short*>(L.d_wt16); a.n_total = L.n_total;
    a.head = d_head; a.head_wt = L.d_head_wt; a.head_bias = L.d_head_bias; a.head_mode = L.head_mode;
    a.head_wt16 = L.d_head_wt16;
    a.in_h = L.in_h; a.in_w = L.in_w; a.cin = L.cin;
    a.stride = L.stride; a.ld_out = L.ld_out; a.co_off = L.co_off;
    a.k = L.k; a.cout = L.cout;
    a.occ = L.d_occ; a.occ_nz = L.occ_nz;
    a.occbits = L.d_occbits; a.occ_w64 = occ_words(L.in_w);
    a.tr_coef = nullptr; a.tr_D = nullptr; a.tr_stat = nullptr;
    a.cls_plane = layer_writes_cls_plane(L) ? L.d_cls_plane : nullptr;
    a.cls_col0 = L.cls_col0; a.cls_ncol = L.cls_ncol;
    if (L.kind == LAYER_SEP) {
        a.px_h = L.out_h; a.px_w = L.out_w; a.epi = 0;
        a.M = (frame0 + batch) * L.out_h * L.out_w;
        if (frame0 > 0) {
            if (!launch_layer_subrange_ok(L, frame0, batch, frame0 + batch)) return PP_ERR_UNSUPPORTED;
            a.tile_lo = (int)(((long long)frame0 * L.out_h * L.out_w) / PX_TILE);
        }
        if (L.cout % 32 != 0) return PP_ERR_UNSUPPORTED;
        // a sparse input is only understood by the split-precision uniform-wave / split-K kernels
        if (a.occ != nullptr && !(sparse_input_supported(L, batch) && split_precision(ablate) && sep_uniform(ablate)))
            return PP_ERR_UNSUPPORTED;
        if (use_ws(L) && sep_uniform(ablate) && a.M < (1 << 24)) {   // k_sep_u's float-reciprocal index math
            const int nt = sep_u_nt(L, batch);
            const long long msel = (long long)batch * L.out_h * L.out_w;    // rows of THIS launch (a.M is the end of its pixel range)
#if PP_SPLIT_MODE != 0
            if (!sep_k4_runs(a.wt16, a.cin, L.n_total, msel, ablate) &&
                sep_p_runs(a.wt16, L.stride, a.cin, L.cout, L.n_total, msel, a.occ, ablate)) {   // depthwise once for 256 channels
                launch_p(a, s);
            } else
#endif
            if (!(ablate & 16) && sep_k4_runs(a.wt16, a.cin, L.n_total, msel, ablate)) {   // small map
                if (L.stride == 1) launch_k4<1>(a, L.n_total, s);
                else launch_k4<2>(a, L.n_total, s);
            } else if (L.stride == 1) {
                if (nt == 128) launch_u<128, 1, 3, PP_SEP128_WPB>(a, L.n_total, s);
                else if (nt == 64) launch_u<64, 1, 4, PP_SEP64_WPB>(a, L.n_total, s);
                else launch_u<32, 1, 4, 3>(a, L.n_total, s);
            } else {
                if (nt == 128) launch_u<128, 2, 2, 2>(a, L.n_total, s);
                else if (nt == 64) launch_u<64, 2, 3, 3>(a, L.n_total, s);
                else launch_u<32, 2, 4, 3>(a, L.n_total, s);
            }
        } else if (use_ws(L)) {
            if (L.stride == 1) {
                if (L.cout % 128 == 0) launch_ws<128, 0, 1>(a, L.n_total, s);
                else if (L.cout % 64 == 0) launch_ws<64, 0, 1>(a, L.n_total, s);
                else launch_ws<32, 0, 1>(a, L.n_total, s);
            } else {
                if (L.cout % 128 == 0) launch_ws<128, 0, 2>(a, L.n_total, s);
                else if (L.cout % 64 == 0) launch_ws<64, 0, 2>(a, L.n_total, s);
                else launch_ws<32, 0, 2>(a, L.n_total, s);
            }
        } else {
            if (L.cout % 128 == 0) launch_t<128, 0>(a, L.n_total, s);
            else if (L.cout % 64 == 0) launch_t<64, 0>(a, L.n_total, s);
            else launch_t<32, 0>(a, L.n_total, s);
        }
    } else if (L.kind == LAYER_DECONV) {
        a.px_h = L.in_h; a.px_w = L.in_w; a.epi = 1;
        a.M = batch * L.in_h * L.in_w;
        if (L.cout % 32 != 0) return PP_ERR_UNSUPPORTED;
        if (deconv_uniform(L, ablate) && deconv_k4_runs(L, a.M, ablate)) {   // small map: split-K
            if (L.cout % 128 == 0) launch_deconv_k4<128>(a, L.n_total, s);
            else if (L.cout % 64 == 0) launch_deconv_k4<64>(a, L.n_total, s);
            else launch_deconv_k4<32>(a, L.n_total, s);
#if PP_SPLIT_MODE != 0
        } else if (deconv_uniform(L, ablate) && deconv_r_runs(L, ablate)) {   // input resident in registers
            if (L.cin == 256) launch_deconv_r<256>(a, s);
            else if (L.cin == 128) launch_deconv_r<128>(a, s);
            else launch_deconv_r<64>(a, s);
#endif
        } else if (deconv_uniform(L, ablate)) {
            if (L.cout % 128 == 0) launch_deconv_u<128>(a, L.n_total, s);
            else if (L.cout % 64 == 0) launch_deconv_u<64>(a, L.n_total, s);
            else launch_deconv_u<32>(a, L.n_total, s);
        } else if (use_ws(L)) {
            if (L.cout % 128 == 0) launch_ws<128, 1, 1>(a, L.n_total, s);
            else if (L.cout % 64 == 0) launch_ws<64, 1, 1>(a, L.n_total, s);
            else launch_ws<32, 1, 1>(a, L.n_total, s);
        } else {
            if (L.cout % 128 == 0) launch_t<128, 1>(a, L.n_total, s);
            else if (L.cout % 64 == 0) launch_t<64, 1>(a, L.n_total, s);
            else launch_t<32, 1>(a, L.n_total, s);
        }
    } else {
        a.px_h = L.in_h; a.px_w = L.in_w; a.epi = 2;
        a.M = batch * L.in_h * L.in_w;
        if (L.n_total != 32) return PP_ERR_UNSUPPORTED;
        if (use_ws(L)) launch_ws<32, 1, 1>(a, L.n_total, s);
        else launch_t<32, 1>(a, L.n_total, s);
    }
    return 0;
}
