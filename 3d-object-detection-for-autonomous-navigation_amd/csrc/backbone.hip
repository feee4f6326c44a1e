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
#include "pp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PX_TILE 128
#define KC 32
#define LDS_STRIDE 36

struct GemmArgs {
    const float* in;
    const float* dw;
    const float* wt;
    const float* bias;
    float* out;
    float* box;
    float* cls;
    float* dir;
    long long M;          // GEMM rows (pixels)
    int in_h, in_w, cin;
    int px_h, px_w;       // pixel space of M (sep: output map; deconv/head: input map)
    int stride;
    int ld_out, co_off;
    int epi;              // 0: bias+ReLU rows; 1: deconv pixel-shuffle; 2: heads
    int k, cout;          // deconv
    int nb, nc, nd;       // head widths
};

template <int NT, int MODE>
__global__ __launch_bounds__(256) void k_gemm_layer(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[PX_TILE * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float sB[NT * LDS_STRIDE];
    constexpr int NTILES = NT / 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware bijective remap of the pixel-tile index
    int mt;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        mt = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const long long p0 = (long long)mt * PX_TILE;
    const int n0 = blockIdx.y * NT;
    const int cin = a.cin;

    // this thread's 4 staging items: pixel (tid>>3) + 32*it, channel group tid&7
    const int c4 = tid & 7;
    const float* ibase[4];
    int yi0[4], xi0[4];
    bool pvalid[4];
    {
        const long long hw = (long long)a.px_h * a.px_w;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long long pix = p0 + (tid >> 3) + 32 * it;
            pvalid[it] = pix < a.M;
            const long long pc = pvalid[it] ? pix : 0;
            const int b = (int)(pc / hw);
            const int rem = (int)(pc - (long long)b * hw);
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

    // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
    const int h = lane >> 5, col_l = lane & 31;
    if (a.epi == 0) {
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
            const int col = n0 + n * 32 + col_l;
            const float bv = a.bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long pix = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (pix < a.M) a.out[(size_t)pix * a.ld_out + a.co_off + col] = fmaxf(acc[n][r] + bv, 0.f);
            }
        }
    } else if (a.epi == 1) {
        // Conv2DTranspose, kernel == stride: out[y*k+i][x*k+j][co] = sum_ci in[y][x][ci] * K[i][j][co][ci]
        const int k = a.k, OW = a.px_w * k;
        const long long hw = (long long)a.px_h * a.px_w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long pix = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (pix >= a.M) continue;
            const int b = (int)(pix / hw);
            const int rem = (int)(pix - (long long)b * hw);
            const int y = rem / a.px_w, x = rem - y * a.px_w;
#pragma unroll
            for (int n = 0; n < NTILES; ++n) {
                const int col = n0 + n * 32 + col_l;
                const int tap = col / a.cout, co = col - tap * a.cout;
                const int i = tap / k, j = tap - i * k;
                const size_t opix = ((size_t)b * a.px_h * k + (size_t)y * k + i) * OW + (size_t)x * k + j;
                a.out[opix * a.ld_out + a.co_off + co] = fmaxf(acc[n][r] + a.bias[co], 0.f);
            }
        }
    } else {
        // heads: columns [0,nb) box, [nb,nb+nc) cls, [nb+nc,nb+nc+nd) dir; bias, no activation
        const int col = n0 + col_l;
        const float bv = a.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long pix = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (pix >= a.M) continue;
            const float v = acc[0][r] + bv;
            if (col < a.nb) a.box[(size_t)pix * a.nb + col] = v;
            else if (col < a.nb + a.nc) a.cls[(size_t)pix * a.nc + (col - a.nb)] = v;
            else if (col < a.nb + a.nc + a.nd) a.dir[(size_t)pix * a.nd + (col - a.nb - a.nc)] = v;
        }
    }
}

template <int NT, int MODE>
static void launch_t(const GemmArgs& a, int n_total, hipStream_t s) {
    const unsigned mt = (unsigned)((a.M + PX_TILE - 1) / PX_TILE);
    dim3 grid(mt, n_total / NT);
    hipLaunchKernelGGL((k_gemm_layer<NT, MODE>), grid, dim3(256), 0, s, a);
}

const char* layer_kernel_name(const LayerDesc& L) {
    const int nt = (L.kind == LAYER_HEAD) ? 32 : (L.cout % 128 == 0 ? 128 : (L.cout % 64 == 0 ? 64 : 32));
    if (L.kind == LAYER_SEP) return nt == 128 ? "k_gemm_layer<128,0>" : (nt == 64 ? "k_gemm_layer<64,0>" : "k_gemm_layer<32,0>");
    return nt == 128 ? "k_gemm_layer<128,1>" : (nt == 64 ? "k_gemm_layer<64,1>" : "k_gemm_layer<32,1>");
}

int launch_layer(const LayerDesc& L, int batch, float* d_box, float* d_cls, float* d_dir, int napl,
                 hipStream_t s) {
    if (batch <= 0) return 0;
    if (L.cin % KC != 0) return PP_ERR_UNSUPPORTED;
    GemmArgs a;
    a.in = L.in; a.dw = L.d_dw; a.wt = L.d_wt; a.bias = L.d_bias; a.out = L.out;
    a.box = d_box; a.cls = d_cls; a.dir = d_dir;
    a.in_h = L.in_h; a.in_w = L.in_w; a.cin = L.cin;
    a.stride = L.stride; a.ld_out = L.ld_out; a.co_off = L.co_off;
    a.k = L.k; a.cout = L.cout;
    a.nb = napl * 7; a.nc = napl; a.nd = napl * 2;
    if (L.kind == LAYER_SEP) {
        a.px_h = L.out_h; a.px_w = L.out_w; a.epi = 0;
        a.M = (long long)batch * L.out_h * L.out_w;
        if (L.cout % 128 == 0) launch_t<128, 0>(a, L.n_total, s);
        else if (L.cout % 64 == 0) launch_t<64, 0>(a, L.n_total, s);
        else if (L.cout % 32 == 0) launch_t<32, 0>(a, L.n_total, s);
        else return PP_ERR_UNSUPPORTED;
    } else if (L.kind == LAYER_DECONV) {
        a.px_h = L.in_h; a.px_w = L.in_w; a.epi = 1;
        a.M = (long long)batch * L.in_h * L.in_w;
        if (L.cout % 128 == 0) launch_t<128, 1>(a, L.n_total, s);
        else if (L.cout % 64 == 0) launch_t<64, 1>(a, L.n_total, s);
        else if (L.cout % 32 == 0) launch_t<32, 1>(a, L.n_total, s);
        else return PP_ERR_UNSUPPORTED;
    } else {
        a.px_h = L.in_h; a.px_w = L.in_w; a.epi = 2;
        a.M = (long long)batch * L.in_h * L.in_w;
        if (L.n_total != 32) return PP_ERR_UNSUPPORTED;
        launch_t<32, 1>(a, L.n_total, s);
    }
    return 0;
}
