// Rotated-box overlaps of the KITTI-style AP evaluator.
//
// Replaces rotate_iou_gpu_eval / rotate_iou_kernel_eval and their device functions
// (reference second/core/non_max_suppression/nms_gpu.py:180-415, :564-653: numba-CUDA) and
// d3_box_overlap (second/utils/eval.py:132-163):
//   out[n][k] = inter(query[k], box[n]) / {a1 + a2 - inter | a1 | a2 | 1}     criterion -1 | 0 | 1 | 2
// where inter() clips two rotated rectangles: corners of each inside the other (>= tests), the 16
// edge/edge intersections, an angular insertion sort about the centroid, a triangle fan.  Every
// operation is float32 in the reference's order (no contraction: the library is built with
// -ffp-contract=off); cos / sin / sqrt are evaluated in double and rounded, like math.cos on a
// float32 scalar.  The 3D overlap multiplies the BEV intersection by the height overlap in float64.
//
// Mapping: k_riou_corners turns every box into 8 corner floats + area once (N + K threads, the
// double-precision sincos is the expensive part); k_riou_pairs runs one thread per (n, k) pair,
// 64 consecutive k per wavefront so the [N][K] output rows are written in full 256-byte segments.
// The per-thread polygon (<= 8 points) and its sort keys live in LDS in a [slot][thread] layout
// (dynamic indexing of a private array would go to scratch memory; this layout is bank-conflict free).
// The reference's int_pts array holds 8 points; a pair that produced more (possible only through
// duplicated corner hits, e.g. identical boxes that also report edge crossings) would index out of
// bounds there -- here the extra points are dropped.
#include "pp_common.h"

#define RIOU_TX 64
#define RIOU_TY 4
#define RIOU_MAXP 8

__global__ __launch_bounds__(256) void k_riou_corners(const float* __restrict__ boxes, int64_t n, float* __restrict__ corners) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* r = boxes + 5 * i;
    const float a_cos = (float)cos((double)r[4]), a_sin = (float)sin((double)r[4]);
    const float cx = r[0], cy = r[1], xd = r[2], yd = r[3];
    const float px[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
    const float py[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
    float* c = corners + 9 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[2 * j] = __fadd_rn(__fadd_rn(__fmul_rn(a_cos, px[j]), __fmul_rn(a_sin, py[j])), cx);
        c[2 * j + 1] = __fadd_rn(__fadd_rn(__fmul_rn(-a_sin, px[j]), __fmul_rn(a_cos, py[j])), cy);
    }
    c[8] = __fmul_rn(xd, yd);
}

__device__ __forceinline__ bool riou_in_quad(float x, float y, const float (&c)[8]) {
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    const float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = x - c[0], ap1 = y - c[1];
    const float abab = __fadd_rn(__fmul_rn(ab0, ab0), __fmul_rn(ab1, ab1));
    const float abap = __fadd_rn(__fmul_rn(ab0, ap0), __fmul_rn(ab1, ap1));
    const float adad = __fadd_rn(__fmul_rn(ad0, ad0), __fmul_rn(ad1, ad1));
    const float adap = __fadd_rn(__fmul_rn(ad0, ap0), __fmul_rn(ad1, ap1));
    return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

// criterion: -1 IoU, 0 / query area, 1 / box area, 2 raw intersection area
__global__ __launch_bounds__(RIOU_TX * RIOU_TY) void k_riou_pairs(const float* __restrict__ bc, int64_t N,
                                                                  const float* __restrict__ qc, int64_t K,
                                                                  int criterion, float* __restrict__ out) {
    __shared__ float s_px[RIOU_MAXP][RIOU_TX * RIOU_TY];
    __shared__ float s_py[RIOU_MAXP][RIOU_TX * RIOU_TY];
    __shared__ float s_vs[RIOU_MAXP][RIOU_TX * RIOU_TY];
    const int t = threadIdx.y * RIOU_TX + threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * RIOU_TX + threadIdx.x;
    const int64_t n = (int64_t)blockIdx.y * RIOU_TY + threadIdx.y;
    if (k >= K || n >= N) return;
    float c1[8], c2[8];   // c1: query (first argument of devRotateIoUEval), c2: box
#pragma unroll
    for (int j = 0; j < 8; ++j) { c1[j] = qc[9 * k + j]; c2[j] = bc[9 * n + j]; }
    const float area1 = qc[9 * k + 8], area2 = bc[9 * n + 8];

    int np = 0;
#define RIOU_PUSH(X, Y) { if (np < RIOU_MAXP) { s_px[np][t] = (X); s_py[np][t] = (Y); } ++np; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (riou_in_quad(c1[2 * i], c1[2 * i + 1], c2)) RIOU_PUSH(c1[2 * i], c1[2 * i + 1])
        if (riou_in_quad(c2[2 * i], c2[2 * i + 1], c1)) RIOU_PUSH(c2[2 * i], c2[2 * i + 1])
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float A0 = c1[2 * i], A1 = c1[2 * i + 1], B0 = c1[2 * ((i + 1) & 3)], B1 = c1[2 * ((i + 1) & 3) + 1];
        const float BA0 = B0 - A0, BA1 = B1 - A1;
        const float ABBA = __fsub_rn(__fmul_rn(A0, B1), __fmul_rn(B0, A1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float C0 = c2[2 * j], C1 = c2[2 * j + 1], D0 = c2[2 * ((j + 1) & 3)], D1 = c2[2 * ((j + 1) & 3) + 1];
            const float DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
            const bool acd = __fmul_rn(DA1, CA0) > __fmul_rn(CA1, DA0);
            const bool bcd = __fmul_rn(D1 - B1, C0 - B0) > __fmul_rn(C1 - B1, D0 - B0);
            if (acd != bcd) {
                const bool abc = __fmul_rn(CA1, BA0) > __fmul_rn(BA1, CA0);
                const bool abd = __fmul_rn(DA1, BA0) > __fmul_rn(BA1, DA0);
                if (abc != abd) {
                    const float DC0 = D0 - C0, DC1 = D1 - C1;
                    const float CDDC = __fsub_rn(__fmul_rn(C0, D1), __fmul_rn(D0, C1));
                    const float DH = __fsub_rn(__fmul_rn(BA1, DC0), __fmul_rn(BA0, DC1));
                    const float Dx = __fsub_rn(__fmul_rn(ABBA, DC0), __fmul_rn(BA0, CDDC));
                    const float Dy = __fsub_rn(__fmul_rn(ABBA, DC1), __fmul_rn(BA1, CDDC));
                    RIOU_PUSH(__fdiv_rn(Dx, DH), __fdiv_rn(Dy, DH))
                }
            }
        }
    }
#undef RIOU_PUSH
    if (np > RIOU_MAXP) np = RIOU_MAXP;
    float area = 0.f;
    if (np > 0) {
        float cx = 0.f, cy = 0.f;
        for (int i = 0; i < np; ++i) { cx = __fadd_rn(cx, s_px[i][t]); cy = __fadd_rn(cy, s_py[i][t]); }
        cx = __fdiv_rn(cx, (float)np);
        cy = __fdiv_rn(cy, (float)np);
        for (int i = 0; i < np; ++i) {
            float v0 = s_px[i][t] - cx, v1 = s_py[i][t] - cy;
            const float d = (float)sqrt((double)__fadd_rn(__fmul_rn(v0, v0), __fmul_rn(v1, v1)));
            v0 = __fdiv_rn(v0, d);
            v1 = __fdiv_rn(v1, d);
            if (v1 < 0.f) v0 = -2.f - v0;
            s_vs[i][t] = v0;
        }
        for (int i = 1; i < np; ++i) {
            if (s_vs[i - 1][t] > s_vs[i][t]) {
                const float temp = s_vs[i][t], tx = s_px[i][t], ty = s_py[i][t];
                int j = i;
                while (j > 0 && s_vs[j - 1][t] > temp) {
                    s_vs[j][t] = s_vs[j - 1][t];
                    s_px[j][t] = s_px[j - 1][t];
                    s_py[j][t] = s_py[j - 1][t];
                    --j;
                }
                s_vs[j][t] = temp;
                s_px[j][t] = tx;
                s_py[j][t] = ty;
            }
        }
        const float a0 = s_px[0][t], a1 = s_py[0][t];
        for (int i = 0; i < np - 2; ++i) {
            const float b0 = s_px[i + 1][t], b1 = s_py[i + 1][t], q0 = s_px[i + 2][t], q1 = s_py[i + 2][t];
            const float tri = __fsub_rn(__fmul_rn(a0 - q0, b1 - q1), __fmul_rn(a1 - q1, b0 - q0)) / 2.0f;
            area = __fadd_rn(area, fabsf(tri));
        }
    }
    float v;
    if (criterion == -1) v = __fdiv_rn(area, __fsub_rn(__fadd_rn(area1, area2), area));
    else if (criterion == 0) v = __fdiv_rn(area, area1);
    else if (criterion == 1) v = __fdiv_rn(area, area2);
    else v = area;
    out[n * K + k] = v;
}

// d3_box_overlap_kernel (eval.py:132-156): camera boxes [x, y, z, l, h, w, ry] float64; rinc = BEV intersection
__global__ __launch_bounds__(256) void k_d3_finish(const double* __restrict__ boxes, int64_t N,
                                                   const double* __restrict__ qboxes, int64_t K, int criterion,
                                                   const float* __restrict__ rinc, double* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = blockIdx.y;
    if (k >= K) return;
    const double r = (double)rinc[n * K + k];
    double res = r;
    if (r > 0.0) {
        const double* b = boxes + 7 * n;
        const double* q = qboxes + 7 * k;
        const double iw = fmin(b[1], q[1]) - fmax(b[1] - b[4], q[1] - q[4]);
        if (iw > 0.0) {
            const double a1 = b[3] * b[4] * b[5], a2 = q[3] * q[4] * q[5];
            const double inc = iw * r;
            double ua;
            if (criterion == -1) ua = a1 + a2 - inc;
            else if (criterion == 0) ua = a1;
            else if (criterion == 1) ua = a2;
            else ua = 1.0;
            res = inc / ua;
        } else {
            res = 0.0;
        }
    }
    out[n * K + k] = res;
}

void launch_riou_corners(const float* boxes, int64_t n, float* corners, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_riou_corners, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, boxes, n, corners);
}

void launch_riou_pairs(const float* bc, int64_t N, const float* qc, int64_t K, int criterion, float* out, hipStream_t s) {
    if (N <= 0 || K <= 0) return;
    dim3 grid((unsigned)((K + RIOU_TX - 1) / RIOU_TX), (unsigned)((N + RIOU_TY - 1) / RIOU_TY));
    hipLaunchKernelGGL(k_riou_pairs, grid, dim3(RIOU_TX, RIOU_TY), 0, s, bc, N, qc, K, criterion, out);
}

void launch_d3_finish(const double* boxes, int64_t N, const double* qboxes, int64_t K, int criterion, const float* rinc,
                      double* out, hipStream_t s) {
    if (N <= 0 || K <= 0) return;
    dim3 grid((unsigned)((K + 255) / 256), (unsigned)N);
    hipLaunchKernelGGL(k_d3_finish, grid, dim3(256), 0, s, boxes, N, qboxes, K, criterion, rinc, out);
}
