// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (no MFMAs around), one / two / four waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
    const f2 m = {s, s * 0.5f}, c = {1e-3f, 2e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PK) a[i] = __builtin_elementwise_fma(a[i], m, c);
            else { a[i].x = __builtin_fmaf(a[i].x, m.x, c.x); a[i].y = __builtin_fmaf(a[i].y, m.y, c.y); }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;   // 4 waves per block, 256 CUs: wps blocks per CU = wps waves per SIMD
        for (int pk = 0; pk < 2; ++pk) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) k<1><<<blocks, 256>>>(d, iters, 0.999f); else k<0><<<blocks, 256>>>(d, iters, 0.999f);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double flops = (double)blocks * 256 * iters * 16 * 2;
            printf("waves/SIMD %d  %s: %.3f ms  %.1f TFLOP/s  (%.2f cyc per FMA-pair issue per wave at 2.4 GHz)\n", wps,
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0) / wps);
        }
    }
    return 0;
}
