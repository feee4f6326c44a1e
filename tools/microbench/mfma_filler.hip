// Microbenchmark: VALU fillers inside one wave's MFMA stream (one wave per SIMD).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_filler tools/microbench/mfma_filler.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NF>
__global__ __launch_bounds__(256) void k_fill(int iters, float seed, float* out) {
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = seed; acc1[r] = seed; acc2[r] = seed; acc3[r] = seed; }
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(seed + j); bb[j] = (__bf16)(seed - j); }
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = seed + j;
    const float m = 1.0000001f, c = 1e-9f;
#define FILL(base) _Pragma("unroll") for (int u = 0; u < NF; ++u) x[(base + u) & 15] = fmaf(x[(base + u) & 15], m, c);
#define MF(acc) if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc, 0, 0, 0); \
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
        MF(acc0) FILL(0)
        __builtin_amdgcn_sched_barrier(0);
        MF(acc1) FILL(4)
        __builtin_amdgcn_sched_barrier(0);
        MF(acc2) FILL(8)
        __builtin_amdgcn_sched_barrier(0);
        MF(acc3) FILL(12)
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r] + x[r];
    if (s == 12345.678f) *out = s;
}

template <int KIND, int NF>
static void run(float* d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    hipLaunchKernelGGL((k_fill<KIND, NF>), dim3(256), dim3(256), 0, 0, iters, 1.0f, d);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_fill<KIND, NF>), dim3(256), dim3(256), 0, 0, iters, 1.0f, d);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double cyc = ms / 5 * 1e-3 * 2.4e9 / (iters * 4.0);
    printf("%s  fillers/MFMA %2d : %.1f us, %.1f cycles per MFMA slot\n", KIND ? "bf16 32x32x16" : "f32  32x32x2 ", NF, ms / 5 * 1e3, cyc);
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    run<0, 0>(d); run<0, 2>(d); run<0, 4>(d); run<0, 8>(d); run<0, 12>(d); run<0, 16>(d);
    run<1, 0>(d); run<1, 2>(d); run<1, 4>(d); run<1, 6>(d); run<1, 8>(d); run<1, 12>(d);
    return 0;
}
