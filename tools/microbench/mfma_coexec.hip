// Microbenchmark: does VALU f32 work on one wave overlap with MFMA work of its SIMD partner?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_coexec tools/microbench/mfma_coexec.hip && ./mfma_coexec
// 512-thread workgroups: waves w and w+4 share a SIMD.  Role A (waves 0-3) runs a chain of MFMAs,
// role B (waves 4-7) a chain of v_fma_f32.  Times: A alone, B alone, both.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 32x32x2, 1: bf16 32x32x16
__device__ __forceinline__ void mfma_chain(int iters, float seed, float* out) {
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = seed; acc1[r] = seed; acc2[r] = seed; acc3[r] = seed; }
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(seed + j); bb[j] = (__bf16)(seed - j); }
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc3, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    if (s == 12345.678f) *out = s;
}

__device__ __forceinline__ void valu_chain(int iters, float seed, float* out) {
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
            x4 = fmaf(x4, m, c); x5 = fmaf(x5, m, c); x6 = fmaf(x6, m, c); x7 = fmaf(x7, m, c);
        }
    }
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (s == 12345.678f) *out = s;
}

template <int KIND>
__global__ __launch_bounds__(512) void k_mix(int mfma_iters, int valu_iters, float seed, float* out) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) { if (mfma_iters) mfma_chain<KIND>(mfma_iters, seed, out); }
    else { if (valu_iters) valu_chain(valu_iters, seed, out); }
}

template <int KIND>
static float run(int mi, int vi, float* d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_mix<KIND>), dim3(256), dim3(512), 0, 0, mi, vi, 1.0f, d);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_mix<KIND>), dim3(256), dim3(512), 0, 0, mi, vi, 1.0f, d);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e3f;
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    const int MI = 4000;   // 16000 MFMAs per wave
    // f32 MFMA: 64 cycles each -> 16000 * 64 = 1.02 M cycles ~ 427 us;  bf16: 32 cycles -> 213 us
    for (int vi : {0, 1500, 3000, 6000}) {
        printf("f32 mfma: mfma-only %.1f us | valu-only(%d) %.1f us | both %.1f us\n", run<0>(MI, 0, d), vi,
               vi ? run<0>(0, vi, d) : 0.f, run<0>(MI, vi, d));
    }
    for (int vi : {0, 750, 1500, 3000}) {
        printf("bf16 mfma: mfma-only %.1f us | valu-only(%d) %.1f us | both %.1f us\n", run<1>(MI, 0, d), vi,
               vi ? run<1>(0, vi, d) : 0.f, run<1>(MI, vi, d));
    }
    return 0;
}
