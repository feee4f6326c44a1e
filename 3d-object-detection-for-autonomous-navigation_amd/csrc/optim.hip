// AdamW parameter update (the reference's optimizer step, train.py:228-239, :301).
//
// The reference uses tfa.optimizers.AdamW (tensorflow-addons 0.11.2, configs/pip/requirements_short.txt:4) over
// tf.keras.optimizers.Adam (TensorFlow 2.2.0); neither is vendored.  Their published update for one variable:
//   var -= weight_decay * var                          (decoupled decay, applied first, NOT scaled by the rate)
//   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2
//   var -= lr_t * m / (sqrt(v) + epsilon),   lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t),  t = iterations + 1
// with lr from the ExponentialDecay schedule (host side, optim.py).  One flat float32 buffer per state:
// HBM-bound, 16 bytes read + 12 written per parameter, 16-byte accesses.
#include "pp_common.h"

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ w, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v, int64_t n,
                                               float lr_t, float beta1, float beta2, float eps, float wd) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    float wv[4], gv[4], mv[4], vv[4];
    const bool full = i4 + 4 <= n;
    if (full) {
        const float4 a = *reinterpret_cast<const float4*>(w + i4), b = *reinterpret_cast<const float4*>(g + i4);
        const float4 c = *reinterpret_cast<const float4*>(m + i4), d = *reinterpret_cast<const float4*>(v + i4);
        wv[0] = a.x; wv[1] = a.y; wv[2] = a.z; wv[3] = a.w;
        gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
        mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
        vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
        for (int k = 0; k < 4; ++k) {
            const bool ok = i4 + k < n;
            wv[k] = ok ? w[i4 + k] : 0.f; gv[k] = ok ? g[i4 + k] : 0.f;
            mv[k] = ok ? m[i4 + k] : 0.f; vv[k] = ok ? v[i4 + k] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wd_w = __fsub_rn(wv[k], __fmul_rn(wd, wv[k]));
        mv[k] = __fadd_rn(__fmul_rn(beta1, mv[k]), __fmul_rn(1.f - beta1, gv[k]));
        vv[k] = __fadd_rn(__fmul_rn(beta2, vv[k]), __fmul_rn(1.f - beta2, __fmul_rn(gv[k], gv[k])));
        wv[k] = __fsub_rn(wd_w, __fdiv_rn(__fmul_rn(lr_t, mv[k]), __fadd_rn(__fsqrt_rn(vv[k]), eps)));
    }
    if (full) {
        *reinterpret_cast<float4*>(w + i4) = make_float4(wv[0], wv[1], wv[2], wv[3]);
        *reinterpret_cast<float4*>(m + i4) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4*>(v + i4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
        for (int k = 0; k < 4; ++k)
            if (i4 + k < n) { w[i4 + k] = wv[k]; m[i4 + k] = mv[k]; v[i4 + k] = vv[k]; }
    }
}

void launch_adamw(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                  float eps, float wd, hipStream_t s) {
    if (n <= 0) return;
    const int64_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_adamw, dim3((unsigned)blocks), dim3(256), 0, s, w, g, m, v, n, lr_t, beta1, beta2, eps, wd);
}
