// Fused multi-tensor Adam step (torch.optim.Adam semantics: coupled L2 weight decay, bias correction), one launch
// for every parameter tensor.  Replaces the optimizer.step() of fumi/models/fumi.py:193 (fumi/utils/utils.py:280-283:
// Adam(lr, weight_decay)), which torch runs as ~7 multi-tensor kernels.  Same operation order as torch's
// _single_tensor_adam so the parameters stay bit-comparable:
//   g += wd * p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g g ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// HBM-bound elementwise kernel: 16-byte accesses, 4 tensors' worth of traffic (p, g, m, v) per element.
#include "common.h"

namespace {

constexpr int MAXT = 32;
struct AdamTensors {
    float* p[MAXT]; const float* g[MAXT]; float* m[MAXT]; float* v[MAXT];
    long end[MAXT];          // cumulative element counts rounded up to 4 per tensor (in float4 units)
    long numel[MAXT];        // element count per tensor (by value: no per-step host-to-device copy)
    int n;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, long total4, float lr_over_bc1, float inv_sqrt_bc2,
                                                   float b1, float b2, float eps, float wd,
                                                   const float* __restrict__ pub_src, int pub_n, float* pub_dst,
                                                   unsigned long long pub_seq) {
    // rider: a deferred publication of the step's statistics (sampler.hip: publish_scalars_kernel) saves its own launch
    if (pub_dst && blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) {
        const int i = threadIdx.x;
        if (i < pub_n) __hip_atomic_store(pub_dst + i, pub_src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (i == 0) __hip_atomic_store((unsigned long long*)(pub_dst + 14), pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
        int k = 0;
        while (k + 1 < t.n && i4 >= t.end[k]) ++k;
        const long base4 = k ? t.end[k - 1] : 0;
        const long e0 = (i4 - base4) * 4;
        const long n = t.numel[k];
        float* p = t.p[k]; const float* g = t.g[k]; float* m = t.m[k]; float* v = t.v[k];
        const bool full = e0 + 3 < n && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
        if (full) {
            f32x4 pp = *(f32x4*)(p + e0), gg = *(const f32x4*)(g + e0), mm = *(f32x4*)(m + e0), vv = *(f32x4*)(v + e0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gr = gg[e] + wd * pp[e];
                mm[e] = mm[e] + (1.f - b1) * (gr - mm[e]);                 // lerp form, as torch does
                vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
                pp[e] -= lr_over_bc1 * (mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps));
            }
            *(f32x4*)(p + e0) = pp; *(f32x4*)(m + e0) = mm; *(f32x4*)(v + e0) = vv;
        } else {
            for (long e = e0; e < n && e < e0 + 4; ++e) {
                const float gr = g[e] + wd * p[e];
                const float mn = m[e] + (1.f - b1) * (gr - m[e]);
                const float vn = b2 * v[e] + (1.f - b2) * gr * gr;
                m[e] = mn; v[e] = vn;
                p[e] -= lr_over_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + eps));
            }
        }
    }
}

}  // namespace

extern "C" int fumi_hip_adam_step(fumi_ws_t* ws, fumi_stream_t stream, int n_tensors, float* const* params,
        const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const long* numel_host,
        float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    if (!ws || !params || !grads || !exp_avg || !exp_avg_sq || !numel_host || n_tensors < 1 || step < 1) return FUMI_EINVAL;
    if (n_tensors > MAXT) return FUMI_ENOTSUP;
    HIP_TRY(hipSetDevice(ws->device));
    hipStream_t st = (hipStream_t)stream;
    AdamTensors t;
    long tot4 = 0;
    for (int k = 0; k < n_tensors; ++k) {
        if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel_host[k] < 0) return FUMI_EINVAL;
        t.p[k] = params[k]; t.g[k] = grads[k]; t.m[k] = exp_avg[k]; t.v[k] = exp_avg_sq[k]; t.numel[k] = numel_host[k];
        tot4 += (numel_host[k] + 3) / 4;
        t.end[k] = tot4;
    }
    t.n = n_tensors;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    int blocks = (int)((tot4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) return FUMI_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, t, tot4, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)),
                       beta1, beta2, eps, weight_decay, ws->pub_src, ws->pub_n, ws->pub_dst, ws->pub_seq);
    ws->pub_dst = nullptr; ws->pub_src = nullptr;                 // a pending publication rode along
    LAUNCH_CHECK();
    return FUMI_OK;
}
