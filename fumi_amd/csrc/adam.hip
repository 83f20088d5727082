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
                float pe = pp[e], me = mm[e], ve = vv[e];
                adam_update1(gg[e], pe, me, ve, lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd);
                pp[e] = pe; mm[e] = me; vv[e] = ve;
            }
            *(f32x4*)(p + e0) = pp; *(f32x4*)(m + e0) = mm; *(f32x4*)(v + e0) = vv;
        } else {
            for (long e = e0; e < n && e < e0 + 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                adam_update1(g[e], pe, me, ve, lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
        }
    }
}

}  // namespace

static void adam_launch(fumi_ws* ws, hipStream_t st, const AdamTensors& t, long tot4, float lr_over_bc1, float inv_sqrt_bc2, float b1,
                        float b2, float eps, float wd) {
    int blocks = (int)((tot4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) return;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, t, tot4, lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd, ws->pub_src,
                       ws->pub_n, ws->pub_dst, ws->pub_seq);
    ws->pub_dst = nullptr; ws->pub_src = nullptr;                 // a pending publication rode along
}

int launch_adam_pending(fumi_ws* ws, hipStream_t st) {
    AdamPending* ap = ws ? ws->adam : nullptr;
    if (!ap || !ap->on) return FUMI_OK;
    AdamTensors t;
    long tot4 = 0;
    for (int k = 0; k < ap->n; ++k) {
        t.p[k] = ap->p[k]; t.g[k] = ap->g[k]; t.m[k] = ap->m[k]; t.v[k] = ap->v[k]; t.numel[k] = ap->numel[k];
        tot4 += (ap->numel[k] + 3) / 4;
        t.end[k] = tot4;
    }
    t.n = ap->n;
    adam_launch(ws, st, t, tot4, ap->lr_over_bc1, ap->inv_sqrt_bc2, ap->b1, ap->b2, ap->eps, ap->wd);
    ap->on = 0;
    LAUNCH_CHECK();
    return FUMI_OK;
}

// Deferred form of fumi_hip_adam_step: nothing is launched; the update is folded into the LAST launch of the next training
// meta-step of this workspace (fumi_hip_fumi_step / _indexed: the final reduction produces every gradient element, Adam follows
// element by element in the same thread) -- single GPU only: with several ranks the all-reduce lies between gradient and update.
// fumi_hip_adam_flush launches whatever is still pending as the ordinary Adam kernel (a step that could not fold it, or none).
extern "C" int fumi_hip_adam_step_deferred(fumi_ws_t* ws, int n_tensors, float* const* params, const float* const* grads,
        float* const* exp_avg, float* const* exp_avg_sq, const long* numel_host,
        float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    if (!ws || !params || !grads || !exp_avg || !exp_avg_sq || !numel_host || n_tensors < 1 || step < 1) return FUMI_EINVAL;
    if (n_tensors > 24) return FUMI_ENOTSUP;
    if (!ws->adam) { ws->adam = new AdamPending(); ws->adam->on = 0; }
    AdamPending* ap = ws->adam;
    for (int k = 0; k < n_tensors; ++k) {
        if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel_host[k] < 0) return FUMI_EINVAL;
        ap->p[k] = params[k]; ap->g[k] = grads[k]; ap->m[k] = exp_avg[k]; ap->v[k] = exp_avg_sq[k]; ap->numel[k] = numel_host[k];
    }
    ap->n = n_tensors;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    ap->lr_over_bc1 = (float)(lr / bc1); ap->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    ap->b1 = beta1; ap->b2 = beta2; ap->eps = eps; ap->wd = weight_decay;
    ap->on = 1;
    return FUMI_OK;
}

// *launched = 1 when the pending step was still to do (it is launched now, as the plain kernel), 0 when a meta-step had folded it
extern "C" int fumi_hip_adam_flush(fumi_ws_t* ws, fumi_stream_t stream, int* launched) {
    if (!ws) return FUMI_EINVAL;
    const int pending = ws->adam && ws->adam->on;
    if (launched) *launched = pending;
    if (!pending) return FUMI_OK;
    HIP_TRY(hipSetDevice(ws->device));
    return launch_adam_pending(ws, (hipStream_t)stream);
}

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
    if (tot4 < 1) return FUMI_OK;
    adam_launch(ws, st, t, tot4, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay);
    LAUNCH_CHECK();
    return FUMI_OK;
}
