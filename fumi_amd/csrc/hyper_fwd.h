// Split hypernetwork forward as a device function, so that it can run either as its own launch (hyper.hip:
// hyper_fwd_split_kernel) or as RIDER workgroups at the front of the forward X-panel launch (xpanel.hip): the text path
// (hypernetwork) and the image path (X-panel product) are independent until the inner loop, the rider's few dozen workgroups
// start first and fit into the slots the X-panel grid leaves free (480 of 512 at the bench shapes), and the step loses one
// dependent launch (12.8 us + the launch gap) without any cross-stream event.
//
// Hypernetwork: fumi/models/fumi.py:76-85 (Linear -> ReLU -> Linear [-> Tanh]), rows = (episode, class) pairs.
#pragma once
#include "common.h"

constexpr int HF_HB = 16;                  // rows per block
constexpr int HF_FCH = 12;                 // 16-deep steps per chunk of weight-fragment loads (LDS row padding unit)
struct FwdDims {
    int R, Dt, Ht, H1, ldx;
    int tanh_head;                         // output activation: 0 none, 1 tanh (FuMI's --tanh_head), 2 sigmoid (AM3's lamda network)
    unsigned drop_thr, drop_key;           // dropout after the ReLU (AM3 g / h, am3.py:80-88): keep iff mix(key ^ element index) >= thr
    float drop_scale;
};
__host__ __device__ inline int fwd_ldx(int Dt) { return (Dt + HF_FCH * 16 - 1) / (HF_FCH * 16) * (HF_FCH * 16) + 4; }

struct HyperFwdArgs {
    FwdDims d;
    const float *c, *A0, *b0, *A1, *b1;    // rows [R,Dt]; layer 0 [Ht,Dt] [Ht]; layer 1 [H1,Ht] [H1]
    float *u, *h, *hpart;                  // hidden activations [R,Ht]; output [R,H1]; partial layer-1 products
    int* cnt;                              // arrival counters, one per 16-row block (zero between launches)
    int nrb;                               // row blocks
    int nblk;                              // workgroups of the grid this forward occupies: 8 * (Ht/64) * ceil(nrb/8); 0 = none
};
constexpr int HF_RIDER_KS = 24;            // the rider form keeps 24 steps (384 columns) of weight fragments in registers at a time (register
                                           // budget beside the X-panel code) and walks wider inputs in several chunks (HF_RIDER_MAXDT)
constexpr int HF_RIDER_MAXDT = 768;
__host__ __device__ inline size_t hyper_fwd_split_lds_bytes(int ldx) { return (size_t)HF_HB * (ldx + 64 + 4) * sizeof(float); }

// A row block is handled by Ht/64 workgroups of 4 waves: each takes 64 hidden columns (one 16-column tile per wave, the
// wave's weight fragments requested up front), writes its slice of the hidden activations and its PARTIAL product with the
// matching 64 columns of layer 1's weights; the workgroup that arrives last at the row block's counter adds the partials in
// chunk order (deterministic), the bias and the optional tanh.  Partials travel with agent-scope stores / loads; nobody
// waits for anybody.  Ids are XCD-grouped: the chunks of a row block share an XCD (ids equal mod 8).
// bid = workgroup id within this forward's own id space [0, nblk); sm = >= hyper_fwd_split_lds_bytes() of LDS; 256 threads.
template <int KS, bool CHUNKED = false>    // KS 16-deep contraction steps of weight fragments in registers: Dt <= 16 KS, or CHUNKED: any Dt,
                                           // 16 KS columns at a time (each chunk's fragments are requested when the previous chunk is done)
__device__ __forceinline__ void hyper_fwd_split_body(const HyperFwdArgs& a, int bid, float* sm, int* s_last) {
    const FwdDims& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int Dt = d.Dt, Ht = d.Ht, H1 = d.H1, ldx = d.ldx;
    const int nch = Ht >> 6;
    const int xcd = bid & 7, slot = bid >> 3;
    const int rb = xcd + 8 * (slot / nch), cb = slot % nch;
    if (rb >= a.nrb) return;
    const int m0 = rb * HF_HB, nr = min(HF_HB, d.R - m0);
    constexpr int ldu = 64 + 4;
    float* xs = sm; float* us = sm + HF_HB * ldx;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    const int n0 = cb * 64 + wave * 16;
    const float* wrow = a.A0 + (long)(n0 + r) * Dt;
    f32x4 wf[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) wf[s_] = *(const f32x4*)(wrow + min(s_ * 16 + 4 * q, Dt - 4));
    {
        const int l4 = ldx >> 2, tot4 = HF_HB * l4;
        for (int i0 = tid; i0 < tot4; i0 += 4 * 256) {
            f32x4 v[4]; bool ok[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int i = min(i0 + x * 256, tot4 - 1);
                const int m = i / l4, k4 = i - m * l4;
                ok[x] = m < nr && 4 * k4 < Dt;
                v[x] = *(const f32x4*)(a.c + (long)(m0 + min(m, nr - 1)) * Dt + min(4 * k4, Dt - 4));
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) if (i0 + x * 256 < tot4) *(f32x4*)(xs + 4 * (i0 + x * 256)) = ok[x] ? v[x] : z4;
        }
    }
    const f32x4 bias0 = *(const f32x4*)(a.b0 + n0 + 4 * q);
    // layer 1 fragments of this chunk: tiles wave and wave + 4 (H1 <= 128), contraction over the chunk's 64 columns
    const int ntile1 = (H1 + 15) >> 4;
    f32x4 w1[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* a1r = a.A1 + (long)min((wave + 4 * t) * 16 + r, H1 - 1) * Ht + cb * 64 + 4 * q;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) w1[t][s_] = *(const f32x4*)(a1r + s_ * 16);
    }
    __syncthreads();

    f32x4 acc = z4;
    const float* xr = xs + r * ldx + 4 * q;
    const int nstep = (Dt + 15) >> 4;
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
        if (s_ < nstep) {
            const f32x4 xf = *(const f32x4*)(xr + s_ * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s_][e], xf[e], acc, 0, 0, 0);
        }
    }
    if constexpr (CHUNKED) {
        for (int s0 = KS; s0 < nstep; s0 += KS) {
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) wf[s_] = *(const f32x4*)(wrow + min((s0 + s_) * 16 + 4 * q, Dt - 4));
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                if (s0 + s_ < nstep) {
                    const f32x4 xf = *(const f32x4*)(xr + (s0 + s_) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s_][e], xf[e], acc, 0, 0, 0);
                }
            }
        }
    }
    {
        f32x4 v = acc + bias0;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        if (d.drop_thr) {                  // the same counter-based mask as gemm.hip's epilogue: element index = row * Ht + column
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned x = d.drop_key ^ (unsigned)((long)(m0 + r) * Ht + n0 + 4 * q + e);
                x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
                v[e] = x >= d.drop_thr ? v[e] * d.drop_scale : 0.f;
            }
        }
        *(f32x4*)(us + r * ldu + wave * 16 + 4 * q) = v;
        if (r < nr) *(f32x4*)(a.u + (long)(m0 + r) * Ht + n0 + 4 * q) = v;
    }
    wg_lds_barrier();
    float* mine = a.hpart + ((long)rb * nch + cb) * HF_HB * H1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int t1 = wave + 4 * t;
        if (t1 < ntile1) {
            f32x4 p = z4;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const f32x4 uf = *(const f32x4*)(us + r * ldu + s_ * 16 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) p = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t][s_][e], uf[e], p, 0, 0, 0);
            }
            const int n = t1 * 16 + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < H1) __hip_atomic_store(mine + r * H1 + n + e, p[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    wg_drain_stores();                                    // every wave: its sc1 partial stores have completed ...
    __syncthreads();                                      // ... before lane 0 signals for all of them
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(a.cnt + rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = old == nch - 1;
        if (old == nch - 1) __hip_atomic_store(a.cnt + rb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!*s_last) return;
    const float* all = a.hpart + (long)rb * nch * HF_HB * H1;
    for (int i = tid; i < nr * H1; i += 256) {
        const int m = i / H1, n = i - m * H1;
        float v = a.b1[n];
        for (int cc = 0; cc < nch; ++cc) v += __hip_atomic_load(all + (long)cc * HF_HB * H1 + m * H1 + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.h[(long)(m0 + m) * H1 + n] = d.tanh_head == 1 ? tanhf(v) : d.tanh_head == 2 ? 1.f / (1.f + expf(-v)) : v;
    }
}

// fills `a` when the split forward applies to these shapes / pointers (hyper.hip); 0 otherwise
int hyper_fwd_split_args(int R, int Dt, int Ht, int H1, int tanh_head, const float* c, const float* A0, const float* b0,
                         const float* A1, const float* b1, float* u, float* h, float* hpart, int* cnt, HyperFwdArgs* a);
int launch_hyper_fwd_split(hipStream_t st, const HyperFwdArgs& a);     // the same forward as its own launch
