// Hypernetwork backward in ONE grid of independent workgroups, as a device function: its own launch (hyper.hip:
// hyper_bwd_fused_kernel) or RIDER workgroups at the front of the backward X-panel launch (xpanel.hip) -- the text path's
// backward only needs head_bar (complete after the reverse sweep) and is independent of the layer-0 gradient pass.
//
// What autograd does for hyper_net = Linear -> ReLU -> Linear [-> Tanh] inside outer_loss.backward() (fumi/models/fumi.py:76-85,192),
// rows = (episode, class) pairs.  Workgroup (rb, cb) owns the 16 rows of row block rb and the 64 hidden columns of chunk cb:
//   hp    = hbar (* (1 - h^2) with the tanh head)                               [16, H1]
//   ubar  = (hp A1[:, chunk]) * relu'(u[:, chunk]) * mscale                     [16, 64]      (never leaves LDS)
//   pA1[rb][:, chunk] = hp^T u[:, chunk]          pb1[rb] = colsum(hp)  (chunk 0)      pb0[rb][chunk] = colsum(ubar)
//   pA0[rb][chunk, :] = ubar^T c[rows]                                          [64, Dt]
// The per-row-block partial slabs are summed (and scaled) by the step's final reduction (launch_reduce_multi), like every
// other sum over episodes: no float atomics, fixed order.  The layer-0 weight gradient used to be a second, dependent
// launch over all R rows (hyper_bwd0_kernel); as row-block slabs it needs no second pass over ubar.
#pragma once
#include "common.h"

constexpr int HBW_HB = 16;

struct HyperBwdArgs {
    int R, Dt, Ht, H1, tanh_head;
    float mscale;                          // factor of the ReLU derivative (1, or 1/(1-p) with dropout after the ReLU)
    const float *c, *u, *h, *hbar, *A1;    // rows [R,Dt]; hidden activations [R,Ht]; output [R,H1]; its adjoint; layer 1 [H1,Ht]
    float *pA1, *pb1, *pb0, *pA0;          // slabs: [nrb,H1,Ht] [nrb,H1] [nrb,Ht] [nrb,Ht,Dt] (pA0 NULL: layer-0 weight gradient not formed)
    float* ub_out;                         // optional [R,Ht]: ubar to memory (callers that form ubar^T c themselves)
    // optional adjoint of the INPUT rows, for a network whose input is another network's output (AM3: h reads g's output):
    //   xpart [nrb][Ht/64][16][Dt] receives this workgroup's partial ubar_chunk A0[chunk, :]   (needs A0 and Dt <= HBW_XDT)
    //   hbar_parts / hbar_nparts: the adjoint this backward starts from is hbar + the sum of that many such partials
    //   (the consumer side: same row blocking, parts [nrb][hbar_nparts][16][H1])
    const float* A0; float* xpart;
    const float* hbar_parts; int hbar_nparts;
    int nrb, nblk;                         // row blocks; workgroups = 8 * (Ht/64) * ceil(nrb/8)
};
constexpr int HBW_XDT = 128;               // widest input whose adjoint partials are formed here (its layer-0 chunk [64, Dt] sits in LDS)
constexpr int HBW_DC = 384;                // input columns of the layer-0 weight slab formed per pass (the rows' LDS image is that wide)
__host__ __device__ inline int hyper_bwd_lds_floats(int Dt, int H1, int with_xpart = 0) {
    return HBW_HB * wg_ld(H1) + 2 * HBW_HB * wg_ld(64) + ((H1 + 3) & ~3) * wg_ld(64) + HBW_HB * wg_ld(Dt < HBW_DC ? Dt : HBW_DC) + 64 +
           (with_xpart ? 64 * wg_ld(Dt) : 0);
}

// bid in [0, nblk); sm >= hyper_bwd_lds_floats() floats; any workgroup size that is a multiple of 64
__device__ __forceinline__ void hyper_bwd_body(const HyperBwdArgs& a, int bid, float* sm) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int Dt = a.Dt, Ht = a.Ht, H1 = a.H1;
    const int nch = Ht >> 6;
    const int xcd = bid & 7, slot = bid >> 3;
    const int rb = xcd + 8 * (slot / nch), cb = slot % nch;
    if (rb >= a.nrb) return;
    const int m0 = rb * HBW_HB, nr = min(HBW_HB, a.R - m0);
    const int dcw = min(Dt, HBW_DC);                                      // (a multiple of 4: Dt is)
    const int ld1 = wg_ld(H1), ldc = wg_ld(64), ldd = wg_ld(dcw), H1r = (H1 + 3) & ~3;
    float* hp = sm; float* uc = hp + HBW_HB * ld1; float* ubc = uc + HBW_HB * ldc; float* A1c = ubc + HBW_HB * ldc;
    float* cr = A1c + H1r * ldc;
    float* A0c = cr + HBW_HB * ldd;                                       // [64][wg_ld(Dt)] rows cb*64.. of layer 0 (xpart only)
    const int ldx = wg_ld(Dt);
    const int tot = HBW_HB * ld1 + 2 * HBW_HB * ldc + H1r * ldc + HBW_HB * ldd + (a.xpart ? 64 * ldx : 0);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid * 4; i < tot; i += nt * 4) *(f32x4*)(sm + i) = z4;
    __syncthreads();
    // ---- staging (everything requested before the first LDS write of each group)
    for (int i = tid; i < nr * H1; i += nt) {
        const int m = i / H1, n = i - m * H1;
        float v = a.hbar[(long)(m0 + m) * H1 + n];
        if (a.hbar_parts) {                                                // + the producer's partials, in chunk order
            const float* pp = a.hbar_parts + ((long)rb * a.hbar_nparts * HBW_HB + m) * H1 + n;
            for (int c = 0; c < a.hbar_nparts; ++c) v += pp[(long)c * HBW_HB * H1];
        }
        if (a.tanh_head) { const float hv = a.h[(long)(m0 + m) * H1 + n]; v *= 1.f - hv * hv; }
        hp[m * ld1 + n] = v;
    }
    for (int i = tid; i < nr * 16; i += nt) {
        const int m = i >> 4, c4 = (i & 15) << 2;
        *(f32x4*)(uc + m * ldc + c4) = *(const f32x4*)(a.u + (long)(m0 + m) * Ht + cb * 64 + c4);
    }
    for (int i0 = tid; i0 < H1 * 16; i0 += 4 * nt) {
        f32x4 v[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int i = min(i0 + x * nt, H1 * 16 - 1);
            v[x] = *(const f32x4*)(a.A1 + (long)(i >> 4) * Ht + cb * 64 + ((i & 15) << 2));
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int i = i0 + x * nt;
            if (i < H1 * 16) *(f32x4*)(A1c + (i >> 4) * ldc + ((i & 15) << 2)) = v[x];
        }
    }
    // the block's rows, columns [dc0, dc0 + w): zero beyond w (the product walks whole 64-column blocks)
    auto stage_rows = [&](int dc0, int w) {
        const int d4 = ldd >> 2, n4 = HBW_HB * d4;
        for (int i0 = tid; i0 < n4; i0 += 4 * nt) {
            f32x4 v[4]; bool ok[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int i = min(i0 + x * nt, n4 - 1);
                const int m = i / d4, k4 = i - m * d4;
                ok[x] = m < nr && 4 * k4 < w;
                v[x] = *(const f32x4*)(a.c + (long)(m0 + min(m, nr - 1)) * Dt + dc0 + min(4 * k4, w - 4));
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int i = i0 + x * nt;
                if (i < n4) { const int m = i / d4, k4 = i - m * d4; *(f32x4*)(cr + m * ldd + 4 * k4) = ok[x] ? v[x] : z4; }
            }
        }
    };
    if (a.pA0) stage_rows(0, dcw);
    if (a.xpart) {
        const int d4 = Dt >> 2;
        for (int i = tid; i < 64 * d4; i += nt) {
            const int k = i / d4, c4 = (i - k * d4) << 2;
            *(f32x4*)(A0c + k * ldx + c4) = *(const f32x4*)(a.A0 + (long)(cb * 64 + k) * Dt + c4);
        }
    }
    __syncthreads();
    // ---- ubar chunk
    wg_lmm_wide<true>(nr, 64, H1, hp, ld1, A1c, ldc, [&](int m, int n, const f32x4& acc, int, auto) {
        const f32x4 uv = *(const f32x4*)(uc + m * ldc + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = uv[e] > 0.f ? acc[e] * a.mscale : 0.f;
        *(f32x4*)(ubc + m * ldc + n) = v;
        if (a.ub_out) *(f32x4*)(a.ub_out + (long)(m0 + m) * Ht + cb * 64 + n) = v;
    });
    // ---- layer-1 weight slab of this chunk, bias slabs
    float* pA = a.pA1 + (long)rb * H1 * Ht + cb * 64;
    wg_lmm_wide<false>(H1, 64, HBW_HB, hp, ld1, uc, ldc, [&](int m, int n, const f32x4& acc, int, auto) {
        *(f32x4*)(pA + (long)m * Ht + n) = acc;
    });
    if (cb == 0) wg_lcolsum(nr, H1, hp, ld1, [&](int n, float s_) { a.pb1[(long)rb * H1 + n] = s_; });
    wg_lds_barrier();
    wg_lcolsum(nr, 64, ubc, ldc, [&](int n, float s_) { a.pb0[(long)rb * Ht + cb * 64 + n] = s_; });
    // ---- layer-0 weight slab: rows of this chunk, every input column
    if (a.xpart) {                                                         // partial input adjoint: ubar_chunk [16,64] A0[chunk] [64,Dt]
        float* xp = a.xpart + ((long)rb * nch + cb) * HBW_HB * Dt;
        wg_lmm_wide<true>(HBW_HB, Dt, 64, ubc, ldc, A0c, ldx, [&](int m, int n, const f32x4& acc, int cnt, auto) {
            wg_st4(xp + (long)m * Dt + n, acc, cnt);
        });
    }
    if (!a.pA0) return;
    float* p0 = a.pA0 + ((long)rb * Ht + cb * 64) * Dt;
    for (int dc0 = 0; dc0 < Dt; dc0 += HBW_DC) {
        const int w = min(HBW_DC, Dt - dc0);
        if (dc0) { __syncthreads(); stage_rows(dc0, w); __syncthreads(); }      // (the previous pass has read its image)
        wg_lmm_wide<false>(64, w, HBW_HB, ubc, ldc, cr, ldd, [&](int m, int n, const f32x4& acc, int cnt, auto) {
            wg_st4(p0 + (long)m * Dt + dc0 + n, acc, cnt);
        });
    }
}

size_t hyper_bwd_fused_workspace_floats(int R, int Dt, int Ht, int H1);
// fills `a` (slabs carved from `part`, >= hyper_bwd_fused_workspace_floats floats) and appends the final sums to `segs`
// (which must have room for 4 more and carry the gradient scale); 0 when the shapes do not fit this form
// gA0 == NULL: no layer-0 weight slabs (ub_out is then the caller's input for that product)
int hyper_bwd_fused_args(int R, int Dt, int Ht, int H1, int tanh_head, float mscale, const float* c, const float* u, const float* h,
                         const float* hbar, const float* A1, float* part, float* gA0, float* gb0, float* gA1, float* gb1,
                         struct ReduceSegs* segs, HyperBwdArgs* a, float* ub_out = nullptr);
int launch_hyper_bwd_fused(hipStream_t st, const HyperBwdArgs& a);
