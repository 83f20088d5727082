// Hypernetwork forward / backward (fumi/models/fumi.py:76-85 hyper_net = Linear -> ReLU -> Linear [-> Tanh], rows =
// (episode, class) pairs; backward = what autograd does for it inside outer_loss.backward(), fumi.py:192).
//
// The problem is tiny (R = B*N = 160 rows, 300|768 -> 256 -> H+1): as plain GEMM launches every product is a dozen
// workgroups walking ~10 dependent contraction slabs.  Here each product is a grid of LDS-resident workgroups (common.h
// "LDS-resident products"): the operands of a block are staged in ONE batch of loads and multiplied from LDS.
//   lin:   y  = act(x W^T + b)               grid (N/NC, R/16)    block: 16 rows x NC columns, whole K; both forward layers
//   bwd1:  hp = hbar [* tanh'], ubar = (hp A1) * relu'(u), partial slabs of hp^T u, colsum(hp), colsum(ubar)   grid (R/16)
//   bwd0:  gA0 = scale * ubar^T c            grid (Dt/64, Ht/64)  block: 64 x 64 outputs, K = R
// The per-row-block partial slabs of bwd1 are summed by launch_reduce_multi.
#include "common.h"

namespace {

constexpr int HB = 16;                 // rows per block
constexpr int HCAP = 38000;            // floats of dynamic LDS

__host__ __device__ inline int h_r4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int h_r16(int x) { return (x + 15) & ~15; }

struct HyperDims { int R, Dt, Ht, H1, NC, tanh_head; float scale; };
struct LinDims { int R, K, Nn, NC, act; };          // act: 1 relu, 2 tanh, 0 none

// ---- forward layer: y[rb, nc] = act(x[rb,:] W[nc,:]^T + b[nc]) -- both hypernetwork layers use it ---------------------------
// 16 rows x NC output columns per workgroup, whole contraction staged at once.
__global__ __launch_bounds__(512) void hyper_lin_kernel(StageTab stg, LinDims d, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ StageTab s_stg;
    const int cb = blockIdx.x, rb = blockIdx.y;
    const int nr = min(HB, d.R - rb * HB), ncols = min(d.NC, d.Nn - cb * d.NC);
    const int ldk = wg_ld(d.K);
    float* xi = sm; float* Wi = sm + HB * ldk; float* bi = Wi + h_r16(d.NC) * ldk;
    // no zero-fill: K is a multiple of 4 (hyper_lds_fits), so the product never reads K padding, and rows / columns past
    // nr / ncols only feed outputs that are not stored
    wg_stage_tab_to_lds(&s_stg);
    __syncthreads();
    wg_stage_rows<16>(&s_stg, 0, rb, cb, nr, sm, ncols, ncols);
    wg_lds_barrier();
    float* yr = y + (long)rb * HB * d.Nn + cb * d.NC;
    wg_lmm<true, true>(nr, ncols, d.K, xi, ldk, Wi, ldk, [&](int m, int n, const f32x4& acc, int cnt) {
        f32x4 v = acc + *(const f32x4*)(bi + n);
        if (d.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        } else if (d.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
        }
        wg_st4(yr + (long)m * d.Nn + n, v, cnt);
    });
}

// ---- bwd1 -------------------------------------------------------------------------------------------------------------
// per row block: hp = hbar (* (1 - h^2)); ubar = (hp A1) * relu'(u) -> global; partial slabs pA1[rb] = hp^T u,
// pb1[rb] = colsum(hp), pb0[rb] = colsum(ubar)
__global__ __launch_bounds__(512) void hyper_bwd1_kernel(StageTab stg, HyperDims d, const float* __restrict__ h,
                                                         float* __restrict__ ub, float* __restrict__ pA1,
                                                         float* __restrict__ pb1, float* __restrict__ pb0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ StageTab s_stg;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int rb = blockIdx.x;
    const int nr = min(HB, d.R - rb * HB);
    const int ldt = wg_ld(d.Ht), ld1 = wg_ld(d.H1);
    float* hp = sm; float* ui = hp + HB * ld1; float* ubi = ui + HB * ldt; float* A1 = ubi + HB * ldt;
    const int tot = HB * ld1 + 2 * HB * ldt + h_r4(d.H1) * ldt;
    wg_stage_tab_to_lds(&s_stg);
    for (int i = tid * 4; i < tot; i += nt * 4) *(f32x4*)(sm + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    wg_stage_rows<12>(&s_stg, 0, rb, 0, nr, sm);          // hbar rows -> hp, u rows, A1
    wg_lds_barrier();
    if (d.tanh_head) {
        const float* hr = h + (long)rb * HB * d.H1;
        for (int i = tid; i < nr * d.H1; i += nt) { const int m = i / d.H1, n = i - m * d.H1; const float hv = hr[i]; hp[m * ld1 + n] *= 1.f - hv * hv; }
        __syncthreads();
    }
    float* ubr = ub + (long)rb * HB * d.Ht;
    wg_lmm_wide<true>(nr, d.Ht, d.H1, hp, ld1, A1, ldt, [&](int m, int n, const f32x4& acc, int cnt, auto) {
        const f32x4 uv = *(const f32x4*)(ui + m * ldt + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = uv[e] > 0.f ? acc[e] : 0.f;
        *(f32x4*)(ubi + m * ldt + n) = v;
        wg_st4(ubr + (long)m * d.Ht + n, v, cnt);
    });
    float* pA = pA1 + (long)rb * d.H1 * d.Ht;
    wg_lmm_wide<false>(d.H1, d.Ht, nr, hp, ld1, ui, ldt, [&](int m, int n, const f32x4& acc, int cnt, auto) {
        wg_st4(pA + (long)m * d.Ht + n, acc, cnt);
    });
    wg_lcolsum(nr, d.H1, hp, ld1, [&](int n, float s_) { pb1[(long)rb * d.H1 + n] = s_; });
    wg_lds_barrier();
    wg_lcolsum(nr, d.Ht, ubi, ldt, [&](int n, float s_) { pb0[(long)rb * d.Ht + n] = s_; });
}

// ---- bwd0: gA0[mc, nc] = scale * ubar[:, mc]^T c[:, nc] ---------------------------------------------------------------
__global__ __launch_bounds__(512) void hyper_bwd0_kernel(StageTab stg, HyperDims d, float* __restrict__ gA0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ StageTab s_stg;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nb = blockIdx.x, mb = blockIdx.y;
    const int mc = min(64, d.Ht - mb * 64), nc = min(64, d.Dt - nb * 64);
    const int ld = wg_ld(64), RS = h_r4(d.R);
    float* ubi = sm; float* ci = sm + RS * ld;
    const int tot = 2 * RS * ld;
    wg_stage_tab_to_lds(&s_stg);
    for (int i = tid * 4; i < tot; i += nt * 4) *(f32x4*)(sm + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // job 0 (ubar columns): part = mb, width mc; job 1 (c columns): tile = nb, width nc -- two run-time widths, so the
    // second one is passed in the "rows" slot trick: both jobs copy d.R rows; widths go through nc / the plan's own cols
    wg_stage_rows<12>(&s_stg, 0, nb, mb, d.R, sm, 0, nc);
    wg_lds_barrier();
    float* out = gA0 + (long)mb * 64 * d.Dt + nb * 64;
    wg_lmm_wide<false>(mc, nc, d.R, ubi, ld, ci, ld, [&](int m, int n, const f32x4& acc, int cnt, auto) {
        wg_st4(out + (long)m * d.Dt + n, d.scale * acc, cnt);
    });
}

}  // namespace

// column chunk of a forward layer with contraction depth K: largest of 64/32/16 whose block stays under LIN_CAP floats
constexpr int LIN_CAP = HCAP;           // (a 76 KB cap would let a block share a CU with the X-panel kernel on the other
                                        //  stream; measured: that slows the matrix pass by 40 % -- keep the blocks large)
static int lin_chunk(int K) {
    int NC = 64;
    while (NC >= 16 && (HB + NC) * wg_ld(K) + NC > LIN_CAP) NC >>= 1;
    return NC >= 16 ? NC : 0;
}

// all-or-nothing: 0 when the shapes do not fit the LDS-resident kernels (the caller then uses plain GEMMs)
int hyper_lds_fits(int R, int Dt, int Ht, int H1) {
    if ((Dt & 3) || (Ht & 63) || R < 1) return 0;           // column chunks are copied as float4 / 64-wide blocks
    if (!lin_chunk(Dt) || !lin_chunk(Ht)) return 0;
    if (HB * wg_ld(H1) + 2 * HB * wg_ld(Ht) + h_r4(H1) * wg_ld(Ht) > HCAP) return 0;
    if (2 * h_r4(R) * wg_ld(64) > HCAP) return 0;
    return 1;
}

static int launch_lin(hipStream_t st, int R, int K, int Nn, int act, const float* x, const float* W, const float* b, float* y) {
    const int NC = lin_chunk(K);
    if (!NC) return FUMI_ENOTSUP;
    LinDims d{R, K, Nn, NC, act};
    const int ldk = wg_ld(K), nrb = (R + HB - 1) / HB;
    StageTab tb; tb.init();
    tb.add(x, 0, (long)HB * K, 0, K, -1, HB, K, 0, ldk);                                 // rows of this block
    tb.add(W, 0, 0, (long)NC * K, K, -2, NC, K, HB * ldk, ldk);                          // weight rows of this column chunk
    tb.add(b, 0, 0, NC, Nn, 1, 1, -NC, (HB + h_r16(NC)) * ldk, NC);
    // a narrow last chunk is copied lane by lane when its width is no multiple of 4: keep the float4 path honest
    if ((Nn % NC) & 3) tb.vec[2] = 0, tb.lg[2] = 6;
    if (tb.bad || tb.nunits > 64 * 8) return FUMI_ENOTSUP;
    const int tot = (HB + h_r16(NC)) * ldk + h_r4(NC);
    FUMI_SET_DYN_LDS(hyper_lin_kernel, tot * 4);
    hipLaunchKernelGGL(hyper_lin_kernel, dim3((Nn + NC - 1) / NC, nrb), dim3(512), tot * 4, st, tb, d, y);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_hyper_fwd(hipStream_t st, int R, int Dt, int Ht, int H1, int tanh_head, const float* c, const float* A0,
                     const float* b0, const float* A1, const float* b1, float* u, float* h) {
    if (!hyper_lds_fits(R, Dt, Ht, H1)) return FUMI_ENOTSUP;
    int rc = launch_lin(st, R, Dt, Ht, 1, c, A0, b0, u);
    if (rc) return rc;
    return launch_lin(st, R, Ht, H1, tanh_head ? 2 : 0, u, A1, b1, h);
}

size_t hyper_bwd_workspace_floats(int R, int Ht, int H1) {
    const size_t nrb = (R + HB - 1) / HB;
    return nrb * ((size_t)H1 * Ht + H1 + Ht) + 192;
}

// g_phi = {gA0 [Ht,Dt], gb0 [Ht], gA1 [H1,Ht], gb1 [H1]}, all scaled by `scale`; ub [R,Ht] and `part` are workspace
int launch_hyper_bwd(hipStream_t st, int R, int Dt, int Ht, int H1, int tanh_head, float scale, const float* c,
                     const float* u, const float* h, const float* hbar, const float* A1, float* ub, float* part,
                     float* gA0, float* gb0, float* gA1, float* gb1, ReduceSegs* defer) {
    if (!hyper_lds_fits(R, Dt, Ht, H1)) return FUMI_ENOTSUP;
    HyperDims d{R, Dt, Ht, H1, 64, tanh_head, scale};
    const int nrb = (R + HB - 1) / HB;
    float* pA1 = part; float* pb1 = pA1 + (size_t)nrb * H1 * Ht; float* pb0 = pb1 + (size_t)nrb * H1;
    {
        const int ldt = wg_ld(Ht), ld1 = wg_ld(H1);
        StageTab tb; tb.init();
        tb.add(hbar, 0, (long)HB * H1, 0, H1, -1, HB, H1, 0, ld1);
        tb.add(u, 0, (long)HB * Ht, 0, Ht, -1, HB, Ht, HB * ld1, ldt);
        tb.add(A1, 0, 0, 0, Ht, H1, H1, Ht, HB * ld1 + 2 * HB * ldt, ldt);
        if (tb.bad || tb.nunits > 64 * 8) return FUMI_ENOTSUP;
        const int tot = HB * ld1 + 2 * HB * ldt + h_r4(H1) * ldt;
        FUMI_SET_DYN_LDS(hyper_bwd1_kernel, tot * 4);
        hipLaunchKernelGGL(hyper_bwd1_kernel, dim3(nrb), dim3(512), tot * 4, st, tb, d, h, ub, pA1, pb1, pb0);
        LAUNCH_CHECK();
    }
    {
        const int ld = wg_ld(64), RS = h_r4(R);
        StageTab tb; tb.init();
        // Ht is a multiple of 64 (hyper_lds_fits); the last Dt chunk may be narrower: its width is the plan's run-time nc
        tb.add(ub, 0, 0, 64, Ht, R, R, 64, 0, ld);
        tb.add(c, 0, 64, 0, Dt, R, R, -64, RS * ld, ld);
        if (tb.bad || tb.nunits > 64 * 8) return FUMI_ENOTSUP;
        const int tot = 2 * RS * ld;
        FUMI_SET_DYN_LDS(hyper_bwd0_kernel, tot * 4);
        hipLaunchKernelGGL(hyper_bwd0_kernel, dim3((Dt + 63) / 64, Ht / 64), dim3(512), tot * 4, st, tb, d, gA0);
        LAUNCH_CHECK();
    }
    ReduceSegs own; own.n = 0; own.scale = scale;
    ReduceSegs& sg = (defer && defer->n + 3 <= 24 && defer->scale == scale) ? *defer : own;
    sg.add(pA1, nrb, (long)H1 * Ht, (long)H1 * Ht, gA1);
    sg.add(pb1, nrb, H1, H1, gb1);
    sg.add(pb0, nrb, Ht, Ht, gb0);
    return &sg == &own ? launch_reduce_multi(st, own) : FUMI_OK;
}
