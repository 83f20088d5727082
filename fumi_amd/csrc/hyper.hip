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
#include <stdlib.h>
#include "hyper_fwd.h"
#include "hyper_bwd.h"

namespace {

constexpr int HB = HF_HB;                 // rows per block
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
    wg_stage_tab_to_lds(&s_stg, 1, (int)sizeof(StageTab) + 128);
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

// ---- forward, both layers in one launch (Ht <= 256) ---------------------------------------------------------------------
// One workgroup per 16-row block walks the whole network.  Layer 0's weights never pass through LDS: a lane of the
// (transposed) 16x16x4 MFMA needs W[n0 + (l & 15)][k .. k+3], four consecutive floats of ONE weight row, i.e. one 16-byte
// global load, so a wave that owns NT 16-column tiles of the hidden layer just keeps two FCH*16-deep chunks of those loads
// in flight (a CU fetching a 300 KB weight matrix alone is latency-bound: everything is requested up front).  The block's input rows (zero-padded to a chunk multiple) and the hidden activations -- layer 1's contraction
// operand -- are the only LDS images; layer 1's weight fragments are fetched while layer 0's epilogue runs.
constexpr int FCH = HF_FCH;            // 16-deep steps per chunk of weight-fragment loads (FwdDims, fwd_ldx: hyper_fwd.h)

template <int NT, int NCH>              // NT tiles per wave (Ht <= 128 NT), NCH chunks of FCH*16 contraction columns
__global__ __launch_bounds__(512) void hyper_fwd_fused_kernel(FwdDims d, const float* __restrict__ c, const float* __restrict__ A0,
                                                              const float* __restrict__ b0, const float* __restrict__ A1,
                                                              const float* __restrict__ b1, float* __restrict__ u,
                                                              float* __restrict__ h) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * HB, nr = min(HB, d.R - m0);
    const int Dt = d.Dt, Ht = d.Ht, H1 = d.H1, ldx = d.ldx, ldu = wg_ld(Ht);
    float* xs = sm; float* us = sm + HB * ldx;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    // layer 0 weight rows of this wave's tiles (tile t -> columns 16 t ..; waves take tiles wave, wave + 8, ...)
    const int ntile = Ht >> 4;
    const float* wrow[NT]; bool live[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tl = wave + 8 * t;
        live[t] = tl < ntile;
        wrow[t] = A0 + (long)min(tl * 16 + r, Ht - 1) * Dt;
    }
    struct Chunk { f32x4 w[NT][FCH]; };
    auto wload = [&](int ci) {
        Chunk ch;
#pragma unroll
        for (int s_ = 0; s_ < FCH; ++s_) {
            const int k = min((ci * FCH + s_) * 16 + 4 * q, Dt - 4);      // past Dt: any valid address (x is zero there)
#pragma unroll
            for (int t = 0; t < NT; ++t) ch.w[t][s_] = *(const f32x4*)(wrow[t] + k);
        }
        return ch;
    };
    Chunk ca = wload(0);
    Chunk cb = wload(NCH > 1 ? 1 : 0);

    // stage the block's input rows, zero-padded (rows past nr, columns past Dt)
    {
        const int l4 = ldx >> 2, tot4 = HB * l4;
        for (int i0 = tid; i0 < tot4; i0 += 4 * 512) {
            f32x4 v[4]; bool ok[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int i = min(i0 + x * 512, tot4 - 1);
                const int m = i / l4, k4 = i - m * l4;
                ok[x] = m < nr && 4 * k4 < Dt;
                v[x] = *(const f32x4*)(c + (long)(m0 + min(m, nr - 1)) * Dt + min(4 * k4, Dt - 4));
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) if (i0 + x * 512 < tot4) *(f32x4*)(xs + 4 * (i0 + x * 512)) = ok[x] ? v[x] : z4;
        }
    }
    f32x4 bias0[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias0[t] = *(const f32x4*)(b0 + min((wave + 8 * t) * 16 + 4 * q, Ht - 4));
    __syncthreads();

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = z4;
    const float* xr = xs + r * ldx + 4 * q;
    const int nstep = (Dt + 15) >> 4;
    auto mma = [&](const Chunk& ch, int ci) {
#pragma unroll
        for (int s_ = 0; s_ < FCH; ++s_) {
            if ((ci * FCH + s_) < nstep) {                // wave-uniform: the padded steps of the last chunk are skipped
                const f32x4 xf = *(const f32x4*)(xr + (ci * FCH + s_) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.w[t][s_][e], xf[e], acc[t], 0, 0, 0);
            }
        }
    };
    // layer 1 weight fragments (tile = wave), fetched as soon as a chunk's registers are free in the last round
    const int ntile1 = (H1 + 15) >> 4, n2 = Ht >> 4;
    constexpr int S1 = 8 * NT;                           // 16-deep steps of layer 1 (Ht <= 128 NT)
    f32x4 w1[S1];
    auto w1load = [&]() {
        const float* a1r = A1 + (long)min(wave * 16 + r, H1 - 1) * Ht + 4 * q;
#pragma unroll
        for (int s_ = 0; s_ < S1; ++s_) w1[s_] = *(const f32x4*)(a1r + min(s_, n2 - 1) * 16);
    };
    // both register sets are in flight from the start (everything, when Dt <= 2 * FCH * 16); the chunk walk is unrolled
    // at compile time so that layer 1's fragments can take over the set that is retired first
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
        if (ci & 1) mma(cb, ci); else mma(ca, ci);
        __builtin_amdgcn_sched_barrier(0);
        if (ci + 2 < NCH) { if (ci & 1) cb = wload(ci + 2); else ca = wload(ci + 2); }
        else if (ci + 2 == NCH || NCH == 1) w1load();
        __builtin_amdgcn_sched_barrier(0);
    }
    float bias1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias1[e] = b1[min(wave * 16 + 4 * q + e, H1 - 1)];

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (!live[t]) continue;
        const int n = (wave + 8 * t) * 16 + 4 * q;
        f32x4 v = acc[t] + bias0[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        *(f32x4*)(us + r * ldu + n) = v;
        if (r < nr) *(f32x4*)(u + (long)(m0 + r) * Ht + n) = v;
    }
    wg_lds_barrier();

    for (int t1 = wave; t1 < ntile1; t1 += 8) {          // (more than 8 tiles: later ones fetch their fragments here)
        f32x4 a = z4;
        const float* ur = us + r * ldu + 4 * q;
        if (t1 == wave) {
#pragma unroll
            for (int s_ = 0; s_ < S1; ++s_) {
                if (s_ < n2) {
                    const f32x4 uf = *(const f32x4*)(ur + s_ * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s_][e], uf[e], a, 0, 0, 0);
                }
            }
        } else {
            const float* a1r = A1 + (long)min(t1 * 16 + r, H1 - 1) * Ht + 4 * q;
            for (int s_ = 0; s_ < n2; ++s_) {
                const f32x4 wf = *(const f32x4*)(a1r + s_ * 16), uf = *(const f32x4*)(ur + s_ * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) a = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], uf[e], a, 0, 0, 0);
            }
        }
        const int n = t1 * 16 + 4 * q;
        if (r < nr && n < H1) {
            const int cnt = min(4, H1 - n);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bb = t1 == wave ? bias1[e] : b1[min(n + e, H1 - 1)];
                v[e] = a[e] + bb;
                if (d.tanh_head) v[e] = tanhf(v[e]);
            }
            wg_st4(h + (long)(m0 + r) * H1 + n, v, cnt);
        }
    }
}

// ---- forward, both layers in one launch, layer 0's columns split over workgroups: hyper_fwd.h (the same body also rides at
// the front of the forward X-panel launch, xpanel.hip) ------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void hyper_fwd_split_kernel(HyperFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_last;
    hyper_fwd_split_body<KS>(a, (int)blockIdx.x, sm, &s_last);
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
    wg_stage_tab_to_lds(&s_stg, 1, (int)sizeof(StageTab) + 128);
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
    wg_stage_tab_to_lds(&s_stg, 1, (int)sizeof(StageTab) + 128);
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

// ---- backward, one grid of independent (row block, column chunk) workgroups: hyper_bwd.h -------------------------------------
__global__ __launch_bounds__(512) void hyper_bwd_fused_kernel(HyperBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    hyper_bwd_body(a, (int)blockIdx.x, sm);
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

size_t hyper_fwd_workspace_floats(int R, int Ht, int H1) {
    return (size_t)((R + HB - 1) / HB) * (Ht / 64 + 1) * HB * H1 + 64;
}

int hyper_fwd_split_args(int R, int Dt, int Ht, int H1, int tanh_head, const float* c, const float* A0, const float* b0,
                         const float* A1, const float* b1, float* u, float* h, float* hpart, int* cnt, HyperFwdArgs* a) {
    if ((Dt & 3) || (Ht & 63) || R < 1 || H1 < 1) return 0;
    const bool al16_ = (((uintptr_t)c | (uintptr_t)A0 | (uintptr_t)A1 | (uintptr_t)b0 | (uintptr_t)u) & 15) == 0;
    const int nrb = (R + HB - 1) / HB, nch = Ht / 64;
    if (!(hpart && cnt && al16_ && Dt <= 768 && H1 <= 128 && nrb <= FUMI_HCNT)) return 0;
    a->d = FwdDims{R, Dt, Ht, H1, fwd_ldx(Dt), tanh_head, 0u, 0u, 1.f};
    a->c = c; a->A0 = A0; a->b0 = b0; a->A1 = A1; a->b1 = b1; a->u = u; a->h = h; a->hpart = hpart; a->cnt = cnt;
    a->nrb = nrb; a->nblk = 8 * nch * ((nrb + 7) / 8);
    return 1;
}

int launch_hyper_fwd_split(hipStream_t st, const HyperFwdArgs& a) {
    const size_t lds = hyper_fwd_split_lds_bytes(a.d.ldx);
    if (a.d.Dt <= 320) {
        FUMI_SET_DYN_LDS(hyper_fwd_split_kernel<20>, lds);
        hipLaunchKernelGGL(hyper_fwd_split_kernel<20>, dim3(a.nblk), dim3(256), lds, st, a);
    } else {
        FUMI_SET_DYN_LDS(hyper_fwd_split_kernel<48>, lds);
        hipLaunchKernelGGL(hyper_fwd_split_kernel<48>, dim3(a.nblk), dim3(256), lds, st, a);
    }
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_hyper_fwd(hipStream_t st, int R, int Dt, int Ht, int H1, int tanh_head, const float* c, const float* A0,
                     const float* b0, const float* A1, const float* b1, float* u, float* h, float* hpart, int* cnt) {
    if (!hyper_lds_fits(R, Dt, Ht, H1)) return FUMI_ENOTSUP;
    // FUMI_HYPER_FWD: 2 (default) one launch, layer-0 columns split over workgroups; 1 one launch, a workgroup per row block;
    // 0 one launch per layer
    static const int hsplit = getenv("FUMI_HYPER_FWD") ? atoi(getenv("FUMI_HYPER_FWD")) : 2;
    {
        HyperFwdArgs a;
        if (hsplit == 2 && hyper_fwd_split_args(R, Dt, Ht, H1, tanh_head, c, A0, b0, A1, b1, u, h, hpart, cnt, &a))
            return launch_hyper_fwd_split(st, a);
    }
    const int no_fuse = hsplit == 0;
    const bool al = (((uintptr_t)c | (uintptr_t)A0 | (uintptr_t)A1 | (uintptr_t)b0 | (uintptr_t)u) & 15) == 0;
    const int nch = (Dt + FCH * 16 - 1) / (FCH * 16);
    if (!no_fuse && Ht <= 256 && nch <= 4 && al) {
        FwdDims d{R, Dt, Ht, H1, fwd_ldx(Dt), tanh_head};
        const size_t lds = (size_t)HB * (d.ldx + wg_ld(Ht)) * sizeof(float);
        const int nrb = (R + HB - 1) / HB;
#define FWD_LAUNCH(NT_, NCH_)                                                                                    \
        do {                                                                                                     \
            FUMI_SET_DYN_LDS((hyper_fwd_fused_kernel<NT_, NCH_>), lds);                                            \
            hipLaunchKernelGGL((hyper_fwd_fused_kernel<NT_, NCH_>), dim3(nrb), dim3(512), lds, st, d, c, A0, b0, A1, b1, u, h); \
        } while (0)
        if (Ht <= 128) {
            if (nch == 1) FWD_LAUNCH(1, 1); else if (nch == 2) FWD_LAUNCH(1, 2); else if (nch == 3) FWD_LAUNCH(1, 3); else FWD_LAUNCH(1, 4);
        } else {
            if (nch == 1) FWD_LAUNCH(2, 1); else if (nch == 2) FWD_LAUNCH(2, 2); else if (nch == 3) FWD_LAUNCH(2, 3); else FWD_LAUNCH(2, 4);
        }
#undef FWD_LAUNCH
        LAUNCH_CHECK();
        return FUMI_OK;
    }
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

// ---- fused backward (hyper_bwd.h) ------------------------------------------------------------------------------------------
size_t hyper_bwd_fused_workspace_floats(int R, int Dt, int Ht, int H1) {
    const size_t nrb = (R + HB - 1) / HB;
    return nrb * ((size_t)H1 * Ht + H1 + Ht + (size_t)Ht * Dt) + 256;        // (Dt = 0: without the layer-0 weight slabs)
}

int hyper_bwd_fused_args(int R, int Dt, int Ht, int H1, int tanh_head, float mscale, const float* c, const float* u, const float* h,
                         const float* hbar, const float* A1, float* part, float* gA0, float* gb0, float* gA1, float* gb1,
                         ReduceSegs* segs, HyperBwdArgs* a, float* ub_out) {
    static const int on = getenv("FUMI_HYPER_BWD") ? atoi(getenv("FUMI_HYPER_BWD")) : 1;     // 0: the two-launch form (bwd1 + bwd0)
    if (!on || R < 1 || (Dt & 3) || (Ht & 63) || H1 < 1 || !part || !segs || segs->n + 4 > 24) return 0;
    if ((((uintptr_t)c | (uintptr_t)u | (uintptr_t)A1 | (uintptr_t)part | (uintptr_t)ub_out) & 15) != 0) return 0;
    if ((size_t)hyper_bwd_lds_floats(Dt, H1) * 4 > 150 * 1024) return 0;
    const int nrb = (R + HB - 1) / HB, nch = Ht / 64;
    a->R = R; a->Dt = Dt; a->Ht = Ht; a->H1 = H1; a->tanh_head = tanh_head; a->mscale = mscale;
    a->c = c; a->u = u; a->h = h; a->hbar = hbar; a->A1 = A1;
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    a->pA1 = part; a->pb1 = a->pA1 + up4((size_t)nrb * H1 * Ht); a->pb0 = a->pb1 + up4((size_t)nrb * H1);
    a->pA0 = gA0 ? a->pb0 + up4((size_t)nrb * Ht) : nullptr;
    a->ub_out = ub_out;
    a->A0 = nullptr; a->xpart = nullptr; a->hbar_parts = nullptr; a->hbar_nparts = 0;
    a->nrb = nrb; a->nblk = 8 * nch * ((nrb + 7) / 8);
    segs->add(a->pA1, nrb, (long)H1 * Ht, (long)H1 * Ht, gA1);
    segs->add(a->pb1, nrb, H1, H1, gb1);
    segs->add(a->pb0, nrb, Ht, Ht, gb0);
    if (gA0) segs->add(a->pA0, nrb, (long)Ht * Dt, (long)Ht * Dt, gA0);
    return 1;
}

int launch_hyper_bwd_fused(hipStream_t st, const HyperBwdArgs& a) {
    const size_t lds = (size_t)hyper_bwd_lds_floats(a.Dt, a.H1, a.xpart != nullptr) * 4;
    if (lds > 160 * 1024 || (a.xpart && (a.Dt > HBW_XDT || !a.A0))) return FUMI_EINVAL;
    FUMI_SET_DYN_LDS(hyper_bwd_fused_kernel, lds);
    hipLaunchKernelGGL(hyper_bwd_fused_kernel, dim3(a.nblk), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return FUMI_OK;
}
