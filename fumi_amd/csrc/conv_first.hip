// Block 1 of the Conv4 encoder without its 84 x 84 x 64 maps.  The first convolution has K = 9 Cin = 27: recomputing it from
// the 3-channel image costs 1.3 ms of matrix time for 5 120 images, storing its pre-activation costs 1.9 MB per image and
// pass (write it, read it for the pool, read it twice and write its gradient in the backward, read that for the weight
// gradient: 16 of the 66 ms of a 32-episode meta-step).  So every pass over block 1 re-derives u = conv(image, W1) for one
// BAND of two image rows (= one pooled row) at a time, keeps it in LDS, and only the pooled activations x1 / their gradients
// (1/4 of the size) and the images (1/21) touch HBM:
//   c1_pool    u -> BN -> ReLU -> 2x2 max-pool -> x1 row            (tangent: also x1' from u' = conv(image, W1'))
//   c1_reduce  u, dx1 row -> arg-max -> sum dv, sum dv xh            (tangent: sum dv', sum dv' xh, sum dv xh')
//   c1_wgrad   u, dx1 row -> du in LDS (in place) -> dW1 += du (x) image   (tangent: du')
// (the batch statistics themselves come from conv1_kernel with its store switched off.)  Workgroups are persistent over a
// chunk of an episode's bands, so the sums stay in registers and leave as one partial slab per workgroup.
#include "conv4.h"

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 cf(const float* coef, int b, int field, int c4) { return ld4(coef + ((long)b * CF_N + field) * 64 + 4 * c4); }

enum { C1_POOL = 0, C1_REDUCE = 1, C1_WGRAD = 2, C1_STATS = 3 };

struct Band { int NPX, NP, ntile; };          // pixels of a band padded to whole 32-pixel tiles; image slab incl. halo

// The image slab of a band (padded rows y0+1, y0+2 plus halo, one plane per channel).  Which slab element a thread copies,
// its column and its row relative to the band do not depend on the band: worked out once (SlabPlan); per band a thread only
// adds the row offset, so the loads of the NEXT band can be issued before the current band's matrix work (prefetch).
constexpr int C1_NE = 8;                              // slab elements per thread: Cin * NP <= 2048
struct SlabPlan { int base[C1_NE]; int dy[C1_NE]; bool okx[C1_NE]; };
__device__ __forceinline__ void slab_plan(SlabPlan& sp, const C1Args& a, const Band& bd) {
    const CvGeom& g = a.g;
    const float rWp = 1.0f / (float)g.Wp;
#pragma unroll
    for (int e = 0; e < C1_NE; ++e) {
        const int i = threadIdx.x + 256 * e;
        const int c = i >= 2 * bd.NP ? 2 : i >= bd.NP ? 1 : 0, pi = i - c * bd.NP;
        int q, x;
        cv_divmod(pi - g.halo + 2 * g.Wp, g.Wp, rWp, q, x);            // slab pixel 0 lies halo = Wp + 1 pixels before the band
        sp.dy[e] = q - 2;                                              // row relative to the band's first row
        sp.okx[e] = i < a.Cin * bd.NP && x >= 1 && x <= g.W;
        sp.base[e] = c * g.H * g.W + (x - 1);
    }
}
__device__ __forceinline__ void slab_load(float (&v)[C1_NE], const SlabPlan& sp, const C1Args& a, const float* img, int y0) {
#pragma unroll
    for (int e = 0; e < C1_NE; ++e) {
        const int yi = y0 + sp.dy[e];
        const bool ok = sp.okx[e] && yi >= 0 && yi < a.g.H;
        const float t = img[ok ? sp.base[e] + yi * a.g.W : 0];           // unconditional load from a clamped address
        v[e] = ok ? t : 0.f;
    }
}
__device__ __forceinline__ void slab_store(float* pl, const float (&v)[C1_NE], const C1Args& a, const Band& bd) {
#pragma unroll
    for (int e = 0; e < C1_NE; ++e) {
        const int i = threadIdx.x + 256 * e;
        if (i < a.Cin * bd.NP) pl[i] = v[e];
    }
}

// this lane's weight fragments of block 1 (k pairs m = 0 .. 13, both column tiles): loaded once per workgroup
constexpr int C1_NK = 14;                             // ceil(27 / 2): Cin <= 3
struct W1Frag { float w0[C1_NK], w1[C1_NK]; int off[C1_NK]; };
__device__ __forceinline__ void load_w1(W1Frag& f, const float* fr, const C1Args& a, const Band& bd) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int nk = (a.Cin * 9 + 1) >> 1;
#pragma unroll
    for (int m = 0; m < C1_NK; ++m) {
        const int kap = 2 * m + h;
        const int c = kap / 9, tap = kap - c * 9;
        const bool ok = m < nk && kap < a.Cin * 9;
        f.w0[m] = ok ? fr[(m * 2 + 0) * 64 + lane] : 0.f;
        f.w1[m] = ok ? fr[(m * 2 + 1) * 64 + lane] : 0.f;
        f.off[m] = ok ? c * bd.NP + (tap / 3 - 1) * a.g.Wp + (tap % 3 - 1) : 0;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Everything stays in the accumulators.  The MFMA's 32 output rows are ours to assign: row 4 w + k of a tile is element k
// (= (dy, dx) in PyTorch's scan order) of 2x2 block w, so after the product a lane (channel = lane & 31, half h = lane >> 5)
// holds in registers 4j .. 4j+3 the FOUR pixels of block 2j + h -- a whole pooling window per register quad.  Batch
// statistics, BN + ReLU + max-pool, the backward reductions and du are then per-lane arithmetic on registers, and for the
// weight gradient the same registers ARE the A operand of the next MFMA (row = channel = lane & 31, k = pixel of lane half
// h): u and du never exist in memory of any kind.  LDS holds only the 3-channel image slab (4 KiB).
// Work units = (tile of 8 blocks, 32-channel half); a wave's units all have channel half (wave & 1).
// ------------------------------------------------------------------------------------------------------------
struct LaneCoef { float A, C0, MU, R, GR, D1, D2, TA, TB, TC, M1, M2, K0, DD1, E12; };
__device__ __forceinline__ float cf1(const float* coef, int b, int field, int c) { return coef[((long)b * CF_N + field) * 64 + c]; }

template <int MODE, bool TAN>
__global__ __launch_bounds__(256) void c1_kernel(C1Args a, int bands_per_img, int chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const CvGeom& g = a.g;
    Band bd;
    const int Wb = (g.W + 1) / 2;                     // 2x2 blocks per band (the last one is half a block when W is odd)
    bd.ntile = (Wb + 7) / 8; bd.NPX = 0; bd.NP = 2 * g.Wp + 2 * g.halo + 32;
    float* pl = lds;                                  // [Cin][NP] image slab of the band
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 1, co = ct * 32 + r;
    const int b = blockIdx.y, t = blockIdx.x;
    const long nbands = (long)a.M * bands_per_img;
    const long b_beg = (long)t * chunk, b_end = min(nbands, b_beg + (long)chunk);
    W1Frag wf, wfd;
    load_w1(wf, a.frag + (long)b * a.frag_stride, a, bd);
    if (TAN) load_w1(wfd, a.fragd + (long)b * a.fragd_stride, a, bd);
    const int Wo = a.Wo, Ho = a.Ho, Wp = g.Wp, W = g.W;
    LaneCoef q;
    if (MODE != C1_STATS) {
        q.A = cf1(a.coef, b, CF_A, co); q.C0 = cf1(a.coef, b, CF_C0, co); q.MU = cf1(a.coef, b, CF_MU, co); q.R = cf1(a.coef, b, CF_R, co);
        q.GR = cf1(a.coef, b, CF_GR, co);
        if (MODE == C1_WGRAD) { q.D1 = cf1(a.coef, b, CF_D1, co); q.D2 = cf1(a.coef, b, CF_D2, co); }
        if (TAN) {
            q.TA = cf1(a.coef, b, CF_TA, co); q.TB = cf1(a.coef, b, CF_TB, co); q.TC = cf1(a.coef, b, CF_TC, co);
            q.M1 = cf1(a.coef, b, CF_M1, co); q.M2 = cf1(a.coef, b, CF_M2, co);
            if (MODE == C1_WGRAD) { q.K0 = cf1(a.coef, b, CF_K0, co); q.DD1 = cf1(a.coef, b, CF_DD1, co); q.E12 = cf1(a.coef, b, CF_E12, co); }
        }
    }
    // conv A operand: MFMA row rho = lane & 31 = 4 (block within tile) + element
    const int a_base = g.halo + 1 + ((r & 3) >> 1) * Wp + 2 * (r >> 2) + (r & 1);
    // wgrad B operand: column kappa = lane & 31; pixel of register 4j + k of lane half h = block 2j + h, element k
    const int kc = r / 9, ktap = r - kc * 9;
    const bool kok = r < a.Cin * 9;
    const int b_base = (kok ? kc * bd.NP + (ktap / 3 - 1) * Wp + (ktap % 3 - 1) : 0) + g.halo + 1 + 2 * h;
    float st1 = 0.f, st2 = 0.f, st3 = 0.f;            // per-lane sums (C1_STATS: 2, C1_REDUCE: 2 or 3)
    f32x16 wacc;                                      // C1_WGRAD: dW1[co half][kappa], this wave's pixels
#pragma unroll
    for (int i = 0; i < 16; ++i) wacc[i] = 0.f;

    SlabPlan sp;
    slab_plan(sp, a, bd);
    float slab[C1_NE];
    const long img_sz = (long)a.Cin * g.H * g.W;
    if (b_beg < b_end) {
        const int im = (int)(b_beg / bands_per_img), yb = (int)(b_beg - (long)im * bands_per_img);
        slab_load(slab, sp, a, a.img + ((long)b * a.M + im) * img_sz, 2 * yb);
    }
    for (long band = b_beg; band < b_end; ++band) {
        const int im = (int)(band / bands_per_img), yb = (int)(band - (long)im * bands_per_img), y0 = 2 * yb;
        const long img_g = (long)b * a.M + im;
        const bool two = y0 + 1 < g.H, full_row = yb < Ho;          // second row exists; the band is a pooled row
        __syncthreads();                                               // the previous band's slab is no longer read
        slab_store(pl, slab, a, bd);
        if (band + 1 < b_end) {                                        // next band's image rows: in flight during this band's work
            const int im2 = (int)((band + 1) / bands_per_img), yb2 = (int)(band + 1 - (long)im2 * bands_per_img);
            slab_load(slab, sp, a, a.img + ((long)b * a.M + im2) * img_sz, 2 * yb2);
        }
        __syncthreads();
        const long orow = (img_g * a.gn.Pp + (long)(yb + 1) * a.gn.Wp + 1) * 64 + co;     // pooled row yb, column 0, channel co

        for (int u = wave; u < 2 * bd.ntile; u += 4) {
            const int tl = u >> 1;
            f32x16 acc, dac;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[i] = 0.f; dac[i] = 0.f; }
            const int pbase = a_base + 16 * tl;
            // gradients w.r.t. this lane's four pooled outputs: requested before the matrix work (unconditional, clamped column)
            float gx[4] = {0.f, 0.f, 0.f, 0.f}, gxd[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == C1_REDUCE || MODE == C1_WGRAD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long o = orow + (long)min(tl * 8 + 2 * j + h, Wo - 1) * 64;
                    gx[j] = a.dxo[o];
                    if (TAN) gxd[j] = a.dxod[o];
                }
            }
#pragma unroll
            for (int m = 0; m < C1_NK; ++m) {
                const float av = pl[wf.off[m] + pbase];
                acc = mfma32(av, ct ? wf.w1[m] : wf.w0[m], acc);
                if (TAN) dac = mfma32(av, ct ? wfd.w1[m] : wfd.w0[m], dac);
            }
            float dreg[16];                                            // C1_WGRAD: du (du') of this lane's 16 pixels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xo = tl * 8 + 2 * j + h;                     // block column
                const bool vb = xo < Wb, vx = 2 * xo + 1 < W;
                const bool val[4] = {vb, vb && vx, vb && two, vb && vx && two};
                const bool full = full_row && xo < Wo;
                const float u0 = acc[4 * j], u1 = acc[4 * j + 1], u2 = acc[4 * j + 2], u3 = acc[4 * j + 3];
                if (MODE == C1_STATS) {
                    // (tangent pass: acc = u, dac = u' -- sums of u' and u u')
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float uu = val[k] ? acc[4 * j + k] : 0.f;
                        if (!TAN) { st1 += uu; st2 += uu * uu; }
                        else { const float ud = val[k] ? dac[4 * j + k] : 0.f; st1 += ud; st2 += ud * uu; }
                    }
                    continue;
                }
                // arg-max of the window (first maximum in scan order), ReLU mask
                float best = q.A * u0 + q.C0; int arg = 0;
                { const float v = q.A * u1 + q.C0; if (v > best) { best = v; arg = 1; } }
                { const float v = q.A * u2 + q.C0; if (v > best) { best = v; arg = 2; } }
                { const float v = q.A * u3 + q.C0; if (v > best) { best = v; arg = 3; } }
                const bool pos = best > 0.f;
                const float ua = arg == 0 ? u0 : arg == 1 ? u1 : arg == 2 ? u2 : u3;
                const float xha = (ua - q.MU) * q.R;
                float uda = 0.f;
                if (TAN) uda = arg == 0 ? dac[4 * j] : arg == 1 ? dac[4 * j + 1] : arg == 2 ? dac[4 * j + 2] : dac[4 * j + 3];
                if (MODE == C1_POOL) {
                    if (full) {
                        if (!TAN) a.x[orow + (long)xo * 64] = pos ? best : 0.f;
                        else a.xd[orow + (long)xo * 64] = pos ? q.TA * uda + q.TB * xha + q.TC : 0.f;
                    }
                }
                if (MODE == C1_REDUCE) {
                    if (full) {
                        const float dv = pos ? gx[j] : 0.f;
                        if (!TAN) { st1 += dv; st2 += dv * xha; }
                        else {
                            const float dvd = pos ? gxd[j] : 0.f;
                            const float xhd = q.R * (uda - q.M1 - xha * q.M2);
                            st1 += dvd; st2 += dvd * xha; st3 += dv * xhd;
                        }
                    }
                }
                if (MODE == C1_WGRAD) {
                    const float dxo = full ? gx[j] : 0.f, dxod = full ? gxd[j] : 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float uu = acc[4 * j + k];
                        const float xh = (uu - q.MU) * q.R;
                        const bool hit = full && pos && arg == k;
                        const float dv = hit ? dxo : 0.f;
                        const float base = dv - q.D1 - xh * q.D2;
                        float o;
                        if (!TAN) o = q.GR * base;
                        else {
                            const float xhd = q.R * (dac[4 * j + k] - q.M1 - xh * q.M2);
                            const float dvd = hit ? dxod : 0.f;
                            o = q.K0 * base + q.GR * (dvd - q.DD1 - xhd * q.D2 - xh * q.E12);
                        }
                        dreg[4 * j + k] = val[k] ? o : 0.f;
                    }
                }
            }
            if (MODE == C1_WGRAD) {
                // dW1[co][kappa] += sum over the unit's 32 pixels: A = du straight from the registers, B = the image slab
                const float* bp = pl + b_base + 16 * tl;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float bv = kok ? bp[((i & 3) >> 1) * Wp + 4 * (i >> 2) + (i & 1)] : 0.f;
                    wacc = mfma32(dreg[i], bv, wacc);
                }
            }
        }

        if (MODE == C1_POOL && full_row) {
            // zero border of the pooled map: columns 0 and Wo+1 of this row; the top / bottom rows once per image
            float* xb = (TAN ? a.xd : a.x) + img_g * a.gn.Pp * 64;
            if (tid < 128) xb[((long)(yb + 1) * a.gn.Wp + (tid >> 6) * (Wo + 1)) * 64 + (tid & 63)] = 0.f;
            if (yb == 0 || yb == Ho - 1)
                for (int it = tid; it < a.gn.Wp * 64; it += 256) {
                    if (yb == 0) xb[it] = 0.f;
                    if (yb == Ho - 1) xb[(long)(Ho + 1) * a.gn.Wp * 64 + it] = 0.f;
                }
        }
    }

    if (MODE == C1_STATS || MODE == C1_REDUCE) {
        constexpr int RK = (MODE == C1_REDUCE && TAN) ? 3 : 2;
        st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32); st3 += __shfl_xor(st3, 32);
        __syncthreads();
        float* red = lds;                                   // [4 waves][3][32]
        if (h == 0) { red[(wave * 3 + 0) * 32 + r] = st1; red[(wave * 3 + 1) * 32 + r] = st2; red[(wave * 3 + 2) * 32 + r] = st3; }
        __syncthreads();
        if (tid < RK * 64) {                                // k = tid >> 6, channel c = tid & 63: waves (c >> 5) and (c >> 5) + 2
            const int k = tid >> 6, c = tid & 63, w0 = c >> 5;
            a.part[(((long)b * gridDim.x + t) * RK + k) * 64 + c] = red[(w0 * 3 + k) * 32 + (c & 31)] + red[((w0 + 2) * 3 + k) * 32 + (c & 31)];
        }
    }
    if (MODE == C1_WGRAD) {
        __syncthreads();
        float* red = lds;                                   // [4 waves][32 co of the wave's half][32 kappa]
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            red[(wave * 32 + row) * 32 + r] = wacc[i];
        }
        __syncthreads();
        float* o = a.wpart + ((long)b * gridDim.x + t) * 2048;
        for (int i = tid; i < 2048; i += 256) {             // output row co = i >> 5: half (co >> 5) = waves (co >> 5) and (co >> 5) + 2
            const int cc = i >> 5, kk = i & 31, w0 = cc >> 5;
            o[i] = red[(w0 * 32 + (cc & 31)) * 32 + kk] + red[((w0 + 2) * 32 + (cc & 31)) * 32 + kk];
        }
    }
}

}  // namespace

int c1_chunks(int B, int M, const CvGeom& g, int* chunk_out) {
    const long nbands = (long)M * ((g.H + 1) / 2);
    long nt = (1024 + B - 1) / B;
    if (nt > nbands) nt = nbands;
    if (nt < 1) nt = 1;
    const long chunk = (nbands + nt - 1) / nt;
    *chunk_out = (int)chunk;
    return (int)((nbands + chunk - 1) / chunk);
}

int launch_c1(hipStream_t st, const C1Args& a, int mode, int tangent) {
    if (a.Cin < 1 || a.Cin > 3) return FUMI_EINVAL;
    int chunk;
    const int nt = c1_chunks(a.B, a.M, a.g, &chunk);
    const int NP = 2 * a.g.Wp + 2 * a.g.halo + 32;
    size_t lds = (size_t)a.Cin * NP * 4;
    if (lds < 16384) lds = 16384;                                           // (the end-of-kernel reductions use up to 16 KiB)
    if (lds > 160 * 1024 || a.Cin * NP > 256 * 8) return FUMI_ENOTSUP;
    const int bpi = (a.g.H + 1) / 2;
    const dim3 grid(nt, a.B), blk(256);
#define C1_LAUNCH(MODE, TAN)                                                            \
    do {                                                                                \
        FUMI_SET_DYN_LDS((c1_kernel<MODE, TAN>), lds);                                  \
        hipLaunchKernelGGL((c1_kernel<MODE, TAN>), grid, blk, lds, st, a, bpi, chunk); \
    } while (0)
    if (mode == C1_POOL) { if (tangent) C1_LAUNCH(C1_POOL, true); else C1_LAUNCH(C1_POOL, false); }
    else if (mode == C1_REDUCE) { if (tangent) C1_LAUNCH(C1_REDUCE, true); else C1_LAUNCH(C1_REDUCE, false); }
    else if (mode == C1_WGRAD) { if (tangent) C1_LAUNCH(C1_WGRAD, true); else C1_LAUNCH(C1_WGRAD, false); }
    else if (mode == C1_STATS) { if (tangent) C1_LAUNCH(C1_STATS, true); else C1_LAUNCH(C1_STATS, false); }
    else return FUMI_EINVAL;
#undef C1_LAUNCH
    LAUNCH_CHECK();
    return FUMI_OK;
}
