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

// u[2 Wp][64] (LDS) = conv(image band, weights).  Work units = (32-pixel tile, 32-channel half): 12 units at Wp = 86, three per
// wave (whole tiles would leave two of the four waves idle in the second round).
__device__ __forceinline__ void conv_band(float* U, const float* pl, const C1Args& a, const Band& bd, const W1Frag& f) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npx = 2 * a.g.Wp;
    for (int u = wave; u < 2 * bd.ntile; u += 4) {
        const int t = u >> 1, ct = u & 1;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int pbase = a.g.halo + t * 32 + r;
#pragma unroll
        for (int m = 0; m < C1_NK; ++m) acc = mfma32(pl[f.off[m] + pbase], ct ? f.w1[m] : f.w0[m], acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int px = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (px < npx) U[px * 64 + ct * 32 + r] = acc[i];
        }
    }
}

// batch statistics of a band straight from the accumulators (nothing goes to LDS): s1 += out, s2 += out * dot over interior
// pixels, out = conv(image, f) and dot = out (plain) or conv(image, fdot) (tangent pass: out = u', dot = u).  A wave's units all
// have the same channel half (unit u = wave mod 4), so a lane keeps two scalars.
template <bool TAN>
__device__ __forceinline__ void conv_band_stats(const float* pl, const C1Args& a, const Band& bd, const W1Frag& f, const W1Frag& fdot,
                                                bool two, float& s1, float& s2) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Wp = a.g.Wp, W = a.g.W;
    for (int u = wave; u < 2 * bd.ntile; u += 4) {
        const int t = u >> 1, ct = u & 1;
        f32x16 acc, dac;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] = 0.f; dac[i] = 0.f; }
        const int pbase = a.g.halo + t * 32 + r;
#pragma unroll
        for (int m = 0; m < C1_NK; ++m) {
            const float av = pl[f.off[m] + pbase];
            acc = mfma32(av, ct ? f.w1[m] : f.w0[m], acc);
            if (TAN) dac = mfma32(av, ct ? fdot.w1[m] : fdot.w0[m], dac);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int px = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const bool row1 = px >= Wp;
            const int x = row1 ? px - Wp : px;
            const bool in = px < 2 * Wp && x >= 1 && x <= W && (!row1 || two);
            const float v = in ? acc[i] : 0.f;
            s1 += v; s2 += v * (TAN ? dac[i] : v);
        }
    }
}

struct Win4 { f32x4 u[4]; };
__device__ __forceinline__ void lds_window(const float* U, int Wp, int xo, int c4, f32x4 (&w)[4]) {
    const float* p = U + (2 * xo + 1) * 64 + 4 * c4;
    w[0] = ld4(p); w[1] = ld4(p + 64); w[2] = ld4(p + Wp * 64); w[3] = ld4(p + Wp * 64 + 64);
}
struct ArgMax { int arg[4]; bool pos[4]; };
__device__ __forceinline__ ArgMax window_argmax(const f32x4 (&u)[4], const f32x4& A, const f32x4& C0, f32x4* vmax = nullptr) {
    ArgMax m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float best = A[e] * u[0][e] + C0[e];
        int arg = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float v = A[e] * u[k][e] + C0[e];
            if (v > best) { best = v; arg = k; }
        }
        m.arg[e] = arg; m.pos[e] = best > 0.f;
        if (vmax) (*vmax)[e] = best;
    }
    return m;
}
__device__ __forceinline__ float pick(const f32x4 (&w)[4], int g, int k) { return g == 0 ? w[0][k] : g == 1 ? w[1][k] : g == 2 ? w[2][k] : w[3][k]; }

template <int MODE, bool TAN>
__global__ __launch_bounds__(256) void c1_kernel(C1Args a, int bands_per_img, int chunk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const CvGeom& g = a.g;
    Band bd;
    bd.ntile = (2 * g.Wp + 31) / 32; bd.NPX = bd.ntile * 32; bd.NP = bd.NPX + 2 * g.halo;
    const int npx = 2 * g.Wp;                         // pixels of a band (two padded rows)
    float* U = lds;                                   // [npx][64]
    float* UD = lds + npx * 64;                       // [npx][64] (TAN)
    float* pl = lds + (TAN ? 2 : 1) * npx * 64;       // [Cin][NP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, t = blockIdx.x;
    const long nbands = (long)a.M * bands_per_img;
    const long b_beg = (long)t * chunk, b_end = min(nbands, b_beg + (long)chunk);
    W1Frag wf, wfd;
    load_w1(wf, a.frag + (long)b * a.frag_stride, a, bd);
    if (TAN) load_w1(wfd, a.fragd + (long)b * a.fragd_stride, a, bd);
    const int Wo = a.Wo, Ho = a.Ho, Wp = g.Wp;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // ---- per-mode persistent state
    constexpr int RK = TAN ? 3 : 2;
    f32x4 rs[3][RK];                                  // C1_REDUCE: sums of this thread's (up to 3) window columns
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < RK; ++k) rs[i][k] = z4;
    float st1 = 0.f, st2 = 0.f;                       // C1_STATS: this lane's channel (32 (wave & 1) + lane & 31), its pixels
    f32x16 wa0, wa1;                                  // C1_WGRAD: dW1 quadrants [co 0..31 | 32..63] x [kappa 0..31], this wave's pixels
#pragma unroll
    for (int i = 0; i < 16; ++i) { wa0[i] = 0.f; wa1[i] = 0.f; }
    const int r = lane & 31, h = lane >> 5;
    const int kc = r / 9, ktap = r - kc * 9;
    const bool kok = r < a.Cin * 9;
    const int koff = kok ? kc * bd.NP + g.halo + (ktap / 3 - 1) * Wp + (ktap % 3 - 1) : 0;

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
        __syncthreads();                                               // previous band's LDS images are no longer read
        slab_store(pl, slab, a, bd);
        if (band + 1 < b_end) {                                        // next band's image rows: in flight during this band's work
            const int im2 = (int)((band + 1) / bands_per_img), yb2 = (int)(band + 1 - (long)im2 * bands_per_img);
            slab_load(slab, sp, a, a.img + ((long)b * a.M + im2) * img_sz, 2 * yb2);
        }
        __syncthreads();
        if (MODE == C1_STATS) {                                        // (tangent: statistics of u' = conv(image, W1') against u)
            if (TAN) conv_band_stats<true>(pl, a, bd, wfd, wf, two, st1, st2);
            else conv_band_stats<false>(pl, a, bd, wf, wf, two, st1, st2);
            continue;
        }
        conv_band(U, pl, a, bd, wf);
        if (TAN) conv_band(UD, pl, a, bd, wfd);
        __syncthreads();

        if (MODE == C1_POOL) {
            // items: padded output columns 0 .. Wo+1 (x borders = 0), 16 channel quads each; plus the top / bottom border rows
            if (full_row) {
                float* xo_row = (TAN ? a.xd : a.x) + (img_g * a.gn.Pp + (long)(yb + 1) * a.gn.Wp) * 64;
                for (int it = tid; it < a.gn.Wp * 16; it += 256) {
                    const int xp = it >> 4, c4 = it & 15;
                    f32x4 out = z4;
                    if (xp >= 1 && xp <= Wo) {
                        f32x4 u[4];
                        lds_window(U, Wp, xp - 1, c4, u);
                        f32x4 vmax;
                        const ArgMax m = window_argmax(u, cf(a.coef, b, CF_A, c4), cf(a.coef, b, CF_C0, c4), &vmax);
                        if (!TAN) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) out[k] = vmax[k] > 0.f ? vmax[k] : 0.f;
                        } else {
                            f32x4 ud[4];
                            lds_window(UD, Wp, xp - 1, c4, ud);
                            const f32x4 mu = cf(a.coef, b, CF_MU, c4), rr = cf(a.coef, b, CF_R, c4);
                            const f32x4 TA = cf(a.coef, b, CF_TA, c4), TB = cf(a.coef, b, CF_TB, c4), TC = cf(a.coef, b, CF_TC, c4);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                out[k] = m.pos[k] ? TA[k] * pick(ud, m.arg[k], k) + TB[k] * ((pick(u, m.arg[k], k) - mu[k]) * rr[k]) + TC[k] : 0.f;
                        }
                    }
                    *(f32x4*)(xo_row + xp * 64 + 4 * c4) = out;
                }
                if (yb == 0 || yb == Ho - 1) {
                    float* base = (TAN ? a.xd : a.x) + img_g * a.gn.Pp * 64;
                    for (int it = tid; it < a.gn.Wp * 16; it += 256) {
                        if (yb == 0) *(f32x4*)(base + it * 4) = z4;
                        if (yb == Ho - 1) *(f32x4*)(base + ((long)(Ho + 1) * a.gn.Wp) * 64 + it * 4) = z4;
                    }
                }
            }
        }

        if (MODE == C1_REDUCE) {
            if (full_row) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int it = tid + 256 * i;
                    if (it < Wo * 16) {
                        const int xo = it >> 4, c4 = it & 15;
                        f32x4 u[4];
                        lds_window(U, Wp, xo, c4, u);
                        const ArgMax m = window_argmax(u, cf(a.coef, b, CF_A, c4), cf(a.coef, b, CF_C0, c4));
                        const f32x4 mu = cf(a.coef, b, CF_MU, c4), rr = cf(a.coef, b, CF_R, c4);
                        const long dpix = (img_g * a.gn.Pp + (long)(yb + 1) * a.gn.Wp + (xo + 1)) * 64 + 4 * c4;
                        const f32x4 dxo = ld4(a.dxo + dpix);
                        f32x4 ud[4], dxod = z4, M1 = z4, M2 = z4;
                        if (TAN) {
                            lds_window(UD, Wp, xo, c4, ud);
                            dxod = ld4(a.dxod + dpix);
                            M1 = cf(a.coef, b, CF_M1, c4); M2 = cf(a.coef, b, CF_M2, c4);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float xh = (pick(u, m.arg[k], k) - mu[k]) * rr[k];
                            const float dv = m.pos[k] ? dxo[k] : 0.f;
                            if (!TAN) { rs[i][0][k] += dv; rs[i][1][k] += dv * xh; }
                            else {
                                const float xhd = rr[k] * (pick(ud, m.arg[k], k) - M1[k] - xh * M2[k]);
                                const float dvd = m.pos[k] ? dxod[k] : 0.f;
                                rs[i][0][k] += dvd; rs[i][1][k] += dvd * xh; rs[i][2][k] += dv * xhd;
                            }
                        }
                    }
                }
            }
        }

        if (MODE == C1_WGRAD) {
            // du (or du') of every pixel of the band, in place over U (UD); border / padding / missing-row pixels = 0
            float* D = TAN ? UD : U;
            const int Wb = (g.W + 1) / 2;
            for (int it = tid; it < Wb * 16; it += 256) {
                const int xb = it >> 4, c4 = it & 15;
                const bool vx = 2 * xb + 1 < g.W, full = full_row && xb < Wo;
                const int p00 = 2 * xb + 1;
                const int offs[4] = {p00, p00 + 1, Wp + p00, Wp + p00 + 1};
                const bool valid[4] = {true, vx, two, vx && two};
                f32x4 u[4], ud[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) u[k] = ld4(U + offs[k] * 64 + 4 * c4);
                if (TAN) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) ud[k] = ld4(UD + offs[k] * 64 + 4 * c4);
                }
                const f32x4 mu = cf(a.coef, b, CF_MU, c4), rr = cf(a.coef, b, CF_R, c4);
                const f32x4 D1 = cf(a.coef, b, CF_D1, c4), D2 = cf(a.coef, b, CF_D2, c4), GR = cf(a.coef, b, CF_GR, c4);
                f32x4 dxo = z4, dxod = z4;
                ArgMax m;
#pragma unroll
                for (int k = 0; k < 4; ++k) { m.arg[k] = -1; m.pos[k] = false; }
                if (full) {
                    const long dpix = (img_g * a.gn.Pp + (long)(yb + 1) * a.gn.Wp + (xb + 1)) * 64 + 4 * c4;
                    dxo = ld4(a.dxo + dpix);
                    if (TAN) dxod = ld4(a.dxod + dpix);
                    m = window_argmax(u, cf(a.coef, b, CF_A, c4), cf(a.coef, b, CF_C0, c4));
                }
                f32x4 M1 = z4, M2 = z4, K0 = z4, DD1 = z4, E12 = z4;
                if (TAN) {
                    M1 = cf(a.coef, b, CF_M1, c4); M2 = cf(a.coef, b, CF_M2, c4); K0 = cf(a.coef, b, CF_K0, c4);
                    DD1 = cf(a.coef, b, CF_DD1, c4); E12 = cf(a.coef, b, CF_E12, c4);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float xh = (u[k][c] - mu[c]) * rr[c];
                        const bool hit = m.arg[c] == k && m.pos[c];
                        const float dv = hit ? dxo[c] : 0.f;
                        const float base = dv - D1[c] - xh * D2[c];
                        if (!TAN) o[c] = GR[c] * base;
                        else {
                            const float xhd = rr[c] * (ud[k][c] - M1[c] - xh * M2[c]);
                            const float dvd = hit ? dxod[c] : 0.f;
                            o[c] = K0[c] * base + GR[c] * (dvd - DD1[c] - xhd * D2[c] - xh * E12[c]);
                        }
                    }
                    if (k == 0 || valid[k]) *(f32x4*)(D + offs[k] * 64 + 4 * c4) = valid[k] ? o : z4;
                }
            }
            // zero what is not an interior pixel of an existing row: the four border pixels (x = 0, x = Wp-1), a missing row
            if (tid < 64) {
                const int k = tid >> 4, c4 = tid & 15;
                *(f32x4*)(D + ((k >> 1) * Wp + (k & 1) * (Wp - 1)) * 64 + 4 * c4) = z4;
            }
            if (!two) for (int it = tid; it < Wp * 16; it += 256) *(f32x4*)(D + (Wp + (it >> 4)) * 64 + 4 * (it & 15)) = z4;
            __syncthreads();
            // dW1[co][kappa] += sum_px du[px][co] * image[kappa][px + off]: pixel pairs round-robin over the waves
            const float* ap = D + h * 64 + r;
            const float* bp = pl + koff + h;
            for (int m = wave; m < Wp; m += 4) {                  // 2 Wp pixels = Wp pairs
                const float bv = kok ? bp[2 * m] : 0.f;
                wa0 = mfma32(ap[m * 128], bv, wa0);
                wa1 = mfma32(ap[m * 128 + 32], bv, wa1);
            }
        }
    }

    if (MODE == C1_STATS) {
        st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32);
        __syncthreads();
        float* red = lds;                                   // [4 waves][2][32]
        if (h == 0) { red[(wave * 2 + 0) * 32 + r] = st1; red[(wave * 2 + 1) * 32 + r] = st2; }
        __syncthreads();
        if (tid < 128) {                                    // k = tid >> 6, channel c = tid & 63: waves (c >> 5) and (c >> 5) + 2
            const int k = tid >> 6, c = tid & 63, w0 = c >> 5;
            a.part[(((long)b * gridDim.x + t) * 2 + k) * 64 + c] = red[(w0 * 2 + k) * 32 + (c & 31)] + red[((w0 + 2) * 2 + k) * 32 + (c & 31)];
        }
    }
    if (MODE == C1_REDUCE) {
        __syncthreads();
        // fold the (up to 3) window columns of a thread, then the 16 threads that share a channel quad
        float* red = lds;                                   // [RK][16 slots][64]
        f32x4 s[RK];
#pragma unroll
        for (int k = 0; k < RK; ++k) s[k] = rs[0][k] + rs[1][k] + rs[2][k];
        const int c4 = tid & 15, slot = tid >> 4;
#pragma unroll
        for (int k = 0; k < RK; ++k) *(f32x4*)(red + (k * 16 + slot) * 64 + 4 * c4) = s[k];
        __syncthreads();
        if (tid < RK * 64) {
            const int k = tid >> 6, c = tid & 63;
            float acc = 0.f;
            for (int sl = 0; sl < 16; ++sl) acc += red[(k * 16 + sl) * 64 + c];
            a.part[(((long)b * gridDim.x + t) * RK + k) * 64 + c] = acc;
        }
    }
    if (MODE == C1_WGRAD) {
        __syncthreads();
        float* red = lds;                                   // [4 waves][64 co][32]
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            red[(wave * 64 + row) * 32 + r] = wa0[i];
            red[(wave * 64 + 32 + row) * 32 + r] = wa1[i];
        }
        __syncthreads();
        float* o = a.wpart + ((long)b * gridDim.x + t) * 2048;
        for (int i = tid; i < 2048; i += 256) o[i] = (red[i] + red[2048 + i]) + (red[4096 + i] + red[6144 + i]);
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
    const int ntile = (2 * a.g.Wp + 31) / 32, NPX = ntile * 32, NP = NPX + 2 * a.g.halo;
    size_t lds = (size_t)(tangent ? 2 : 1) * 2 * a.g.Wp * 256 + (size_t)a.Cin * NP * 4;
    if (lds < 32768) lds = 32768;                                           // (the end-of-kernel reductions use up to 32 KiB)
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
