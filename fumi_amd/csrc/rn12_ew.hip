// Element-wise passes of the bf16 ResNet-12 encoder (rn12.h): batch-statistic BN + LeakyReLU, the residual join + max-pool, their
// first-order backward and the tangent (forward-over-reverse) forms of both -- formulas of oracle/resnet12_manual.py.  HBM-bound:
// a thread owns 8 channels (16 bytes) of one padded pixel; per-(episode, channel) coefficients come from tables the coefficient
// kernel builds from partial sums (all sums in fixed order, the final ones in double: no float atomics, bit-reproducible).
#include "rn12.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct F8 { float v[8]; };

__device__ __forceinline__ F8 unpack8(const u32x4& r) {
    F8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o.v[2 * j] = __uint_as_float(r[j] << 16); o.v[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u); }
    return o;
}
__device__ __forceinline__ u32x4 pack8(const F8& f) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (unsigned)rn_f2bf(f.v[2 * j]) | ((unsigned)rn_f2bf(f.v[2 * j + 1]) << 16);
    return r;
}
__device__ __forceinline__ F8 ldbf(const rbf16* p) { return unpack8(*(const u32x4*)p); }
__device__ __forceinline__ F8 ldcf(const float* coef, int field, int C, int c) {
    F8 o;
    const f32x4 a = *(const f32x4*)(coef + (long)field * C + c), b = *(const f32x4*)(coef + (long)field * C + c + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o.v[j] = a[j]; o.v[4 + j] = b[j]; }
    return o;
}
__device__ __forceinline__ float lmask(float v) { return v > 0.f ? 1.f : RN_SLOPE; }

// pixel index within an episode -> interior flag (and padded coordinates)
__device__ __forceinline__ bool interior_of(long p, const RnGeom& g, int& y, int& x) {
    const unsigned q = (unsigned)(p % g.Pp);
    y = q / (unsigned)g.Wp; x = q - y * g.Wp;
    return y >= 1 && y <= g.H && x >= 1 && x <= g.W;
}

// ---- BN + LeakyReLU ----------------------------------------------------------------------------------------------------------------
// Map passes: a thread owns ONE 8-channel chunk and walks pixels (thread = (row group rg, chunk ch), pixels pbeg + rg, + nrg, ..), so
// the per-channel coefficient vectors are loaded once per thread instead of once per pixel -- the one-pixel-per-thread form moved
// 4.5 TB/s where the reductions, which always had this shape, move 5.6-6.6 (a tangent pass reads 6-11 coefficient vectors = 12-22
// 16-byte loads through the vector L1 per 2-5 16-byte loads of map data).  UNR pixels per iteration keep several loads in flight.
__device__ __forceinline__ bool interior32(unsigned p, const RnGeom& g) {
    const unsigned q = p % (unsigned)g.Pp, y = q / (unsigned)g.Wp, x = q - y * (unsigned)g.Wp;
    return y >= 1 && y <= (unsigned)g.H && x >= 1 && x <= (unsigned)g.W;
}
// (the plain pass has two coefficient vectors and one load per pixel: one pixel per thread moves 6.1 TB/s, the walking form 5.7)
__global__ __launch_bounds__(256) void rn_act1_kernel(RnMap m, const rbf16* u, const float* coef, rbf16* out) {
    const int b = blockIdx.y, nch = m.C >> 3;
    const long npix = (long)m.M * m.g.Pp;
    const long unit = (long)blockIdx.x * 256 + threadIdx.x;
    if (unit >= npix * nch) return;
    const long p = unit / nch; const int c = (int)(unit - p * nch) * 8;
    const long off = ((long)b * npix + p) * m.C + c;
    if (!interior32((unsigned)p, m.g)) { *(u32x4*)(out + off) = (u32x4){0u, 0u, 0u, 0u}; return; }
    const float* cf = coef + (long)b * RCF_N * m.C;
    const F8 uv = ldbf(u + off), A = ldcf(cf, RCF_A, m.C, c), C0 = ldcf(cf, RCF_C0, m.C, c);
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = A.v[j] * uv.v[j] + C0.v[j]; o.v[j] = v > 0.f ? v : RN_SLOPE * v; }
    *(u32x4*)(out + off) = pack8(o);
}
template <bool TAN>
__global__ __launch_bounds__(256) void rn_act_kernel(RnMap m, const rbf16* u, const rbf16* ud, const float* coef, rbf16* out, int RB) {
    constexpr int UNR = TAN ? 2 : 4;
    const int b = blockIdx.y, nch = m.C >> 3, nrg = 256 / nch;
    const unsigned npix = (unsigned)m.M * (unsigned)m.g.Pp;
    const int ch = threadIdx.x % nch, rg = threadIdx.x / nch, c = ch * 8;
    if (rg >= nrg) return;
    const float* cf = coef + (long)b * RCF_N * m.C;
    const F8 A = ldcf(cf, RCF_A, m.C, c), C0 = ldcf(cf, RCF_C0, m.C, c);
    F8 MU, R, TB, TC;
    if (TAN) { MU = ldcf(cf, RCF_MU, m.C, c); R = ldcf(cf, RCF_R, m.C, c); TB = ldcf(cf, RCF_TB, m.C, c); TC = ldcf(cf, RCF_TC, m.C, c); }
    const unsigned pbeg = blockIdx.x * (unsigned)RB, pend = min(npix, pbeg + (unsigned)RB);
    const long base = (long)b * npix * m.C + c;
    for (unsigned p0 = pbeg + rg; p0 < pend; p0 += UNR * nrg) {
        u32x4 uv[UNR], udv[UNR]; bool in[UNR], ok[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const unsigned p = p0 + k * nrg;
            ok[k] = p < pend; in[k] = ok[k] && interior32(p, m.g);
            const long off = base + (long)(ok[k] ? p : pbeg) * m.C;
            uv[k] = *(const u32x4*)(u + off);
            if (TAN) udv[k] = *(const u32x4*)(ud + off);
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (!ok[k]) continue;
            const unsigned p = p0 + k * nrg;
            F8 o;
            const F8 uf = unpack8(uv[k]);
            if (!TAN) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float v = A.v[j] * uf.v[j] + C0.v[j]; o.v[j] = v > 0.f ? v : RN_SLOPE * v; }
            } else {
                const F8 udf = unpack8(udv[k]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = A.v[j] * uf.v[j] + C0.v[j];
                    const float xh = (uf.v[j] - MU.v[j]) * R.v[j];
                    o.v[j] = lmask(v) * (A.v[j] * udf.v[j] + TB.v[j] * xh + TC.v[j]);
                }
            }
            *(u32x4*)(out + base + (long)p * m.C) = in[k] ? pack8(o) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
}

// partial sums over RB consecutive pixels per workgroup: thread (row group rg, chunk ch)
template <bool TAN>
__global__ __launch_bounds__(256) void rn_bwd_reduce_kernel(RnMap m, const rbf16* u, const rbf16* ud, const rbf16* da, const rbf16* dad,
                                                            const float* coef, float* part, int RB) {
    extern __shared__ float red[];                          // [nrg][K][C]
    constexpr int K = TAN ? 3 : 2;
    const int b = blockIdx.y, nch = m.C >> 3, nrg = 256 / nch;
    const long npix = (long)m.M * m.g.Pp;
    const int ch = threadIdx.x % nch, rg = threadIdx.x / nch, c = ch * 8;
    float s[K][8];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[k][j] = 0.f;
    if (rg < nrg) {
        const float* cf = coef + (long)b * RCF_N * m.C;
        const F8 A = ldcf(cf, RCF_A, m.C, c), C0 = ldcf(cf, RCF_C0, m.C, c), MU = ldcf(cf, RCF_MU, m.C, c), R = ldcf(cf, RCF_R, m.C, c);
        F8 M1, M2;
        if (TAN) { M1 = ldcf(cf, RCF_M1, m.C, c); M2 = ldcf(cf, RCF_M2, m.C, c); }
        const long pbeg = (long)blockIdx.x * RB, pend = min(npix, pbeg + RB);
        for (long p = pbeg + rg; p < pend; p += nrg) {
            int y, x;
            if (!interior_of(p, m.g, y, x)) continue;
            const long off = ((long)b * npix + p) * m.C + c;
            const F8 uv = ldbf(u + off), dav = ldbf(da + off);
            F8 udv, dadv;
            if (TAN) { udv = ldbf(ud + off); dadv = ldbf(dad + off); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = A.v[j] * uv.v[j] + C0.v[j], mk = lmask(v);
                const float xh = (uv.v[j] - MU.v[j]) * R.v[j];
                const float dv = dav.v[j] * mk;
                if (!TAN) { s[0][j] += dv; s[1][j] += dv * xh; }
                else {
                    const float dvd = dadv.v[j] * mk;
                    const float xhd = R.v[j] * (udv.v[j] - M1.v[j] - xh * M2.v[j]);
                    s[0][j] += dvd; s[1][j] += dvd * xh; s[2][j] += dv * xhd;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[((long)rg * K + k) * m.C + c + j] = s[k][j];
    }
    __syncthreads();
    float* dst = part + ((long)b * gridDim.x + blockIdx.x) * K * m.C;
    for (int i = threadIdx.x; i < K * m.C; i += 256) {
        float t = 0.f;
        for (int g = 0; g < nrg; ++g) t += red[(long)g * K * m.C + i];
        dst[i] = t;
    }
}

template <bool TAN>
__global__ __launch_bounds__(256) void rn_bwd_apply_kernel(RnMap m, const rbf16* u, const rbf16* ud, const rbf16* da, const rbf16* dad,
                                                           const float* coef, rbf16* du, int RB) {
    constexpr int UNR = 2;
    const int b = blockIdx.y, nch = m.C >> 3, nrg = 256 / nch;
    const unsigned npix = (unsigned)m.M * (unsigned)m.g.Pp;
    const int ch = threadIdx.x % nch, rg = threadIdx.x / nch, c = ch * 8;
    if (rg >= nrg) return;
    const float* cf = coef + (long)b * RCF_N * m.C;
    const F8 A = ldcf(cf, RCF_A, m.C, c), C0 = ldcf(cf, RCF_C0, m.C, c), MU = ldcf(cf, RCF_MU, m.C, c), R = ldcf(cf, RCF_R, m.C, c),
             D1 = ldcf(cf, RCF_D1, m.C, c), D2 = ldcf(cf, RCF_D2, m.C, c);
    F8 M1, M2, K0, DD1, E12;
    if (TAN) {
        M1 = ldcf(cf, RCF_M1, m.C, c); M2 = ldcf(cf, RCF_M2, m.C, c); K0 = ldcf(cf, RCF_K0, m.C, c);
        DD1 = ldcf(cf, RCF_DD1, m.C, c); E12 = ldcf(cf, RCF_E12, m.C, c);
    }
    const unsigned pbeg = blockIdx.x * (unsigned)RB, pend = min(npix, pbeg + (unsigned)RB);
    const long base = (long)b * npix * m.C + c;
    for (unsigned p0 = pbeg + rg; p0 < pend; p0 += UNR * nrg) {
        u32x4 uv[UNR], dav[UNR], udv[UNR], dadv[UNR]; bool in[UNR], ok[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const unsigned p = p0 + k * nrg;
            ok[k] = p < pend; in[k] = ok[k] && interior32(p, m.g);
            const long off = base + (long)(ok[k] ? p : pbeg) * m.C;
            uv[k] = *(const u32x4*)(u + off); dav[k] = *(const u32x4*)(da + off);
            if (TAN) { udv[k] = *(const u32x4*)(ud + off); dadv[k] = *(const u32x4*)(dad + off); }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (!ok[k]) continue;
            const unsigned p = p0 + k * nrg;
            const F8 uf = unpack8(uv[k]), daf = unpack8(dav[k]);
            F8 o;
            if (!TAN) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = A.v[j] * uf.v[j] + C0.v[j];
                    const float xh = (uf.v[j] - MU.v[j]) * R.v[j];
                    o.v[j] = A.v[j] * (daf.v[j] * lmask(v) - D1.v[j] - xh * D2.v[j]);
                }
            } else {
                const F8 udf = unpack8(udv[k]), dadf = unpack8(dadv[k]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = A.v[j] * uf.v[j] + C0.v[j], mk = lmask(v);
                    const float xh = (uf.v[j] - MU.v[j]) * R.v[j];
                    const float xhd = R.v[j] * (udf.v[j] - M1.v[j] - xh * M2.v[j]);
                    const float dv = daf.v[j] * mk, dvd = dadf.v[j] * mk;
                    o.v[j] = K0.v[j] * (dv - D1.v[j] - xh * D2.v[j]) + A.v[j] * (dvd - DD1.v[j] - xhd * D2.v[j] - xh * E12.v[j]);
                }
            }
            *(u32x4*)(du + base + (long)p * m.C) = in[k] ? pack8(o) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
}

// ---- residual join: s = BN3(u3) + BNs(us), o = maxpool2(lrelu(s)) ----------------------------------------------------------------------
struct JoinCoef { F8 A3, C3, As, Cs; };
__device__ __forceinline__ JoinCoef join_coef(const float* c3, const float* cs, int C, int c) {
    JoinCoef k; k.A3 = ldcf(c3, RCF_A, C, c); k.C3 = ldcf(c3, RCF_C0, C, c); k.As = ldcf(cs, RCF_A, C, c); k.Cs = ldcf(cs, RCF_C0, C, c);
    return k;
}
// s of the 2 x 2 window with top-left padded coordinates (y0, x0) of image `img`: sv[w][j]; returns per channel the FIRST arg-max of
// lrelu(s) (monotone: the arg-max of s) in window order (dy, dx) row-major
__device__ __forceinline__ void window_s(const RnJoin& J, const JoinCoef& k, long imgbase, int y0, int x0, int c, float sv[4][8], int arg[8]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const long off = (imgbase + (long)(y0 + (w >> 1)) * J.m.g.Wp + x0 + (w & 1)) * J.m.C + c;
        const F8 a = ldbf(J.u3 + off), s_ = ldbf(J.us + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[w][j] = (k.A3.v[j] * a.v[j] + k.C3.v[j]) + (k.As.v[j] * s_.v[j] + k.Cs.v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int am = 0; float mx = sv[0][j];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (sv[w][j] > mx) { mx = sv[w][j]; am = w; }
        arg[j] = am;
    }
}
// tangent of s at one pixel
__device__ __forceinline__ F8 s_tangent(const RnJoin& J, const float* c3, const float* cs, const JoinCoef& k, long off, int c) {
    const int C = J.m.C;
    const F8 u3 = ldbf(J.u3 + off), us = ldbf(J.us + off), u3d = ldbf(J.u3d + off), usd = ldbf(J.usd + off);
    const F8 MU3 = ldcf(c3, RCF_MU, C, c), R3 = ldcf(c3, RCF_R, C, c), TB3 = ldcf(c3, RCF_TB, C, c), TC3 = ldcf(c3, RCF_TC, C, c);
    const F8 MUs = ldcf(cs, RCF_MU, C, c), Rs = ldcf(cs, RCF_R, C, c), TBs = ldcf(cs, RCF_TB, C, c), TCs = ldcf(cs, RCF_TC, C, c);
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xh3 = (u3.v[j] - MU3.v[j]) * R3.v[j], xhs = (us.v[j] - MUs.v[j]) * Rs.v[j];
        o.v[j] = (k.A3.v[j] * u3d.v[j] + TB3.v[j] * xh3 + TC3.v[j]) + (k.As.v[j] * usd.v[j] + TBs.v[j] * xhs + TCs.v[j]);
    }
    return o;
}

template <bool TAN>
__global__ __launch_bounds__(256) void rn_join_fwd_kernel(RnJoin J, rbf16* o) {
    const int b = blockIdx.y, C = J.m.C, nch = C >> 3;
    const long npo = (long)J.m.M * J.gn.Pp;                          // output pixels per episode (padded grid of the next block)
    const long unit = (long)blockIdx.x * 256 + threadIdx.x;
    if (unit >= npo * nch) return;
    const long po = unit / nch; const int c = (int)(unit - po * nch) * 8;
    const long ooff = ((long)b * npo + po) * C + c;
    const long img = po / J.gn.Pp; const int q = (int)(po - img * J.gn.Pp);
    const int yo = q / J.gn.Wp, xo = q - yo * J.gn.Wp;
    if (yo < 1 || yo > J.Ho || xo < 1 || xo > J.Wo) { *(u32x4*)(o + ooff) = (u32x4){0u, 0u, 0u, 0u}; return; }
    const float* c3 = J.coef3 + (long)b * RCF_N * C; const float* cs = J.coefs + (long)b * RCF_N * C;
    const JoinCoef k = join_coef(c3, cs, C, c);
    const long imgbase = ((long)b * J.m.M + img) * J.m.g.Pp;
    const int y0 = 2 * (yo - 1) + 1, x0 = 2 * (xo - 1) + 1;
    float sv[4][8]; int arg[8];
    window_s(J, k, imgbase, y0, x0, c, sv, arg);
    F8 r;
    if (!TAN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float mx = sv[0][j];
#pragma unroll
            for (int w = 1; w < 4; ++w) mx = sv[w][j] > mx ? sv[w][j] : mx;
            r.v[j] = mx > 0.f ? mx : RN_SLOPE * mx;
        }
    } else {
        F8 sd[4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
            sd[w] = s_tangent(J, c3, cs, k, (imgbase + (long)(y0 + (w >> 1)) * J.m.g.Wp + x0 + (w & 1)) * C + c, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float sa = sv[0][j], da = sd[0].v[j];
#pragma unroll
            for (int w = 1; w < 4; ++w) if (arg[j] == w) { sa = sv[w][j]; da = sd[w].v[j]; }
            r.v[j] = lmask(sa) * da;
        }
    }
    *(u32x4*)(o + ooff) = pack8(r);
}

// Backward of the join.  The pooled gradient goes to the window's arg-max only, so the unit of work is a pooling WINDOW: its four
// pixels of u3 / us are loaded once (a thread per pixel re-read the whole window to find the arg-max: 4x the L1 / L2 traffic, 2.1-2.7
// TB/s), the arg-max and lrelu'(s) at the arg-max are found per channel, then every pixel gets its ds.
struct JoinWin { u32x4 u3[4], us[4]; int am[8]; float lm[8]; };
__device__ __forceinline__ void join_window(const RnJoin& J, const JoinCoef& k, const long off[4], int c, JoinWin& w) {
    float sv[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { w.u3[q] = *(const u32x4*)(J.u3 + off[q]); w.us[q] = *(const u32x4*)(J.us + off[q]); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const F8 a = unpack8(w.u3[q]), s_ = unpack8(w.us[q]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[q][j] = (k.A3.v[j] * a.v[j] + k.C3.v[j]) + (k.As.v[j] * s_.v[j] + k.Cs.v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int am = 0; float mx = sv[0][j];
#pragma unroll
        for (int q = 1; q < 4; ++q) if (sv[q][j] > mx) { mx = sv[q][j]; am = q; }      // the FIRST maximum in window order, as window_s
        w.am[j] = am; w.lm[j] = lmask(mx);
    }
}
// offsets of the window (wy, wx) of image img: its four conv-resolution pixels and its pooled pixel
__device__ __forceinline__ long join_offsets(const RnJoin& J, int b, unsigned img, int wy, int wx, int c, long off[4]) {
    const long imgbase = ((long)b * J.m.M + img) * J.m.g.Pp;
#pragma unroll
    for (int q = 0; q < 4; ++q) off[q] = (imgbase + (long)(2 * wy + 1 + (q >> 1)) * J.m.g.Wp + 2 * wx + 1 + (q & 1)) * J.m.C + c;
    return (((long)b * J.m.M + img) * J.gn.Pp + (long)(wy + 1) * J.gn.Wp + wx + 1) * J.m.C + c;
}

// thread (row group rg, chunk ch) walks windows wbeg + rg, + nrg, ..: per channel only the arg-max pixel has ds != 0
template <bool TAN>
__global__ __launch_bounds__(256) void rn_join_reduce_kernel(RnJoin J, const rbf16* dout, const rbf16* doutd, float* part, unsigned WB) {
    extern __shared__ float red[];
    constexpr int K = TAN ? 5 : 3;
    const int b = blockIdx.y, C = J.m.C, nch = C >> 3, nrg = 256 / nch;
    const unsigned nwi = (unsigned)J.Ho * J.Wo, nwin = (unsigned)J.m.M * nwi;
    const int ch = threadIdx.x % nch, rg = threadIdx.x / nch, c = ch * 8;
    float s[K][8];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[k][j] = 0.f;
    if (rg < nrg) {
        const float* c3 = J.coef3 + (long)b * RCF_N * C; const float* cs = J.coefs + (long)b * RCF_N * C;
        const JoinCoef k = join_coef(c3, cs, C, c);
        const F8 MU3 = ldcf(c3, RCF_MU, C, c), R3 = ldcf(c3, RCF_R, C, c), MUs = ldcf(cs, RCF_MU, C, c), Rs = ldcf(cs, RCF_R, C, c);
        F8 M13, M23, M1s, M2s;
        if (TAN) { M13 = ldcf(c3, RCF_M1, C, c); M23 = ldcf(c3, RCF_M2, C, c); M1s = ldcf(cs, RCF_M1, C, c); M2s = ldcf(cs, RCF_M2, C, c); }
        const unsigned wbeg = blockIdx.x * WB, wend = min(nwin, wbeg + WB);
        for (unsigned wi = wbeg + rg; wi < wend; wi += nrg) {
            const unsigned img = wi / nwi, r = wi - img * nwi;
            const int wy = (int)(r / (unsigned)J.Wo), wx = (int)(r - (unsigned)wy * J.Wo);
            long off[4];
            const long doff = join_offsets(J, b, img, wy, wx, c, off);
            JoinWin w;
            join_window(J, k, off, c, w);
            const F8 dov = ldbf(dout + doff);
            F8 dodv; u32x4 u3d[4], usd[4];
            if (TAN) {
                dodv = ldbf(doutd + doff);
#pragma unroll
                for (int q = 0; q < 4; ++q) { u3d[q] = *(const u32x4*)(J.u3d + off[q]); usd[q] = *(const u32x4*)(J.usd + off[q]); }
            }
            // the arg-max pixel's values, per channel
            F8 u3, us, u3t, ust;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned a3 = w.u3[0][j >> 1], as_ = w.us[0][j >> 1], a3d = 0, asd = 0;
                if (TAN) { a3d = u3d[0][j >> 1]; asd = usd[0][j >> 1]; }
#pragma unroll
                for (int q = 1; q < 4; ++q) if (w.am[j] == q) {
                    a3 = w.u3[q][j >> 1]; as_ = w.us[q][j >> 1];
                    if (TAN) { a3d = u3d[q][j >> 1]; asd = usd[q][j >> 1]; }
                }
                const int sh = (j & 1) ? 0 : 16;                        // (low half = even channel)
                u3.v[j] = __uint_as_float((a3 << sh) & 0xffff0000u); us.v[j] = __uint_as_float((as_ << sh) & 0xffff0000u);
                if (TAN) { u3t.v[j] = __uint_as_float((a3d << sh) & 0xffff0000u); ust.v[j] = __uint_as_float((asd << sh) & 0xffff0000u); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float ds = dov.v[j] * w.lm[j];
                const float xh3 = (u3.v[j] - MU3.v[j]) * R3.v[j], xhs = (us.v[j] - MUs.v[j]) * Rs.v[j];
                if (!TAN) { s[0][j] += ds; s[1][j] += ds * xh3; s[2][j] += ds * xhs; }
                else {
                    const float dsd = dodv.v[j] * w.lm[j];
                    const float xh3d = R3.v[j] * (u3t.v[j] - M13.v[j] - xh3 * M23.v[j]);
                    const float xhsd = Rs.v[j] * (ust.v[j] - M1s.v[j] - xhs * M2s.v[j]);
                    s[0][j] += dsd; s[1][j] += dsd * xh3; s[2][j] += ds * xh3d; s[3][j] += dsd * xhs; s[4][j] += ds * xhsd;
                }
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[((long)rg * K + k2) * C + c + j] = s[k2][j];
    }
    __syncthreads();
    float* dst = part + ((long)b * gridDim.x + blockIdx.x) * K * C;
    for (int i = threadIdx.x; i < K * C; i += 256) {
        float t = 0.f;
        for (int g = 0; g < nrg; ++g) t += red[(long)g * K * C + i];
        dst[i] = t;
    }
}

// (one-cell-per-thread form, kept for the TANGENT pass: with its 22 coefficient vectors the cell-walking form below spills)
// du3 / dus of ONE pixel from its ds (ds' in the tangent pass), coefficients loaded per pixel
template <bool TAN>
__device__ __forceinline__ void join_apply_pixel_ld(const RnJoin& J, const float* c3, const float* cs, const JoinCoef& k, long off, int c,
                                                 const F8& u3, const F8& us, const float ds[8], const float dsd[8], rbf16* du3, rbf16* dus) {
    const int C = J.m.C;
    const F8 MU3 = ldcf(c3, RCF_MU, C, c), R3 = ldcf(c3, RCF_R, C, c), MUs = ldcf(cs, RCF_MU, C, c), Rs = ldcf(cs, RCF_R, C, c);
    const F8 D13 = ldcf(c3, RCF_D1, C, c), D23 = ldcf(c3, RCF_D2, C, c), D1s = ldcf(cs, RCF_D1, C, c), D2s = ldcf(cs, RCF_D2, C, c);
    F8 o3, os;
    if (!TAN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh3 = (u3.v[j] - MU3.v[j]) * R3.v[j], xhs = (us.v[j] - MUs.v[j]) * Rs.v[j];
            o3.v[j] = k.A3.v[j] * (ds[j] - D13.v[j] - xh3 * D23.v[j]);
            os.v[j] = k.As.v[j] * (ds[j] - D1s.v[j] - xhs * D2s.v[j]);
        }
    } else {
        const F8 u3d = ldbf(J.u3d + off), usd = ldbf(J.usd + off);
        const F8 M13 = ldcf(c3, RCF_M1, C, c), M23 = ldcf(c3, RCF_M2, C, c), M1s = ldcf(cs, RCF_M1, C, c), M2s = ldcf(cs, RCF_M2, C, c);
        const F8 K03 = ldcf(c3, RCF_K0, C, c), DD13 = ldcf(c3, RCF_DD1, C, c), E3 = ldcf(c3, RCF_E12, C, c);
        const F8 K0s = ldcf(cs, RCF_K0, C, c), DD1s = ldcf(cs, RCF_DD1, C, c), Es = ldcf(cs, RCF_E12, C, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh3 = (u3.v[j] - MU3.v[j]) * R3.v[j], xhs = (us.v[j] - MUs.v[j]) * Rs.v[j];
            const float xh3d = R3.v[j] * (u3d.v[j] - M13.v[j] - xh3 * M23.v[j]);
            const float xhsd = Rs.v[j] * (usd.v[j] - M1s.v[j] - xhs * M2s.v[j]);
            o3.v[j] = K03.v[j] * (ds[j] - D13.v[j] - xh3 * D23.v[j]) + k.A3.v[j] * (dsd[j] - DD13.v[j] - xh3d * D23.v[j] - xh3 * E3.v[j]);
            os.v[j] = K0s.v[j] * (ds[j] - D1s.v[j] - xhs * D2s.v[j]) + k.As.v[j] * (dsd[j] - DD1s.v[j] - xhsd * D2s.v[j] - xhs * Es.v[j]);
        }
    }
    *(u32x4*)(du3 + off) = pack8(o3);
    *(u32x4*)(dus + off) = pack8(os);
}

// thread = one 2 x 2 CELL of the padded grid x one 8-channel chunk.  Cells start at odd coordinates (cell (cy, cx) = padded pixels
// y in {2 cy - 1, 2 cy}, x in {2 cx - 1, 2 cx}), so a cell is either a whole pooling window (1 <= cy <= Ho, 1 <= cx <= Wo) or lies
// on the rim: border pixels (zeros) and interior pixels no window covers (odd H / W: ds = 0).
template <bool TAN>
__global__ __launch_bounds__(256) void rn_join_apply_cell_kernel(RnJoin J, const rbf16* dout, const rbf16* doutd, rbf16* du3, rbf16* dus,
                                                            int ncy, int ncx) {
    const int b = blockIdx.y, C = J.m.C, nch = C >> 3;
    const unsigned nci = (unsigned)ncy * ncx, ncell = (unsigned)J.m.M * nci;
    const unsigned unit = blockIdx.x * 256u + threadIdx.x;
    if (unit >= ncell * (unsigned)nch) return;
    const unsigned cell = unit / (unsigned)nch; const int c = (int)(unit - cell * nch) * 8;
    const unsigned img = cell / nci, q0 = cell - img * nci;
    const int cy = (int)(q0 / (unsigned)ncx), cx = (int)(q0 - (unsigned)cy * ncx);
    const float* c3 = J.coef3 + (long)b * RCF_N * C; const float* cs = J.coefs + (long)b * RCF_N * C;
    const JoinCoef k = join_coef(c3, cs, C, c);
    float ds[8], dsd[8];
    if (cy >= 1 && cy <= J.Ho && cx >= 1 && cx <= J.Wo) {
        long off[4];
        const long doff = join_offsets(J, b, img, cy - 1, cx - 1, c, off);
        JoinWin w;
        join_window(J, k, off, c, w);
        const F8 dov = ldbf(dout + doff);
        F8 dodv;
        if (TAN) dodv = ldbf(doutd + doff);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (an opaque zero per pixel: without it the 22 coefficient vectors of the tangent form are hoisted over the four pixels --
            // 264 VGPRs, one wave per SIMD; re-loading them per pixel from L1 keeps the occupancy)
            int z;
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float mk = w.am[j] == q ? w.lm[j] : 0.f;
                ds[j] = dov.v[j] * mk; dsd[j] = TAN ? dodv.v[j] * mk : 0.f;
            }
            join_apply_pixel_ld<TAN>(J, c3 + z, cs + z, k, off[q], c, unpack8(w.u3[q]), unpack8(w.us[q]), ds, dsd, du3, dus);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ds[j] = 0.f; dsd[j] = 0.f; }
    const long imgbase = ((long)b * J.m.M + img) * J.m.g.Pp;
    for (int q = 0; q < 4; ++q) {
        const int y = 2 * cy - 1 + (q >> 1), x = 2 * cx - 1 + (q & 1);
        if (y < 0 || y >= J.m.g.Hp || x < 0 || x >= J.m.g.Wp) continue;
        const long off = (imgbase + (long)y * J.m.g.Wp + x) * C + c;
        if (y < 1 || y > J.m.g.H || x < 1 || x > J.m.g.W) {
            *(u32x4*)(du3 + off) = (u32x4){0u, 0u, 0u, 0u}; *(u32x4*)(dus + off) = (u32x4){0u, 0u, 0u, 0u};
            continue;
        }
        join_apply_pixel_ld<TAN>(J, c3, cs, k, off, c, ldbf(J.u3 + off), ldbf(J.us + off), ds, dsd, du3, dus);
    }
}

// du3 / dus of ONE pixel from its ds (ds' in the tangent pass); the coefficient vectors of the thread's channel chunk are loaded
// once per thread (JoinApplyCoef) -- the thread walks cells
template <bool TAN>
struct JoinApplyCoef {
    F8 MU3, R3, MUs, Rs, D13, D23, D1s, D2s;                    // (the ten tangent-only vectors are re-loaded per pixel: with all 22 in
    __device__ __forceinline__ void load(const float* c3, const float* cs, int C, int c) {       //  registers the kernel spills)
        MU3 = ldcf(c3, RCF_MU, C, c); R3 = ldcf(c3, RCF_R, C, c); MUs = ldcf(cs, RCF_MU, C, c); Rs = ldcf(cs, RCF_R, C, c);
        D13 = ldcf(c3, RCF_D1, C, c); D23 = ldcf(c3, RCF_D2, C, c); D1s = ldcf(cs, RCF_D1, C, c); D2s = ldcf(cs, RCF_D2, C, c);
    }
};
template <bool TAN>
__device__ __forceinline__ void join_apply_pixel(const JoinCoef& k, const JoinApplyCoef<TAN>& q, const float* c3, const float* cs, int C, int c,
                                                 const F8& u3, const F8& us, const F8& u3d, const F8& usd, const float ds[8],
                                                 const float dsd[8], rbf16* du3, rbf16* dus, long off) {
    F8 o3, os;
    if (!TAN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh3 = (u3.v[j] - q.MU3.v[j]) * q.R3.v[j], xhs = (us.v[j] - q.MUs.v[j]) * q.Rs.v[j];
            o3.v[j] = k.A3.v[j] * (ds[j] - q.D13.v[j] - xh3 * q.D23.v[j]);
            os.v[j] = k.As.v[j] * (ds[j] - q.D1s.v[j] - xhs * q.D2s.v[j]);
        }
    } else {
        {
            const F8 M13 = ldcf(c3, RCF_M1, C, c), M23 = ldcf(c3, RCF_M2, C, c), K03 = ldcf(c3, RCF_K0, C, c), DD13 = ldcf(c3, RCF_DD1, C, c),
                     E3 = ldcf(c3, RCF_E12, C, c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh3 = (u3.v[j] - q.MU3.v[j]) * q.R3.v[j];
                const float xh3d = q.R3.v[j] * (u3d.v[j] - M13.v[j] - xh3 * M23.v[j]);
                o3.v[j] = K03.v[j] * (ds[j] - q.D13.v[j] - xh3 * q.D23.v[j]) + k.A3.v[j] * (dsd[j] - DD13.v[j] - xh3d * q.D23.v[j] - xh3 * E3.v[j]);
            }
        }
        {
            const F8 M1s = ldcf(cs, RCF_M1, C, c), M2s = ldcf(cs, RCF_M2, C, c), K0s = ldcf(cs, RCF_K0, C, c), DD1s = ldcf(cs, RCF_DD1, C, c),
                     Es = ldcf(cs, RCF_E12, C, c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xhs = (us.v[j] - q.MUs.v[j]) * q.Rs.v[j];
                const float xhsd = q.Rs.v[j] * (usd.v[j] - M1s.v[j] - xhs * M2s.v[j]);
                os.v[j] = K0s.v[j] * (ds[j] - q.D1s.v[j] - xhs * q.D2s.v[j]) + k.As.v[j] * (dsd[j] - DD1s.v[j] - xhsd * q.D2s.v[j] - xhs * Es.v[j]);
            }
        }
    }
    *(u32x4*)(du3 + off) = pack8(o3);
    *(u32x4*)(dus + off) = pack8(os);
}

// thread = (row group rg, 8-channel chunk ch) walking 2 x 2 CELLS of the padded grid: cell (cy, cx) = padded pixels y in {2 cy - 1,
// 2 cy}, x in {2 cx - 1, 2 cx}, so a cell is either a whole pooling window (1 <= cy <= Ho, 1 <= cx <= Wo) or lies on the rim: border
// pixels (zeros) and interior pixels no window covers (odd H / W: ds = 0).  (One cell per thread re-loaded the 10 / 22 coefficient
// vectors of its chunk for every pixel: 3.2-3.4 TB/s; the reductions, which walk with their coefficients in registers, move 6.)
template <bool TAN>
__global__ __launch_bounds__(256) void rn_join_apply_kernel(RnJoin J, const rbf16* dout, const rbf16* doutd, rbf16* du3, rbf16* dus,
                                                            int ncy, int ncx, unsigned CB) {
    const int b = blockIdx.y, C = J.m.C, nch = C >> 3, nrg = 256 / nch;
    const unsigned nci = (unsigned)ncy * ncx, ncell = (unsigned)J.m.M * nci;
    const int ch = threadIdx.x % nch, rg = threadIdx.x / nch, c = ch * 8;
    if (rg >= nrg) return;
    const float* c3 = J.coef3 + (long)b * RCF_N * C; const float* cs = J.coefs + (long)b * RCF_N * C;
    const JoinCoef k = join_coef(c3, cs, C, c);
    JoinApplyCoef<TAN> q; q.load(c3, cs, C, c);
    const unsigned cbeg = blockIdx.x * CB, cend = min(ncell, cbeg + CB);
    float ds[8], dsd[8];
    const F8 zf = unpack8((u32x4){0u, 0u, 0u, 0u});
    for (unsigned cell = cbeg + rg; cell < cend; cell += nrg) {
        const unsigned img = cell / nci, q0 = cell - img * nci;
        const int cy = (int)(q0 / (unsigned)ncx), cx = (int)(q0 - (unsigned)cy * ncx);
        if (cy >= 1 && cy <= J.Ho && cx >= 1 && cx <= J.Wo) {
            long off[4];
            const long doff = join_offsets(J, b, img, cy - 1, cx - 1, c, off);
            JoinWin w;
            join_window(J, k, off, c, w);
            const F8 dov = ldbf(dout + doff);
            F8 dodv;
            if (TAN) dodv = ldbf(doutd + doff);
#pragma unroll
            for (int p_ = 0; p_ < 4; ++p_) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float mk = w.am[j] == p_ ? w.lm[j] : 0.f;
                    ds[j] = dov.v[j] * mk; dsd[j] = TAN ? dodv.v[j] * mk : 0.f;
                }
                int z = 0;
                if (TAN) asm volatile("s_mov_b32 %0, 0" : "=s"(z));       // (opaque: keeps the per-pixel coefficient loads per pixel)
                join_apply_pixel<TAN>(k, q, c3 + z, cs + z, C, c, unpack8(w.u3[p_]), unpack8(w.us[p_]), TAN ? ldbf(J.u3d + off[p_] + z) : zf,
                                      TAN ? ldbf(J.usd + off[p_] + z) : zf, ds, dsd, du3, dus, off[p_]);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { ds[j] = 0.f; dsd[j] = 0.f; }
        const long imgbase = ((long)b * J.m.M + img) * J.m.g.Pp;
        for (int p_ = 0; p_ < 4; ++p_) {
            const int y = 2 * cy - 1 + (p_ >> 1), x = 2 * cx - 1 + (p_ & 1);
            if (y < 0 || y >= J.m.g.Hp || x < 0 || x >= J.m.g.Wp) continue;
            const long off = (imgbase + (long)y * J.m.g.Wp + x) * C + c;
            if (y < 1 || y > J.m.g.H || x < 1 || x > J.m.g.W) {
                *(u32x4*)(du3 + off) = (u32x4){0u, 0u, 0u, 0u}; *(u32x4*)(dus + off) = (u32x4){0u, 0u, 0u, 0u};
                continue;
            }
            join_apply_pixel<TAN>(k, q, c3, cs, C, c, ldbf(J.u3 + off), ldbf(J.us + off), TAN ? ldbf(J.u3d + off) : zf,
                                  TAN ? ldbf(J.usd + off) : zf, ds, dsd, du3, dus, off);
        }
    }
}

// ---- coefficients from partial sums ----------------------------------------------------------------------------------------------
// first stage for long lists of partials (a conv at 84 x 84 leaves one per 128-pixel tile: 17 000 per episode for 300 query images):
// a workgroup adds PR_GRP consecutive slabs [K][C] in fixed order -> [B][ceil(nt / PR_GRP)][K][C]
constexpr int PR_GRP = 64;
__global__ __launch_bounds__(256) void rn_partial_reduce_kernel(int nt, int KC, const float* part, float* out) {
    __shared__ float red[4][64];
    const int b = blockIdx.z, cl = threadIdx.x & 63, g4 = threadIdx.x >> 6, i = blockIdx.y * 64 + cl;
    const int t0 = blockIdx.x * PR_GRP, t1 = min(nt, t0 + PR_GRP);
    float s = 0.f;
    if (i < KC) {
        const float* base = part + (long)b * nt * KC + i;
        for (int t = t0 + g4; t < t1; t += 4) s += base[(long)t * KC];
    }
    red[g4][cl] = s;
    __syncthreads();
    if (g4 == 0 && i < KC) out[((long)b * gridDim.x + blockIdx.x) * KC + i] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__global__ __launch_bounds__(256) void rn_coef_kernel(RnCoefArgs a) {
    __shared__ double red[4][3][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, g4 = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const bool ok = c < a.C;
    double s[3] = {0.0, 0.0, 0.0};
    const int ks[3] = {a.k0, a.k1, a.k2};
    const int nk = a.mode == RCM_TBWD ? 3 : 2;
    if (ok) {
        const float* base = a.part + (long)b * a.nt * a.K * a.C + c;
        for (int t = g4; t < a.nt; t += 4)
            for (int k = 0; k < nk; ++k) s[k] += (double)base[((long)t * a.K + ks[k]) * a.C];
    }
    for (int k = 0; k < 3; ++k) red[g4][k][cl] = s[k];
    __syncthreads();
    if (g4 || !ok) return;
    for (int k = 0; k < 3; ++k) s[k] = red[0][k][cl] + red[1][k][cl] + red[2][k][cl] + red[3][k][cl];
    float* cf = a.coef + (long)b * RCF_N * a.C + c;
    const double n = a.n;
    if (a.mode == RCM_FWD) {
        const double mu = s[0] / n;
        double var = s[1] / n - mu * mu;
        var = var > 0.0 ? var : 0.0;
        const double r = 1.0 / sqrt(var + (double)RN_EPS);
        const double A = (double)a.g[(long)b * a.pstride + c] * r;
        cf[RCF_MU * a.C] = (float)mu; cf[RCF_R * a.C] = (float)r; cf[RCF_A * a.C] = (float)A;
        cf[RCF_C0 * a.C] = (float)((double)a.beta[(long)b * a.pstride + c] - mu * A);
    } else if (a.mode == RCM_BWD) {
        cf[RCF_D1 * a.C] = (float)(s[0] / n); cf[RCF_D2 * a.C] = (float)(s[1] / n);
        a.dbeta[(long)b * a.gstride + c] = (float)s[0];
        a.dg[(long)b * a.gstride + c] = (float)s[1];
    } else if (a.mode == RCM_TFWD) {
        const double mu = cf[RCF_MU * a.C], r = cf[RCF_R * a.C], A = cf[RCF_A * a.C];
        const double m1 = s[0] / n, m2 = r * (s[1] - mu * s[0]) / n;
        cf[RCF_M1 * a.C] = (float)m1; cf[RCF_M2 * a.C] = (float)m2;
        cf[RCF_TB * a.C] = (float)((double)a.gd[(long)b * a.dstride + c] - A * m2);
        cf[RCF_TC * a.C] = (float)((double)a.betad[(long)b * a.dstride + c] - A * m1);
    } else {
        const double r = cf[RCF_R * a.C], A = cf[RCF_A * a.C], m2 = cf[RCF_M2 * a.C];
        cf[RCF_DD1 * a.C] = (float)(s[0] / n);
        cf[RCF_E12 * a.C] = (float)((s[1] + s[2]) / n);
        cf[RCF_K0 * a.C] = (float)((double)a.gd[(long)b * a.dstride + c] * r - A * r * m2);
        a.dbeta[(long)b * a.gstride + c] = (float)s[0];
        a.dg[(long)b * a.gstride + c] = (float)(s[1] + s[2]);
    }
}

// ---- global average pool ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rn_avgpool_kernel(int C, RnGeom g, const rbf16* o, float* f) {
    const long img = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int y = 1; y <= g.H; ++y)
        for (int x = 1; x <= g.W; ++x) s += rn_bf2f(o[(img * g.Pp + y * g.Wp + x) * C + c]);
    f[img * C + c] = s / (float)(g.H * g.W);
}
__global__ __launch_bounds__(256) void rn_avgpool_bwd_kernel(int C, RnGeom g, const float* df, rbf16* dout) {
    const long img = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)g.Pp * C) return;
    const int q = (int)(i / C), c = (int)(i - (long)q * C);
    const int y = q / g.Wp, x = q - y * g.Wp;
    const bool in = y >= 1 && y <= g.H && x >= 1 && x <= g.W;
    dout[img * g.Pp * C + i] = in ? rn_f2bf(df[img * C + c] / (float)(g.H * g.W)) : (rbf16)0;
}

__global__ __launch_bounds__(256) void rn_img_prep_kernel(long BM, int Cin, RnGeom g, const float* img, rbf16* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;             // one padded pixel
    if (i >= BM * g.Pp) return;
    const long im = i / g.Pp; const int q = (int)(i - im * g.Pp);
    const int y = q / g.Wp, x = q - y * g.Wp;
    F8 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo.v[j] = 0.f; hi.v[j] = 0.f; }
    if (y >= 1 && y <= g.H && x >= 1 && x <= g.W)
        for (int c = 0; c < Cin && c < 8; ++c) lo.v[c] = img[((im * Cin + c) * g.H + (y - 1)) * g.W + (x - 1)];
    *(u32x4*)(out + i * 16) = pack8(lo);
    *(u32x4*)(out + i * 16 + 8) = pack8(hi);
}

}  // namespace
size_t rn_coef_scratch_floats(int B, int nt, int K, int C) { return (size_t)B * ((nt + PR_GRP - 1) / PR_GRP) * K * C; }
namespace {
// pixels per workgroup of the reductions: from PER-EPISODE quantities only, so that an episode's summation order (and with bf16
// storage everything downstream of it) does not depend on how many episodes share the chunk -- chunks and lanes are bit-neutral
inline int red_rb(const RnMap& m) {
    const long npix = (long)m.M * m.g.Pp;
    long rb = npix / 512;
    rb = rb < 64 ? 64 : (rb > 1024 ? 1024 : rb);
    return (int)rb;
}

}  // namespace

int launch_rn_coef(hipStream_t st, const RnCoefArgs& a0, float* scratch) {
    RnCoefArgs a = a0;
    if (a.B < 1 || a.C < 8 || a.nt < 1) return FUMI_EINVAL;
    if (a.nt > 2 * PR_GRP && scratch) {
        const int nt2 = (a.nt + PR_GRP - 1) / PR_GRP, KC = a.K * a.C;
        hipLaunchKernelGGL(rn_partial_reduce_kernel, dim3(nt2, (KC + 63) / 64, a.B), dim3(256), 0, st, a.nt, KC, a.part, scratch);
        LAUNCH_CHECK();
        a.part = scratch; a.nt = nt2;
    }
    hipLaunchKernelGGL(rn_coef_kernel, dim3((a.C + 63) / 64, a.B), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    return FUMI_OK;
}

static inline dim3 unit_grid(const RnMap& m, long npix) { return dim3((unsigned)((npix * (m.C >> 3) + 255) / 256), m.B); }
// pixels per workgroup of the pixel-looped map passes: every thread walks ~8 pixels (per-episode quantities only)
static inline int map_rb(const RnMap& m) { return (256 / (m.C >> 3)) * 8; }

int launch_rn_act(hipStream_t st, const RnMap& m, const rbf16* u, const rbf16* ud, const float* coef, rbf16* out) {
    const long npix = (long)m.M * m.g.Pp;
    if ((m.C >> 3) > 256 || npix >= (1L << 31)) return FUMI_ENOTSUP;
    const int rb = map_rb(m);
    const dim3 grid((unsigned)((npix + rb - 1) / rb), m.B);
    if (ud) hipLaunchKernelGGL(rn_act_kernel<true>, grid, dim3(256), 0, st, m, u, ud, coef, out, rb);
    else hipLaunchKernelGGL(rn_act1_kernel, unit_grid(m, npix), dim3(256), 0, st, m, u, coef, out);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int rn_red_nt(const RnMap& m) {
    const long npix = (long)m.M * m.g.Pp;
    const int rb = red_rb(m);
    return (int)((npix + rb - 1) / rb);
}

int launch_rn_bwd_reduce(hipStream_t st, const RnMap& m, const rbf16* u, const rbf16* ud, const rbf16* da, const rbf16* dad,
                         const float* coef, float* part, int tangent) {
    if ((m.C >> 3) > 256) return FUMI_ENOTSUP;
    const int nrg = 256 / (m.C >> 3), K = tangent ? 3 : 2;
    const size_t lds = (size_t)nrg * K * m.C * 4;
    const dim3 grid(rn_red_nt(m), m.B);
    if (tangent) {
        FUMI_SET_DYN_LDS(rn_bwd_reduce_kernel<true>, lds);
        hipLaunchKernelGGL(rn_bwd_reduce_kernel<true>, grid, dim3(256), lds, st, m, u, ud, da, dad, coef, part, red_rb(m));
    } else {
        FUMI_SET_DYN_LDS(rn_bwd_reduce_kernel<false>, lds);
        hipLaunchKernelGGL(rn_bwd_reduce_kernel<false>, grid, dim3(256), lds, st, m, u, ud, da, dad, coef, part, red_rb(m));
    }
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_bwd_apply(hipStream_t st, const RnMap& m, const rbf16* u, const rbf16* ud, const rbf16* da, const rbf16* dad,
                        const float* coef, rbf16* du, int tangent) {
    const long npix = (long)m.M * m.g.Pp;
    if ((m.C >> 3) > 256 || npix >= (1L << 31)) return FUMI_ENOTSUP;
    const int rb = map_rb(m);
    const dim3 grid((unsigned)((npix + rb - 1) / rb), m.B);
    if (tangent) hipLaunchKernelGGL(rn_bwd_apply_kernel<true>, grid, dim3(256), 0, st, m, u, ud, da, dad, coef, du, rb);
    else hipLaunchKernelGGL(rn_bwd_apply_kernel<false>, grid, dim3(256), 0, st, m, u, ud, da, dad, coef, du, rb);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_join_fwd(hipStream_t st, const RnJoin& j, rbf16* o, int tangent) {
    const long npo = (long)j.m.M * j.gn.Pp;
    if (tangent) hipLaunchKernelGGL(rn_join_fwd_kernel<true>, unit_grid(j.m, npo), dim3(256), 0, st, j, o);
    else hipLaunchKernelGGL(rn_join_fwd_kernel<false>, unit_grid(j.m, npo), dim3(256), 0, st, j, o);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_join_reduce(hipStream_t st, const RnJoin& j, const rbf16* dout, const rbf16* doutd, float* part, int tangent) {
    if ((j.m.C >> 3) > 256) return FUMI_ENOTSUP;
    const int nrg = 256 / (j.m.C >> 3), K = tangent ? 5 : 3;
    const size_t lds = (size_t)nrg * K * j.m.C * 4;
    const int nt = rn_red_nt(j.m);                                    // (the coefficient pass is sized for this many slabs)
    const long nwin = (long)j.m.M * j.Ho * j.Wo;
    if (nwin >= (1L << 31)) return FUMI_ENOTSUP;
    const unsigned wb = (unsigned)((nwin + nt - 1) / nt);             // windows per workgroup
    const dim3 grid(nt, j.m.B);
    if (tangent) {
        FUMI_SET_DYN_LDS(rn_join_reduce_kernel<true>, lds);
        hipLaunchKernelGGL(rn_join_reduce_kernel<true>, grid, dim3(256), lds, st, j, dout, doutd, part, wb);
    } else {
        FUMI_SET_DYN_LDS(rn_join_reduce_kernel<false>, lds);
        hipLaunchKernelGGL(rn_join_reduce_kernel<false>, grid, dim3(256), lds, st, j, dout, doutd, part, wb);
    }
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_join_apply(hipStream_t st, const RnJoin& j, const rbf16* dout, const rbf16* doutd, rbf16* du3, rbf16* dus, int tangent) {
    const int ncy = j.m.g.Hp / 2 + 1, ncx = j.m.g.Wp / 2 + 1;          // cells (2 cy - 1 .. 2 cy) cover padded rows 0 .. Hp - 1
    const long ncell = (long)j.m.M * ncy * ncx;
    if ((j.m.C >> 3) > 256 || ncell * (j.m.C >> 3) >= (1L << 32) - 256) return FUMI_ENOTSUP;
    if (tangent) {                                                       // one cell per thread (see rn_join_apply_cell_kernel)
        hipLaunchKernelGGL(rn_join_apply_cell_kernel<true>, unit_grid(j.m, ncell), dim3(256), 0, st, j, dout, doutd, du3, dus, ncy, ncx);
        LAUNCH_CHECK();
        return FUMI_OK;
    }
    const unsigned cb = (unsigned)(256 / (j.m.C >> 3)) * 4;              // cells per workgroup: every thread walks ~4 cells (16 pixels)
    const dim3 grid((unsigned)((ncell + cb - 1) / cb), j.m.B);
    hipLaunchKernelGGL(rn_join_apply_kernel<false>, grid, dim3(256), 0, st, j, dout, doutd, du3, dus, ncy, ncx, cb);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_avgpool(hipStream_t st, int BM, int C, const RnGeom& g, const rbf16* o, float* f) {
    hipLaunchKernelGGL(rn_avgpool_kernel, dim3((C + 255) / 256, BM), dim3(256), 0, st, C, g, o, f);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_avgpool_bwd(hipStream_t st, int BM, int C, const RnGeom& g, const float* df, rbf16* dout) {
    hipLaunchKernelGGL(rn_avgpool_bwd_kernel, dim3((unsigned)(((long)g.Pp * C + 255) / 256), BM), dim3(256), 0, st, C, g, df, dout);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_rn_img_prep(hipStream_t st, long BM, int Cin, const RnGeom& g, const float* img, rbf16* out) {
    if (Cin < 1 || Cin > 8) return FUMI_EINVAL;
    hipLaunchKernelGGL(rn_img_prep_kernel, dim3((unsigned)((BM * g.Pp + 255) / 256)), dim3(256), 0, st, BM, Cin, g, img, out);
    LAUNCH_CHECK();
    return FUMI_OK;
}
