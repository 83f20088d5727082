// Memory-bound passes of the Conv4 meta-step: batch-statistic BatchNorm (forward, backward, and both tangents of the
// second-order sweep), ReLU and 2x2 max-pool with the arg-max recomputed from the pre-activation (no index tensor is ever
// stored), the logit head with soft-max cross-entropy, and the parameter updates.  Algebra: oracle/conv4_manual.py.
//
// Every pass works on padded channels-last tensors (conv4.h): a thread owns FOUR consecutive channels of its pixels
// (16-byte accesses; 16 lanes cover a pixel's 64 channels = one 256-byte line), per-(episode, channel) coefficients come
// from a small table [B][CF_N][64] that stays in L1.  Sums over an episode's pixels are written as per-workgroup partial
// slabs and added in a fixed order by the coefficient kernel (double accumulation): no float atomics, bit-reproducible.
#include "conv4.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 cf(const float* coef, int b, int field, int c4) { return ld4(coef + ((long)b * CF_N + field) * 64 + 4 * c4); }
__device__ __forceinline__ f32x4 splat(float v) { f32x4 r = {v, v, v, v}; return r; }

// first maximum of the window per channel (PyTorch's max_pool2d scan order: (0,0), (0,1), (1,0), (1,1)); mask = max > 0
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

__device__ __forceinline__ void load_window(const float* u, long pix0, const CvGeom& g, int yo, int xo, int c4, f32x4 (&w)[4]) {
    const float* p = u + (pix0 + (long)(2 * yo + 1) * g.Wp + (2 * xo + 1)) * 64 + 4 * c4;
    w[0] = ld4(p); w[1] = ld4(p + 64); w[2] = ld4(p + (long)g.Wp * 64); w[3] = ld4(p + (long)g.Wp * 64 + 64);
}

// gradient w.r.t. the pooled output of window (yo, xo) of image img, 4 channels
__device__ __forceinline__ f32x4 load_dxo(const float* dxo, const EwGeom& e, long img, int yo, int xo, int c4) {
    if (!e.last) return ld4(dxo + ((long)img * e.gn.Pp + (long)(yo + 1) * e.gn.Wp + (xo + 1)) * 64 + 4 * c4);
    const int hw = e.Ho * e.Wo;
    const float* p = dxo + (long)img * 64 * hw + (long)(4 * c4) * hw + yo * e.Wo + xo;
    f32x4 r = {p[0], p[hw], p[2 * hw], p[3 * hw]};
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// coefficients
// ------------------------------------------------------------------------------------------------------------
// level 1 of the fixed-order sum over an episode's partial slabs: workgroup (chunk, b) adds COEF_CHUNK slabs per channel in
// double (4 row groups x 4 independent loads in flight) -> one double slab per chunk
constexpr int COEF_CHUNK = 64;
__global__ __launch_bounds__(256) void coef_sum_kernel(int nt, int K, const float* part_all, double* out) {
    __shared__ double red[4][3][64];
    const int b = blockIdx.y, ch = blockIdx.x, c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nch = gridDim.x;
    const float* part = part_all + (long)b * nt * K * 64;
    const int t0 = ch * COEF_CHUNK, t1 = min(nt, t0 + COEF_CHUNK);
    double s[3] = {0.0, 0.0, 0.0};
    for (int t = t0 + grp; t < t1; t += 16) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = min(t + 4 * u, t1 - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) v[u][k] = k < K ? part[((long)tt * K + k) * 64 + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t + 4 * u < t1)
#pragma unroll
                for (int k = 0; k < 3; ++k) s[k] += (double)v[u][k];
    }
    for (int k = 0; k < 3; ++k) red[grp][k][c] = s[k];
    __syncthreads();
    if (grp) return;
    for (int k = 0; k < K; ++k)
        out[(((long)b * nch + ch) * 3 + k) * 64 + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
}

__global__ __launch_bounds__(1024) void coef_kernel(CoefArgs a, const double* sums, int nch) {
    __shared__ double red[16][3][64];
    const int b = blockIdx.x, c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    double s[3] = {0.0, 0.0, 0.0};
    for (int t = grp; t < nch; t += 16)
        for (int k = 0; k < a.K; ++k) s[k] += sums[(((long)b * nch + t) * 3 + k) * 64 + c];
    for (int k = 0; k < 3; ++k) red[grp][k][c] = s[k];
    __syncthreads();
    if (grp) return;
    for (int k = 0; k < 3; ++k) { double t = 0.0; for (int g = 0; g < 16; ++g) t += red[g][k][c]; s[k] = t; }
    float* cfp = a.coef + (long)b * CF_N * 64 + c;
    const double n = (double)a.n;
    if (a.mode == CFM_FWD) {
        const double mu = s[0] / n;
        double var = s[1] / n - mu * mu;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)CV_EPS));
        const float g = a.g[(long)b * a.pstride + c], be = a.beta[(long)b * a.pstride + c];
        const float A = g * r;
        cfp[CF_MU * 64] = (float)mu; cfp[CF_R * 64] = r; cfp[CF_A * 64] = A; cfp[CF_C0 * 64] = be - (float)mu * A;
        cfp[CF_GR * 64] = A;
    } else if (a.mode == CFM_BWD) {
        cfp[CF_D1 * 64] = (float)(s[0] / n); cfp[CF_D2 * 64] = (float)(s[1] / n);
        if (a.dg) { a.dg[(long)b * a.gstride + c] = (float)s[1]; a.dbeta[(long)b * a.gstride + c] = (float)s[0]; }
    } else if (a.mode == CFM_TFWD) {
        const float mu = cfp[CF_MU * 64], r = cfp[CF_R * 64], gr = cfp[CF_GR * 64];
        const float m1 = (float)(s[0] / n);
        const float m2 = r * (float)(s[1] / n - (double)mu * (s[0] / n));
        const float gd = a.gd[(long)b * a.dstride + c], bd = a.betad[(long)b * a.dstride + c];
        cfp[CF_M1 * 64] = m1; cfp[CF_M2 * 64] = m2;
        cfp[CF_TA * 64] = gr; cfp[CF_TB * 64] = gd - gr * m2; cfp[CF_TC * 64] = bd - gr * m1;
        cfp[CF_K0 * 64] = gd * r - gr * r * m2;
    } else {
        cfp[CF_DD1 * 64] = (float)(s[0] / n); cfp[CF_E12 * 64] = (float)((s[1] + s[2]) / n);
        if (a.dg) { a.dg[(long)b * a.gstride + c] = (float)(s[1] + s[2]); a.dbeta[(long)b * a.gstride + c] = (float)s[0]; }
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward: x_next = maxpool2(relu(A u + C0)) over the padded output grid (border written as 0) -- or the feature matrix
// [image][c * Ho * Wo + yo * Wo + xo] (PyTorch's flatten order of NCHW) after the last block.
// TAN: also x' = v' at the arg-max (masked), v' = TA u' + TB xh + TC.
// ------------------------------------------------------------------------------------------------------------
template <bool TAN>
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolFwdArgs a, long nthreads) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= nthreads) return;
    const EwGeom& e = a.e;
    const int c4 = (int)(id & 15);
    const long q = id >> 4;
    long img; int yo, xo; bool inside = true; long opix = 0;
    if (e.last) {
        const int hw = e.Ho * e.Wo;
        img = q / hw;
        const int wi = (int)(q - img * hw);
        yo = wi / e.Wo; xo = wi - yo * e.Wo;
    } else {
        img = q / e.gn.Pp;
        const int po = (int)(q - img * e.gn.Pp);
        const int yp = po / e.gn.Wp, xp = po - yp * e.gn.Wp;
        inside = yp >= 1 && yp <= e.Ho && xp >= 1 && xp <= e.Wo;
        yo = yp - 1; xo = xp - 1;
        opix = q;
    }
    f32x4 out = {0.f, 0.f, 0.f, 0.f}, outd = out;
    if (inside) {
        const int b = (int)(img / e.M);
        f32x4 u[4];
        load_window(a.u, img * e.g.Pp, e.g, yo, xo, c4, u);
        const f32x4 A = cf(a.coef, b, CF_A, c4), C0 = cf(a.coef, b, CF_C0, c4);
        f32x4 vmax;
        const ArgMax m = window_argmax(u, A, C0, &vmax);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = vmax[k] > 0.f ? vmax[k] : 0.f;
        if (TAN) {
            f32x4 ud[4];
            load_window(a.ud, img * e.g.Pp, e.g, yo, xo, c4, ud);
            const f32x4 mu = cf(a.coef, b, CF_MU, c4), r = cf(a.coef, b, CF_R, c4);
            const f32x4 TA = cf(a.coef, b, CF_TA, c4), TB = cf(a.coef, b, CF_TB, c4), TC = cf(a.coef, b, CF_TC, c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = m.arg[k];
                const float uu = g == 0 ? u[0][k] : g == 1 ? u[1][k] : g == 2 ? u[2][k] : u[3][k];
                const float dd = g == 0 ? ud[0][k] : g == 1 ? ud[1][k] : g == 2 ? ud[2][k] : ud[3][k];
                outd[k] = m.pos[k] ? TA[k] * dd + TB[k] * ((uu - mu[k]) * r[k]) + TC[k] : 0.f;
            }
        }
    }
    if (e.last) {
        const int hw = e.Ho * e.Wo;
        const long o = img * 64 * hw + (long)(4 * c4) * hw + yo * e.Wo + xo;
        if (!TAN) { a.x[o] = out[0]; a.x[o + hw] = out[1]; a.x[o + 2 * hw] = out[2]; a.x[o + 3 * hw] = out[3]; }
        else { a.xd[o] = outd[0]; a.xd[o + hw] = outd[1]; a.xd[o + 2 * hw] = outd[2]; a.xd[o + 3 * hw] = outd[3]; }
    } else {
        if (!TAN) *(f32x4*)(a.x + opix * 64 + 4 * c4) = out;
        else *(f32x4*)(a.xd + opix * 64 + 4 * c4) = outd;
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward reductions over the pooled windows of one episode.  dv is non-zero only at a window's arg-max:
//   plain:   sum dv = sum dxo mask,  sum dv xh = sum dxo mask xh[arg]
//   tangent: sum dv' = sum dxo' mask, sum dv' xh = ..., sum dv xh' = sum dxo mask xh'[arg],  xh' = r (u' - M1 - xh M2)
// grid (nt, B): workgroup (t, b) takes every nt-th group of 16 windows; 16 window slots x 16 channel quads per workgroup.
// ------------------------------------------------------------------------------------------------------------
template <bool TAN>
__global__ __launch_bounds__(256) void bwd_reduce_kernel(BwdRedArgs a) {
    __shared__ f32x4 red[3][16][16];
    const EwGeom& e = a.e;
    const int b = blockIdx.y, t = blockIdx.x, c4 = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int hw = e.Ho * e.Wo;
    const long nwin = (long)e.M * hw;
    const f32x4 A = cf(a.coef, b, CF_A, c4), C0 = cf(a.coef, b, CF_C0, c4), mu = cf(a.coef, b, CF_MU, c4), r = cf(a.coef, b, CF_R, c4);
    f32x4 M1 = splat(0.f), M2 = M1;
    if (TAN) { M1 = cf(a.coef, b, CF_M1, c4); M2 = cf(a.coef, b, CF_M2, c4); }
    f32x4 s0 = splat(0.f), s1 = s0, s2 = s0;
    for (long w = (long)t * 16 + slot; w < nwin; w += (long)a.nt * 16) {
        const long img = (long)b * e.M + w / hw;
        const int wi = (int)(w % hw), yo = wi / e.Wo, xo = wi - yo * e.Wo;
        f32x4 u[4];
        load_window(a.u, img * e.g.Pp, e.g, yo, xo, c4, u);
        const f32x4 dxo = load_dxo(a.dxo, e, img, yo, xo, c4);
        f32x4 ud[4], dxod;
        if (TAN) { load_window(a.ud, img * e.g.Pp, e.g, yo, xo, c4, ud); dxod = load_dxo(a.dxod, e, img, yo, xo, c4); }
        const ArgMax m = window_argmax(u, A, C0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = m.arg[k];
            const float uu = g == 0 ? u[0][k] : g == 1 ? u[1][k] : g == 2 ? u[2][k] : u[3][k];
            const float xh = (uu - mu[k]) * r[k];
            const float dv = m.pos[k] ? dxo[k] : 0.f;
            if (!TAN) { s0[k] += dv; s1[k] += dv * xh; }
            else {
                const float dd = g == 0 ? ud[0][k] : g == 1 ? ud[1][k] : g == 2 ? ud[2][k] : ud[3][k];
                const float xhd = r[k] * (dd - M1[k] - xh * M2[k]);
                const float dvd = m.pos[k] ? dxod[k] : 0.f;
                s0[k] += dvd; s1[k] += dvd * xh; s2[k] += dv * xhd;
            }
        }
    }
    red[0][slot][c4] = s0; red[1][slot][c4] = s1; red[2][slot][c4] = s2;
    __syncthreads();
    const int K = TAN ? 3 : 2;
    if (threadIdx.x < K * 64) {
        const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
        float s = 0.f;
        for (int sl = 0; sl < 16; ++sl) s += red[k][sl][c >> 2][c & 3];
        a.part[(((long)b * a.nt + t) * K + k) * 64 + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// du = GR (dv - D1 - xh D2) on every interior pixel (pixels outside a pooling window -- odd sizes -- have dv = 0 but a
// non-zero du), 0 on the border.  Units per image: ceil(H/2) * ceil(W/2) aligned 2x2 blocks, then the border pixels.
// TAN: du' = K0 (dv - D1 - xh D2) + GR (dv' - DD1 - xh' D2 - xh E12).
// ------------------------------------------------------------------------------------------------------------
template <bool TAN>
__global__ __launch_bounds__(256) void bwd_apply_kernel(BwdApplyArgs a, long nthreads, int Hb, int Wb, int units) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= nthreads) return;
    const EwGeom& e = a.e;
    const CvGeom& g = e.g;
    const int c4 = (int)(id & 15);
    const long q = id >> 4;
    const long img = q / units;
    const int unit = (int)(q - img * units);
    float* du = a.du + img * g.Pp * 64 + 4 * c4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (unit >= Hb * Wb) {                                            // border pixel
        const int k = unit - Hb * Wb;
        int y, x;
        if (k < g.Wp) { y = 0; x = k; }
        else if (k < 2 * g.Wp) { y = g.Hp - 1; x = k - g.Wp; }
        else { const int kk = k - 2 * g.Wp; y = 1 + (kk >> 1); x = (kk & 1) ? g.Wp - 1 : 0; }
        *(f32x4*)(du + ((long)y * g.Wp + x) * 64) = z4;
        return;
    }
    const int b = (int)(img / e.M);
    const int yb = unit / Wb, xb = unit - yb * Wb;
    const bool vy = 2 * yb + 1 < g.H, vx = 2 * xb + 1 < g.W;          // second row / column of the block exists
    const bool full = yb < e.Ho && xb < e.Wo;                         // the block is a pooling window
    const long p00 = (long)(2 * yb + 1) * g.Wp + (2 * xb + 1);
    const long offs[4] = {p00, p00 + (vx ? 1 : 0), p00 + (vy ? g.Wp : 0), p00 + (vy ? g.Wp : 0) + (vx ? 1 : 0)};
    const bool valid[4] = {true, vx, vy, vx && vy};
    const float* up = a.u + img * g.Pp * 64 + 4 * c4;
    f32x4 u[4], ud[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = ld4(up + offs[k] * 64);
    if (TAN) {
        const float* udp = a.ud + img * g.Pp * 64 + 4 * c4;
#pragma unroll
        for (int k = 0; k < 4; ++k) ud[k] = ld4(udp + offs[k] * 64);
    }
    const f32x4 A = cf(a.coef, b, CF_A, c4), C0 = cf(a.coef, b, CF_C0, c4), mu = cf(a.coef, b, CF_MU, c4), r = cf(a.coef, b, CF_R, c4);
    const f32x4 D1 = cf(a.coef, b, CF_D1, c4), D2 = cf(a.coef, b, CF_D2, c4), GR = cf(a.coef, b, CF_GR, c4);
    f32x4 dxo = z4, dxod = z4;
    ArgMax m;
#pragma unroll
    for (int k = 0; k < 4; ++k) { m.arg[k] = -1; m.pos[k] = false; }
    if (full) {
        dxo = load_dxo(a.dxo, e, img, yb, xb, c4);
        if (TAN) dxod = load_dxo(a.dxod, e, img, yb, xb, c4);
        m = window_argmax(u, A, C0);
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
            const float xh = (u[k][c] - mu[c]) * r[c];
            const bool hit = m.arg[c] == k && m.pos[c];
            const float dv = hit ? dxo[c] : 0.f;
            const float base = dv - D1[c] - xh * D2[c];
            if (!TAN) o[c] = GR[c] * base;
            else {
                const float xhd = r[c] * (ud[k][c] - M1[c] - xh * M2[c]);
                const float dvd = hit ? dxod[c] : 0.f;
                o[c] = K0[c] * base + GR[c] * (dvd - DD1[c] - xhd * D2[c] - xh * E12[c]);
            }
        }
        if (valid[k]) *(f32x4*)(du + offs[k] * 64) = o;
    }
}

// ------------------------------------------------------------------------------------------------------------
// head.  One workgroup per episode: a wave takes a row, its lanes stride over the F features (coalesced) with N partial
// sums each, then a wave reduction.  Plain pass: logits, soft-max, loss, first arg-max, accuracy, dz = (p - y) scale.
// Tangent pass: z' = f' Wh^T + f Wh'^T + bh',  dz' = p (z' - <p, z'>) scale.
// ------------------------------------------------------------------------------------------------------------
constexpr int HEAD_MAXN = 32;
__global__ __launch_bounds__(256) void head_logits_kernel(HeadArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.y * 4 + wave, rstep = gridDim.y * 4;
    const int N = a.N, F = a.F, F1 = F + 1;
    const float* head = a.head + (long)b * N * F1;
    const bool tan = a.fd != nullptr;
    const float* headd = tan ? a.headd + (long)b * N * F1 : nullptr;
    for (int m = row0; m < a.M; m += rstep) {
        const long row = (long)b * a.M + m;
        const float* f = a.f + row * F;
        const float* fd = tan ? a.fd + row * F : nullptr;
        float s[HEAD_MAXN];
#pragma unroll
        for (int n = 0; n < HEAD_MAXN; ++n) s[n] = 0.f;
        for (int k = lane; k < F; k += 64) {
            const float fv = f[k];
            const float fdv = tan ? fd[k] : 0.f;
#pragma unroll
            for (int n = 0; n < HEAD_MAXN; ++n)
                if (n < N) s[n] += tan ? fdv * head[n * F1 + k] + fv * headd[n * F1 + k] : fv * head[n * F1 + k];
        }
#pragma unroll
        for (int n = 0; n < HEAD_MAXN; ++n)
            if (n < N) {
#pragma unroll
                for (int o = 32; o; o >>= 1) s[n] += __shfl_xor(s[n], o);
                s[n] += tan ? headd[n * F1 + F] : head[n * F1 + F];
            }
        // every lane now holds the row's N values
        float* dz = a.dz + row * N;
        if (!tan) {
            float mx = s[0]; int arg = 0;
#pragma unroll
            for (int n = 1; n < HEAD_MAXN; ++n) if (n < N && s[n] > mx) { mx = s[n]; arg = n; }
            float den = 0.f;
#pragma unroll
            for (int n = 0; n < HEAD_MAXN; ++n) if (n < N) den += __expf(s[n] - mx);
            long y = a.y[row];
            if (y < 0 || y >= N) { if (lane == 0 && a.status) atomicOr(a.status, FUMI_ST_LABEL_RANGE); y = 0; }
            const float lse = mx + __logf(den);
            if (lane == 0) {
                float sy = 0.f;
#pragma unroll
                for (int n = 0; n < HEAD_MAXN; ++n) if (n < N) {
                    const float p = __expf(s[n] - lse);
                    a.p[row * N + n] = p;
                    dz[n] = (p - (n == (int)y ? 1.f : 0.f)) * a.scale;
                    if (a.z) a.z[row * N + n] = s[n];
                    if (n == (int)y) sy = s[n];
                }
                if (a.row_loss) { a.row_loss[row] = lse - sy; a.row_hit[row] = arg == (int)y ? 1.f : 0.f; }
                if (a.preds) a.preds[row] = arg;
                if (a.preds_f) a.preds_f[row] = (float)arg;
            }
        } else if (lane == 0) {
            const float* p = a.p + row * N;
            float dot = 0.f;
#pragma unroll
            for (int n = 0; n < HEAD_MAXN; ++n) if (n < N) dot += p[n] * s[n];
#pragma unroll
            for (int n = 0; n < HEAD_MAXN; ++n) if (n < N) dz[n] = p[n] * (s[n] - dot) * a.scale;
        }
    }
}

// loss_b / acc_b of an episode = mean over its rows (fixed order)
__global__ __launch_bounds__(64) void head_finish_kernel(int M, const float* row_loss, const float* row_hit, float* loss_b, float* acc_b) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float l = 0.f, h = 0.f;
    for (int m = lane; m < M; m += 64) { l += row_loss[(long)b * M + m]; h += row_hit[(long)b * M + m]; }
#pragma unroll
    for (int o = 32; o; o >>= 1) { l += __shfl_xor(l, o); h += __shfl_xor(h, o); }
    if (lane == 0) { if (loss_b) loss_b[b] = l / M; if (acc_b) acc_b[b] = h / M; }
}

// thread = feature column k of episode b: dh[n][k] = sum_s sum_m dz_s[m][n] f_s[m][k];  df[m][k] = sum_s sum_n dz_s[m][n] head_s[n][k]
// (column F of dh = colsum of dz_0: the bias).  dz of the episode is staged in LDS.
__global__ __launch_bounds__(256) void head_grad_kernel(HeadGradArgs a) {
    extern __shared__ float dzl[];                       // [nsrc][M][N]
    const int b = blockIdx.y, N = a.N, F = a.F, F1 = F + 1, M = a.M;
    for (int s = 0; s < a.nsrc; ++s)
        for (int i = threadIdx.x; i < M * N; i += 256) dzl[s * M * N + i] = a.dz[s][(long)b * M * N + i];
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > F) return;
    float* dh = a.dh + (long)b * N * F1;
    if (k == F) {
        for (int n = 0; n < N; ++n) { float s = 0.f; for (int m = 0; m < M; ++m) s += dzl[m * N + n]; dh[n * F1 + F] = s; }
        return;
    }
    float acc[HEAD_MAXN];
#pragma unroll
    for (int n = 0; n < HEAD_MAXN; ++n) acc[n] = 0.f;
    for (int s = 0; s < a.nsrc; ++s) {
        const float* f = a.f[s] + (long)b * M * F + k;
        const float* dz = dzl + s * M * N;
        for (int m = 0; m < M; ++m) {
            const float fv = f[(long)m * F];
#pragma unroll
            for (int n = 0; n < HEAD_MAXN; ++n) if (n < N) acc[n] += dz[m * N + n] * fv;
        }
    }
#pragma unroll
    for (int n = 0; n < HEAD_MAXN; ++n) if (n < N) dh[n * F1 + k] = acc[n];
    if (!a.df) return;
    float* df = a.df + (long)b * M * F + k;
    float w[2][HEAD_MAXN];
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < HEAD_MAXN; ++n) w[s][n] = (s < a.nsrc && n < N) ? a.head[s][(long)b * N * F1 + n * F1 + k] : 0.f;
    for (int m = 0; m < M; ++m) {
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < HEAD_MAXN; ++n) if (n < N) {
            v += dzl[m * N + n] * w[0][n];
            if (a.nsrc > 1) v += dzl[M * N + m * N + n] * w[1][n];
        }
        df[(long)m * F] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// small elementwise / layout kernels
// ------------------------------------------------------------------------------------------------------------
__global__ void axpy_kernel(long n4, long n, const float* a, float s, const float* b, float* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { const f32x4 x = ld4(a + 4 * i), y = ld4(b + 4 * i); *(f32x4*)(out + 4 * i) = x + y * s; }
    else { const long j = 4 * n4 + (i - n4); if (j < n) out[j] = a[j] + s * b[j]; }
}

__global__ void reduce_batched_kernel(int ns, long n, const float* part, float scale, float* out, long ostride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float* p = part + (long)b * ns * n + i;
    float s0 = 0.f, s1 = 0.f;
    int s = 0;
    for (; s + 1 < ns; s += 2) { s0 += p[(long)s * n]; s1 += p[(long)(s + 1) * n]; }
    if (s < ns) s0 += p[(long)s * n];
    out[(long)b * ostride + i] = scale * (s0 + s1);
}

__global__ void w1_to_canon_kernel(int n, int Cin, const float* oihw, long istride, float* canon, long ostride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * 2048) return;
    const int w = (int)(i >> 11), e = (int)(i & 2047), co = e >> 5, kap = e & 31;
    canon[(long)w * ostride + e] = kap < Cin * 9 ? oihw[(long)w * istride + co * Cin * 9 + kap] : 0.f;
}
__global__ void w1_from_canon_kernel(int Cin, const float* canon, float* oihw, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * Cin * 9) return;
    const int co = i / (Cin * 9), kap = i - co * Cin * 9;
    oihw[i] = scale * canon[co * 32 + kap];
}
__global__ void broadcast_kernel(long n, const float* src, float* dst, long dstride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[(long)blockIdx.y * dstride + i] = src[i];
}

// dense channels-last [M][H][W][64] <-> padded [M][(H+2)(W+2)][64] (border 0)
__global__ void pad_cl_kernel(long n16, CvGeom g, const float* src, float* dst) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n16) return;
    const int c4 = (int)(id & 15);
    const long q = id >> 4, img = q / g.Pp;
    const int p = (int)(q - img * g.Pp), y = p / g.Wp, x = p - y * g.Wp;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (x >= 1 && x <= g.W && y >= 1 && y <= g.H) v = ld4(src + ((img * g.H + (y - 1)) * g.W + (x - 1)) * 64 + 4 * c4);
    *(f32x4*)(dst + q * 64 + 4 * c4) = v;
}
__global__ void unpad_cl_kernel(long n16, CvGeom g, const float* src, float* dst) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n16) return;
    const int c4 = (int)(id & 15);
    const long q = id >> 4, img = q / (g.H * g.W);
    const int p = (int)(q - img * g.H * g.W), y = p / g.W, x = p - y * g.W;
    *(f32x4*)(dst + q * 64 + 4 * c4) = ld4(src + (img * g.Pp + (long)(y + 1) * g.Wp + (x + 1)) * 64 + 4 * c4);
}

// soft-max cross-entropy of M rows (one thread per row), mean loss by a single workgroup
__global__ __launch_bounds__(256) void ce_kernel(int M, int N, const float* z, const int64_t* y, float* loss, float* dz, int64_t* preds,
                                                 int* status) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int m = threadIdx.x; m < M; m += 256) {
        const float* r = z + (long)m * N;
        float mx = r[0]; int arg = 0;
        for (int n = 1; n < N; ++n) if (r[n] > mx) { mx = r[n]; arg = n; }
        float den = 0.f;
        for (int n = 0; n < N; ++n) den += __expf(r[n] - mx);
        long t = y[m];
        if (t < 0 || t >= N) { atomicOr(status, FUMI_ST_LABEL_RANGE); t = 0; }
        const float lse = mx + __logf(den);
        acc += lse - r[t];
        if (dz) for (int n = 0; n < N; ++n) dz[(long)m * N + n] = (__expf(r[n] - lse) - (n == (int)t ? 1.f : 0.f)) / M;
        if (preds) preds[m] = arg;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) loss[0] = red[0] / M;
}

// per-class mean of the support rows (no scatter atomics: a thread owns one output column and walks the S rows)
__global__ void proto_kernel(int S, int N, int P, const float* x, const int64_t* y, float* out, int* status) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P) return;
    for (int n = 0; n < N; ++n) {
        float s = 0.f; int cnt = 0;
        for (int i = 0; i < S; ++i) {
            const long t = y[(long)b * S + i];
            if (t < 0 || t >= N) { if (c == 0 && n == 0) atomicOr(status, FUMI_ST_LABEL_RANGE); continue; }
            if (t == n) { s += x[((long)b * S + i) * P + c]; ++cnt; }
        }
        out[((long)b * N + n) * P + c] = s / (cnt > 0 ? cnt : 1);
    }
}

}  // namespace

size_t coef_scratch_doubles(int B, int nt) { return (size_t)B * ((nt + COEF_CHUNK - 1) / COEF_CHUNK) * 3 * 64; }

int launch_coef(hipStream_t st, const CoefArgs& a, double* scratch) {
    if (a.K < 1 || a.K > 3 || a.nt < 1 || !scratch) return FUMI_EINVAL;
    const int nch = (a.nt + COEF_CHUNK - 1) / COEF_CHUNK;
    hipLaunchKernelGGL(coef_sum_kernel, dim3(nch, a.B), dim3(256), 0, st, a.nt, a.K, a.part, scratch);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(coef_kernel, dim3(a.B), dim3(1024), 0, st, a, (const double*)scratch, nch);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_pool_fwd(hipStream_t st, const PoolFwdArgs& a, int tangent) {
    const EwGeom& e = a.e;
    const long units = e.last ? (long)e.Ho * e.Wo : (long)e.gn.Pp;
    const long nthreads = (long)e.B * e.M * units * 16;
    const unsigned grid = (unsigned)((nthreads + 255) / 256);
    if (tangent) hipLaunchKernelGGL(pool_fwd_kernel<true>, dim3(grid), dim3(256), 0, st, a, nthreads);
    else hipLaunchKernelGGL(pool_fwd_kernel<false>, dim3(grid), dim3(256), 0, st, a, nthreads);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int ew_bwd_red_nt(const EwGeom& e) {
    const long nwin = (long)e.M * e.Ho * e.Wo;
    long nt = (1024 + e.B - 1) / e.B;
    const long maxt = (nwin + 63) / 64;
    if (nt > maxt) nt = maxt;
    return nt < 1 ? 1 : (int)nt;
}

int launch_bwd_reduce(hipStream_t st, const BwdRedArgs& a, int tangent) {
    if (tangent) hipLaunchKernelGGL(bwd_reduce_kernel<true>, dim3(a.nt, a.e.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(bwd_reduce_kernel<false>, dim3(a.nt, a.e.B), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_bwd_apply(hipStream_t st, const BwdApplyArgs& a, int tangent) {
    const CvGeom& g = a.e.g;
    const int Hb = (g.H + 1) / 2, Wb = (g.W + 1) / 2;
    const int units = Hb * Wb + (g.Pp - g.H * g.W);
    const long nthreads = (long)a.e.B * a.e.M * units * 16;
    const unsigned grid = (unsigned)((nthreads + 255) / 256);
    if (tangent) hipLaunchKernelGGL(bwd_apply_kernel<true>, dim3(grid), dim3(256), 0, st, a, nthreads, Hb, Wb, units);
    else hipLaunchKernelGGL(bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, st, a, nthreads, Hb, Wb, units);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_head_logits(hipStream_t st, const HeadArgs& a) {
    if (a.N < 1 || a.N > HEAD_MAXN) return FUMI_ENOTSUP;
    if ((a.loss_b || a.acc_b) && (!a.row_loss || !a.row_hit)) return FUMI_EINVAL;
    int gy = (a.M + 3) / 4;
    if (gy > 16) gy = 16;
    hipLaunchKernelGGL(head_logits_kernel, dim3(a.B, gy), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    if (!a.fd && (a.loss_b || a.acc_b)) {
        hipLaunchKernelGGL(head_finish_kernel, dim3(a.B), dim3(64), 0, st, a.M, a.row_loss, a.row_hit, a.loss_b, a.acc_b);
        LAUNCH_CHECK();
    }
    return FUMI_OK;
}

int launch_head_grad(hipStream_t st, const HeadGradArgs& a) {
    if (a.N < 1 || a.N > HEAD_MAXN || a.nsrc < 1 || a.nsrc > 2) return FUMI_ENOTSUP;
    const size_t lds = (size_t)a.nsrc * a.M * a.N * 4;
    if (lds > 64 * 1024) return FUMI_ENOTSUP;
    hipLaunchKernelGGL(head_grad_kernel, dim3((a.F + 1 + 255) / 256, a.B), dim3(256), lds, st, a);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_axpy(hipStream_t st, long n, const float* a, float s, const float* b, float* out) {
    const bool al = ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0;
    const long n4 = al ? n / 4 : 0, tot = n4 + (n - 4 * n4);
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n4, n, a, s, b, out);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_reduce_batched(hipStream_t st, int B, int ns, long n, const float* part, float scale, float* out, long ostride) {
    hipLaunchKernelGGL(reduce_batched_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, ns, n, part, scale, out, ostride);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_w1_to_canon(hipStream_t st, int n, int Cin, const float* oihw, long istride, float* canon, long ostride) {
    const long tot = (long)n * 2048;
    hipLaunchKernelGGL(w1_to_canon_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, Cin, oihw, istride, canon, ostride);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_w1_from_canon(hipStream_t st, int Cin, const float* canon, float* oihw, float scale) {
    hipLaunchKernelGGL(w1_from_canon_kernel, dim3((64 * Cin * 9 + 255) / 256), dim3(256), 0, st, Cin, canon, oihw, scale);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_broadcast(hipStream_t st, int B, long n, const float* src, float* dst, long dstride) {
    hipLaunchKernelGGL(broadcast_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, n, src, dst, dstride);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int launch_pad_cl(hipStream_t st, long M, const CvGeom& g, const float* src, float* dst) {
    const long n16 = M * g.Pp * 16;
    hipLaunchKernelGGL(pad_cl_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, n16, g, src, dst);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_unpad_cl(hipStream_t st, long M, const CvGeom& g, const float* src, float* dst) {
    const long n16 = M * g.H * g.W * 16;
    hipLaunchKernelGGL(unpad_cl_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, n16, g, src, dst);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_ce(hipStream_t st, int M, int N, const float* z, const int64_t* y, float* loss, float* dz, int64_t* preds, int* status) {
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, st, M, N, z, y, loss, dz, preds, status);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_proto(hipStream_t st, int B, int S, int N, int P, const float* x, const int64_t* y, float* out, int* status) {
    hipLaunchKernelGGL(proto_kernel, dim3((P + 255) / 256, B), dim3(256), 0, st, S, N, P, x, y, out, status);
    LAUNCH_CHECK();
    return FUMI_OK;
}
