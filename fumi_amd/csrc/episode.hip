// Episode engine: the MAML inner loop of FuMI / MAML with a hand-written forward tape and second-order reverse sweep.
//
// Replaces, for B episodes at once (citations into /root/reference):
//   fumi/models/fumi.py:159-185   im_params / inner loop / query forward / CE / argmax
//   fumi/models/maml.py:162-183   same without the hypernetwork
//   torchmeta gradient_update_parameters (requirements.txt:10)  p <- p - alpha * grad, graph kept
//   fumi/models/fumi.py:190-192   the second-order part of outer_loss.backward()
//
// Algebra (verified against autograd in fp64 by oracle/manual_sweep.py + tests/test_manual_sweep.py):
//   layer 0 is never materialised per episode.  With dz0_t the support pre-activation gradient of layer 0 at step t,
//       W0_t = W0 - alpha * D_t^T Xs,  D_t = sum_{tau<t} dz0_tau [S,h0],  b0_t = b0 - alpha * colsum(D_t)
//       z0_t(X) = A0(X) - alpha * G(X) D_t + b0_t,     A0 = X W0^T,  G = X Xs^T
//   so X is touched only by the shared GEMMs (gemm.hip): forward  A0|G = X [W0;Xs]^T, backward gW0 = Abar0^T X.
//   Everything else is per-episode work on [S|Qn, h] matrices that stay L2-resident:
//       adapt   (1 workgroup / episode)       T inner steps on the support set, tape kept when a gradient is needed
//       query   (1 workgroup / 32 query rows) forward with (theta_T, h_T), CE/argmax, first-order backward, partial slabs
//       reverse (1 workgroup / episode)       sums the slabs, then walks the tape backwards (second order)
//   Small products run on v_mfma_f32_16x16x4_f32 straight from memory (wg_mm in common.h).
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

unsigned long long* g_epi_trace = nullptr;     // dev tracing only
constexpr int QR = 32;                 // query rows per workgroup
constexpr int MAXL = FUMI_MAX_HIDDEN;

struct EpiBuf {
    float *A0, *G;                             // [B,R,h0] [B,R,S], R = S+Qn, support rows first (xpanel.hip)
    float *D, *cs;                             // [B,S,h0] [B,h0]
    float *bcur[MAXL], *bh;                    // i>=1: [B,h_i];  [B,N]
    float *Wslot[MAXL], *Whslot;               // i>=1: [B,nslot,h_i*h_{i-1}];  [B,nslot,N*H]
    float *ta[MAXL], *tdz[MAXL], *tp, *te;     // tape: [B,ntape,S*h_i] ..., [B,ntape,S*N]
    float *lg;                                 // [B,S*N] support logits scratch
    float *aq[MAXL], *zq[MAXL], *lbar, *qcs;   // query: [B,Qn,h_i] x2, [B,Qn,N], [B,ntile,h0]
    float *pW[MAXL], *pb[MAXL], *pWh, *pbh, *pb0, *pD, *ploss, *pcorr;   // per-tile partial slabs
    float *Wb[MAXL], *bb[MAXL], *Whb, *bhb, *b0b, *Db;                   // adjoint state per episode
    float *abar[MAXL], *X0, *X1, *eb, *lb;     // reverse scratch
    float *A0bar;                              // [B,R,h0] adjoint of A0 (support rows: sum over inner steps)
    int nslot, ntape, ntile, maxh;
    unsigned long long* trace;                 // dev: per-phase wall-clock stamps of block 0 (tools/trace_adapt.py)
    int lds_adapt, lds_query, lds_reverse;     // floats of dynamic LDS each kernel stages its products through
};

struct EpiDims {
    int B, N, S, Qn, L, T, H;
    int h[MAXL];
    float alpha;
    int need_grad, second_order, taped;
    // inner-loop dropout (fumi.py:93-99, train mode): keep iff hash(seed, episode, call, layer, element) >= drop_thr,
    // kept activations are scaled by mscale = 1/(1-p); every ReLU-derivative mask carries the same factor
    unsigned drop_thr, seed_lo, seed_hi;
    float mscale;
};

// counter-based dropout mask: the same function is restated in numpy by the tests (tests/helpers.py:dropout_keep)
__device__ __forceinline__ unsigned drop_mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_key(const EpiDims& d, int b, int call, int layer) {
    unsigned h = drop_mix(d.seed_lo + 0x9E3779B9U * (unsigned)b);
    h = drop_mix(h ^ (d.seed_hi + 0x85EBCA6BU * (unsigned)call));
    return drop_mix(h + 0xC2B2AE35U * (unsigned)layer);
}
// relu + dropout of a pre-activation z at flat element index idx of its [rows, h] matrix
__device__ __forceinline__ float drop_relu(const EpiDims& d, unsigned key, long idx, float z) {
    if (z <= 0.f) return 0.f;
    if (d.drop_thr == 0) return z;
    return drop_mix(key ^ (unsigned)idx) >= d.drop_thr ? z * d.mscale : 0.f;
}
__device__ __forceinline__ f32x4 drop_relu4(const EpiDims& d, unsigned key, long idx, const f32x4& z) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = drop_relu(d, key, idx + e, z[e]);
    return o;
}
constexpr int QUERY_CALL = 1 << 20;      // "call" id of the query forward (support step t uses t)

struct EpiParams {                             // meta-parameters of the hidden layers (device pointers)
    const float* W[MAXL];
    const float* b[MAXL];
};

__device__ __forceinline__ int label(const int64_t* y, long i, int N, int* status) {
    long v = y[i];
    if (v < 0 || v >= N) { atomicOr(status, FUMI_ST_LABEL_RANGE); v = 0; }
    return (int)v;
}

// ------------------------------------------------------------------------------------------------------------
// adapt: T inner SGD steps on the support set of one episode
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void adapt_kernel(EpiDims d, EpiBuf w, EpiParams prm, const int64_t* y_s,
                                                     const float* head, int* status) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int sm_cap = w.lds_adapt;
    int stamp_i = 0;
#define STAMP() if (w.trace && threadIdx.x == 0 && blockIdx.x == 0) w.trace[stamp_i++] = __builtin_amdgcn_s_memrealtime();
    STAMP()
    const float* const* Wm = prm.W;
    const float* const* bm = prm.b;
    const float* b0 = prm.b[0];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int S = d.S, N = d.N, L = d.L, H = d.H, h0 = d.h[0];
    const float alpha = d.alpha;
    const int64_t* ys = y_s + (long)b * S;
    float* D = w.D + (long)b * S * h0;
    float* cs = w.cs + (long)b * h0;
    float* bh = w.bh + (long)b * N;
    const float* A0s = w.A0 + (long)b * (S + d.Qn) * h0;
    const float* Gss = w.G + (long)b * (S + d.Qn) * S;
    float* lg = w.lg + (long)b * S * N;

    // ---- initial fast weights: copies of the meta-parameters (slot 0) and of this episode's head
    for (int i = tid; i < S * h0; i += nt) D[i] = 0.f;
    for (int i = 1; i < L; ++i) {
        const long sz = (long)d.h[i] * d.h[i - 1];
        wg_copy(w.Wslot[i] + (long)b * w.nslot * sz, Wm[i], sz);
        float* bd = w.bcur[i] + (long)b * d.h[i];
        for (int j = tid; j < d.h[i]; j += nt) bd[j] = bm[i][j];
    }
    {
        float* dst = w.Whslot + (long)b * w.nslot * N * H;
        const float* hb = head + (long)b * N * (H + 1);
        for (int j = tid; j < N * H; j += nt) dst[j] = hb[(j / H) * (H + 1) + (j % H)];
        for (int j = tid; j < N; j += nt) bh[j] = hb[j * (H + 1) + H];
    }
    __syncthreads(); STAMP()

    for (int t = 0; t < d.T; ++t) {
        const int slot = d.taped ? t : 0, nslot = d.taped ? t + 1 : 0, tp = d.taped ? t : 0;
        // pointers are recomputed from the kernel arguments where needed: local pointer arrays indexed at run time
        // would live in scratch memory
        auto a = [&](int i) { return w.ta[i] + ((long)b * w.ntape + tp) * S * d.h[i]; };
        auto dz = [&](int i) { return w.tdz[i] + ((long)b * w.ntape + tp) * S * d.h[i]; };
        auto Wc = [&](int i) { return (const float*)(w.Wslot[i] + ((long)b * w.nslot + slot) * ((long)d.h[i] * d.h[i - 1])); };
        auto Wn = [&](int i) { return w.Wslot[i] + ((long)b * w.nslot + nslot) * ((long)d.h[i] * d.h[i - 1]); };
        const float* Whc = w.Whslot + ((long)b * w.nslot + slot) * N * H;
        float* Whn = w.Whslot + ((long)b * w.nslot + nslot) * N * H;
        float* p = w.tp + ((long)b * w.ntape + tp) * S * N;
        float* e = w.te + ((long)b * w.ntape + tp) * S * N;

        // 1. layer 0 through the low-rank form (D_0 = 0: the first step is just relu(A0s + b0))
        if (t == 0) {
            const unsigned key0 = drop_key(d, b, t, 0);
            wg_ew((long)S * h0, a(0), A0s, nullptr, nullptr, [&](long i, float x, float, float) {
                return drop_relu(d, key0, i, x + b0[i % h0]);
            });
        } else {
            wg_colsum(sm, sm_cap, S, h0, D, h0, [&](int n, float s) { cs[n] = s; });
            __syncthreads(); STAMP()
            float* a0 = a(0);
            const unsigned key0 = drop_key(d, b, t, 0);
            wg_mm2(sm, sm_cap, S, h0, S, Gss, S, 1, D, h0, 1,
                   [&](int m, int n) { return A0s[(long)m * h0 + n] + (b0[n] - alpha * cs[n]); },
                   [&](int m, int n, float acc, float pre) {
                       a0[(long)m * h0 + n] = drop_relu(d, key0, (long)m * h0 + n, pre - alpha * acc);
                   });
        }
        __syncthreads(); STAMP()
        // 2. deeper layers with the episode's fast weights
        for (int i = 1; i < L; ++i) {
            const int hi = d.h[i], hp = d.h[i - 1];
            const float* bi = w.bcur[i] + (long)b * hi;
            float* ai = a(i);
            const unsigned keyi = drop_key(d, b, t, i);
            wg_mm2(sm, sm_cap, S, hi, hp, a(i - 1), hp, 1, Wc(i), 1, hp, [&](int m, int n) { return bi[n]; },
                   [&](int m, int n, float acc, float pre) {
                       ai[(long)m * hi + n] = drop_relu(d, keyi, (long)m * hi + n, acc + pre);
                   });
            __syncthreads(); STAMP()
        }
        // 3. head logits, softmax, e = (p - onehot)/S
        wg_mm(sm, sm_cap, S, N, H, a(L - 1), H, 1, Whc, 1, H, [&](int m, int n, float acc) { lg[m * N + n] = acc + bh[n]; });
        __syncthreads(); STAMP()
        for (int s = tid; s < S; s += nt) {
            const int y = label(ys, s, N, status);
            float mx = lg[s * N];
            for (int n = 1; n < N; ++n) mx = fmaxf(mx, lg[s * N + n]);
            float sum = 0.f;
            for (int n = 0; n < N; ++n) sum += expf(lg[s * N + n] - mx);
            const float inv = 1.f / sum;
            for (int n = 0; n < N; ++n) {
                const float pv = expf(lg[s * N + n] - mx) * inv;
                p[s * N + n] = pv;
                e[s * N + n] = (pv - (n == y ? 1.f : 0.f)) / (float)S;
            }
        }
        __syncthreads(); STAMP()
        // 4. backward through the head: dz_{L-1} = (e Wh) * relu'   (before Wh may be overwritten in place)
        {
            float* dzl = dz(L - 1); const float* al = a(L - 1);
            wg_mm2(sm, sm_cap, S, H, N, e, N, 1, Whc, H, 1, [&](int m, int n) { return al[(long)m * H + n]; },
                   [&](int m, int n, float acc, float pre) { dzl[(long)m * H + n] = pre > 0.f ? acc * d.mscale : 0.f; });
        }
        __syncthreads(); STAMP()
        // head update: Wh <- Wh - alpha e^T a,  bh <- bh - alpha colsum(e)
        wg_mm2(sm, sm_cap, N, H, S, e, 1, N, a(L - 1), H, 1, [&](int m, int n) { return Whc[m * H + n]; },
               [&](int m, int n, float acc, float pre) { Whn[m * H + n] = pre - alpha * acc; });
        wg_colsum(sm, sm_cap, S, N, e, N, [&](int n, float s) { bh[n] -= alpha * s; });
        // 5. hidden layers, top down
        for (int i = L - 1; i >= 1; --i) {
            const int hi = d.h[i], hp = d.h[i - 1];
            float* dzp = dz(i - 1); const float* ap = a(i - 1);
            wg_mm2(sm, sm_cap, S, hp, hi, dz(i), hi, 1, Wc(i), hp, 1, [&](int m, int n) { return ap[(long)m * hp + n]; },
                   [&](int m, int n, float acc, float pre) { dzp[(long)m * hp + n] = pre > 0.f ? acc * d.mscale : 0.f; });
            __syncthreads(); STAMP()
            const float* Wci = Wc(i); float* Wni = Wn(i);
            wg_mm2(sm, sm_cap, hi, hp, S, dz(i), 1, hi, a(i - 1), hp, 1, [&](int m, int n) { return Wci[(long)m * hp + n]; },
                   [&](int m, int n, float acc, float pre) { Wni[(long)m * hp + n] = pre - alpha * acc; });
            float* bi = w.bcur[i] + (long)b * hi;
            wg_colsum(sm, sm_cap, S, hi, dz(i), hi, [&](int n, float s) { bi[n] -= alpha * s; });
        }
        if (L == 1) __syncthreads();
        // 6. layer 0: only the low-rank factor moves
        {
            wg_ew((long)S * h0, D, D, dz(0), nullptr, [&](long, float x, float y, float) { return x + y; });
        }
        __syncthreads(); STAMP()
    }
    // colsum(D_T): every query tile needs it (b0_T = b0 - alpha * colsum(D_T))
    wg_colsum(sm, sm_cap, S, h0, D, h0, [&](int n, float s) { cs[n] = s; });
}

// ------------------------------------------------------------------------------------------------------------
// query: forward with the adapted weights, loss / argmax, first-order backward of the query loss (partial slabs)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void query_kernel(EpiDims d, EpiBuf w, const float* const b0, const int64_t* y_q,
                                                    float* logits_q, int64_t* preds_q, int* status) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int sm_cap = w.lds_query;
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int S = d.S, N = d.N, L = d.L, H = d.H, h0 = d.h[0], Qn = d.Qn;
    const int r0 = tile * QR, nr = min(QR, Qn - r0);
    const float alpha = d.alpha;
    const int slot = d.taped ? d.T : 0;
    const float* D = w.D + (long)b * S * h0;
    const float* cs = w.cs + (long)b * h0;                 // colsum(D_T), left by adapt
    const float* A0q = w.A0 + ((long)b * (S + Qn) + S + r0) * h0;
    const float* Gqs = w.G + ((long)b * (S + Qn) + S + r0) * S;
    const float* bh = w.bh + (long)b * N;
    const float* Whc = w.Whslot + ((long)b * w.nslot + slot) * N * H;
    float* lq = logits_q + ((long)b * Qn + r0) * N;
    float* lbar = w.lbar + ((long)b * Qn + r0) * N;
    const int64_t* yq = y_q + (long)b * Qn + r0;
    auto a = [&](int i) { return w.aq[i] + ((long)b * Qn + r0) * d.h[i]; };
    auto z = [&](int i) { return w.zq[i] + ((long)b * Qn + r0) * d.h[i]; };
    auto Wc = [&](int i) { return (const float*)(w.Wslot[i] + ((long)b * w.nslot + slot) * ((long)d.h[i] * d.h[i - 1])); };

    {
        float* a0 = a(0);
        const unsigned keyq0 = drop_key(d, b, QUERY_CALL, 0);
        wg_mm2(sm, sm_cap, nr, h0, S, Gqs, S, 1, D, h0, 1,
               [&](int m, int n) { return A0q[(long)m * h0 + n] + (b0[n] - alpha * cs[n]); },
               [&](int m, int n, float acc, float pre) {
                   a0[(long)m * h0 + n] = drop_relu(d, keyq0, (long)(r0 + m) * h0 + n, pre - alpha * acc);
               });
    }
    __syncthreads();
    for (int i = 1; i < L; ++i) {
        const int hi = d.h[i], hp = d.h[i - 1];
        const float* bi = w.bcur[i] + (long)b * hi;
        float* ai = a(i);
        const unsigned keyqi = drop_key(d, b, QUERY_CALL, i);
        wg_mm2(sm, sm_cap, nr, hi, hp, a(i - 1), hp, 1, Wc(i), 1, hp, [&](int m, int n) { return bi[n]; },
               [&](int m, int n, float acc, float pre) {
                   ai[(long)m * hi + n] = drop_relu(d, keyqi, (long)(r0 + m) * hi + n, acc + pre);
               });
        __syncthreads();
    }
    wg_mm(sm, sm_cap, nr, N, H, a(L - 1), H, 1, Whc, 1, H, [&](int m, int n, float acc) { lq[m * N + n] = acc + bh[n]; });
    __syncthreads();

    // per row: log-softmax loss, first arg-max (torch.max semantics, fumi.py:180), lbar = (p - onehot)/Qn
    __shared__ float s_loss[QR];
    __shared__ float s_corr[QR];
    for (int m = tid; m < nr; m += nt) {
        const int y = label(yq, m, N, status);
        float mx = lq[m * N]; int arg = 0;
        for (int n = 1; n < N; ++n) { const float v = lq[m * N + n]; if (v > mx) { mx = v; arg = n; } }
        float sum = 0.f;
        for (int n = 0; n < N; ++n) sum += expf(lq[m * N + n] - mx);
        const float lse = mx + logf(sum), inv = 1.f / sum;
        s_loss[m] = lse - lq[m * N + y];
        s_corr[m] = (arg == y) ? 1.f : 0.f;
        preds_q[(long)b * Qn + r0 + m] = arg;
        for (int n = 0; n < N; ++n)
            lbar[m * N + n] = (expf(lq[m * N + n] - mx) * inv - (n == y ? 1.f : 0.f)) / (float)Qn;
    }
    __syncthreads();
    if (tid == 0) {
        float ls = 0.f, cs_ = 0.f;
        for (int m = 0; m < nr; ++m) { ls += s_loss[m]; cs_ += s_corr[m]; }
        w.ploss[(long)b * w.ntile + tile] = ls;
        w.pcorr[(long)b * w.ntile + tile] = cs_;
    }
    if (!d.need_grad) return;

    // ---- backward of the query loss w.r.t. (theta_T, h_T): partial sums over this tile's rows
    const long pt = (long)b * w.ntile + tile;
    {
        float* pWh = w.pWh + pt * N * H;
        wg_mm(sm, sm_cap, N, H, nr, lbar, 1, N, a(L - 1), H, 1, [&](int m, int n, float acc) { pWh[m * H + n] = acc; });
        float* pbh = w.pbh + pt * N;
        wg_colsum(sm, sm_cap, nr, N, lbar, N, [&](int n, float s) { pbh[n] = s; });
        float* zl = z(L - 1); const float* al = a(L - 1);
        wg_mm2(sm, sm_cap, nr, H, N, lbar, N, 1, Whc, H, 1, [&](int m, int n) { return al[(long)m * H + n]; },
               [&](int m, int n, float acc, float pre) { zl[(long)m * H + n] = pre > 0.f ? acc * d.mscale : 0.f; });
    }
    __syncthreads();
    for (int i = L - 1; i >= 1; --i) {
        const int hi = d.h[i], hp = d.h[i - 1];
        float* pWi = w.pW[i] + pt * (long)hi * hp;
        wg_mm(sm, sm_cap, hi, hp, nr, z(i), 1, hi, a(i - 1), hp, 1, [&](int m, int n, float acc) { pWi[(long)m * hp + n] = acc; });
        float* pbi = w.pb[i] + pt * hi;
        wg_colsum(sm, sm_cap, nr, hi, z(i), hi, [&](int n, float s) { pbi[n] = s; });
        float* zp = z(i - 1); const float* ap = a(i - 1);
        wg_mm2(sm, sm_cap, nr, hp, hi, z(i), hi, 1, Wc(i), hp, 1, [&](int m, int n) { return ap[(long)m * hp + n]; },
               [&](int m, int n, float acc, float pre) { zp[(long)m * hp + n] = pre > 0.f ? acc * d.mscale : 0.f; });
        __syncthreads();
    }
    // layer 0: Abar0 rows of the query set, b0bar, and the adjoint of the low-rank factor D_T
    float* A0bq = w.A0bar + ((long)b * (S + Qn) + S + r0) * h0;
    wg_copy(A0bq, z(0), (long)nr * h0);
    float* pb0 = w.pb0 + pt * h0;
    wg_colsum(sm, sm_cap, nr, h0, z(0), h0, [&](int n, float s) { pb0[n] = s; });
    __syncthreads();
    float* pD = w.pD + pt * (long)S * h0;
    wg_mm2(sm, sm_cap, S, h0, nr, Gqs, 1, S, z(0), h0, 1, [&](int m, int n) { return pb0[n]; },
           [&](int m, int n, float acc, float pre) { pD[(long)m * h0 + n] = -alpha * (acc + pre); });
}

// ------------------------------------------------------------------------------------------------------------
// query, LDS-resident form: the same arithmetic as query_kernel, but the tile's whole phase chain lives in LDS
// (common.h "LDS-resident products").  Staged once: the tile's A0 rows (into a_0), its G rows, D_T, the adapted fast
// weights of every deeper layer, the head, the bias vectors.  z_i overwrites a_i in place (same-position dependence
// only), so the footprint is  QR*sum ld(h_i) + sum h_i*ld(h_{i-1}) + S*ld(h0) + small.  Taken when that fits
// (query_layout().total <= QLDS_CAP), i.e. for the reference's [256, 64] image network and anything smaller.
// ------------------------------------------------------------------------------------------------------------
constexpr int QLDS_CAP = 40000;         // floats of dynamic LDS (160 KiB = 40960 less the kernel's static arrays)
struct QLay { int a[MAXL], W[MAXL], bi[MAXL], Wh, Gq, D, lq, b0, cs, pb0, bh, total; };
__host__ __device__ inline int q_r4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int q_r16(int x) { return (x + 15) & ~15; }
__host__ __device__ inline void query_layout(QLay& y, int L, const int* h, int S, int N) {
    int off = 0;
    for (int i = 0; i < MAXL; ++i) { y.a[i] = y.W[i] = y.bi[i] = 0; }
    for (int i = 0; i < L; ++i) { y.a[i] = off; off += QR * wg_ld(h[i]); }
    for (int i = 1; i < L; ++i) { y.W[i] = off; off += q_r16(h[i]) * wg_ld(h[i - 1]); y.bi[i] = off; off += q_r4(h[i]); }
    y.Wh = off; off += q_r16(N) * wg_ld(h[L - 1]);
    y.Gq = off; off += QR * wg_ld(S);
    y.D = off; off += q_r4(S) * wg_ld(h[0]);
    y.lq = off; off += QR * wg_ld(N);
    y.b0 = off; off += q_r4(h[0]);
    y.cs = off; off += q_r4(h[0]);
    y.pb0 = off; off += q_r4(h[0]);
    y.bh = off; off += q_r4(N);
    y.total = off;
}

__global__ __launch_bounds__(512) void query_lds_kernel(StageTab stg, EpiDims d, EpiBuf w, QLay y, const int64_t* y_q,
                                                        float* logits_q, int64_t* preds_q, int* status) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float s_loss[QR];
    __shared__ float s_corr[QR];
    __shared__ StageTab s_stg;
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    int qi = 0;
#define QSTAMP() if (w.trace && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) w.trace[64 + qi++] = __builtin_amdgcn_s_memrealtime();
    QSTAMP()
    const int S = d.S, N = d.N, L = d.L, H = d.H, h0 = d.h[0], Qn = d.Qn;
    const int r0 = tile * QR, nr = min(QR, Qn - r0);
    const float alpha = d.alpha;
    const int ldq = wg_ld(N), ldG = wg_ld(S), ld0 = wg_ld(h0), ldH = wg_ld(H);
    auto a = [&](int i) { return sm + y.a[i]; };
    auto lda = [&](int i) { return wg_ld(d.h[i]); };

    // ---- zero the arena (padding must read as zero), then stage everything this tile needs in one batch (plan: host)
    wg_stage_tab_to_lds(&s_stg);
    for (int i = tid * 4, tot = y.total; i < tot; i += nt * 4) *(f32x4*)(sm + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads(); QSTAMP()
    wg_stage_rows<12>(&s_stg, b, tile, nr, sm);
    wg_lds_barrier(); QSTAMP()

    // ---- forward (epilogues handle 4 consecutive columns of one row: wg_lmm)
    {
        float* a0 = a(0);
        const float* b0l = sm + y.b0; const float* csl = sm + y.cs;
        const unsigned keyq0 = drop_key(d, b, QUERY_CALL, 0);
        wg_lmm_wide<true>(nr, h0, S, sm + y.Gq, ldG, sm + y.D, ld0, [&](int m, int n, const f32x4& acc, int) {
            float* p = a0 + m * ld0 + n;
            const f32x4 pre = *(const f32x4*)p + (*(const f32x4*)(b0l + n) - alpha * *(const f32x4*)(csl + n));
            *(f32x4*)p = drop_relu4(d, keyq0, (long)(r0 + m) * h0 + n, pre - alpha * acc);
        });
    }
    wg_lds_barrier(); QSTAMP()
    for (int i = 1; i < L; ++i) {
        const int hi = d.h[i], hp = d.h[i - 1];
        float* ai = a(i); const int ldi = lda(i);
        const float* bi = sm + y.bi[i];
        const unsigned keyqi = drop_key(d, b, QUERY_CALL, i);
        wg_lmm<true, true>(nr, hi, hp, a(i - 1), lda(i - 1), sm + y.W[i], wg_ld(hp), [&](int m, int n, const f32x4& acc, int) {
            *(f32x4*)(ai + m * ldi + n) = drop_relu4(d, keyqi, (long)(r0 + m) * hi + n, acc + *(const f32x4*)(bi + n));
        });
        wg_lds_barrier(); QSTAMP()
    }
    float* lq = sm + y.lq;
    {
        float* lg = logits_q + ((long)b * Qn + r0) * N;
        const float* bhl = sm + y.bh;
        wg_lmm<true, true>(nr, N, H, a(L - 1), ldH, sm + y.Wh, ldH, [&](int m, int n, const f32x4& acc, int cnt) {
            const f32x4 v = acc + *(const f32x4*)(bhl + n);        // bh is zero-padded: columns past N stay 0
            *(f32x4*)(lq + m * ldq + n) = v;
            wg_st4(lg + m * N + n, v, cnt);
        });
    }
    wg_lds_barrier(); QSTAMP()
    // per row: log-softmax loss, first arg-max (torch.max semantics, fumi.py:180), lbar = (p - onehot)/Qn in place
    for (int m = tid; m < nr; m += nt) {
        const int yy = label(y_q + (long)b * Qn + r0, m, N, status);
        float* row = lq + m * ldq;
        float mx = row[0]; int arg = 0;
        for (int n = 1; n < N; ++n) { const float v = row[n]; if (v > mx) { mx = v; arg = n; } }
        float sum = 0.f;
        for (int n = 0; n < N; ++n) sum += expf(row[n] - mx);
        const float lse = mx + logf(sum), inv = 1.f / sum;
        s_loss[m] = lse - row[yy];
        s_corr[m] = (arg == yy) ? 1.f : 0.f;
        preds_q[(long)b * Qn + r0 + m] = arg;
        for (int n = 0; n < N; ++n) row[n] = (expf(row[n] - mx) * inv - (n == yy ? 1.f : 0.f)) / (float)Qn;
    }
    wg_lds_barrier(); QSTAMP()
    if (tid == 0) {
        float ls = 0.f, cs_ = 0.f;
        for (int m = 0; m < nr; ++m) { ls += s_loss[m]; cs_ += s_corr[m]; }
        w.ploss[(long)b * w.ntile + tile] = ls;
        w.pcorr[(long)b * w.ntile + tile] = cs_;
    }
    if (!d.need_grad) return;

    // ---- backward of the query loss w.r.t. (theta_T, h_T): partial sums over this tile's rows
    const long pt = (long)b * w.ntile + tile;
    const float* lbar = lq;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto relu_bwd4 = [&](const f32x4& act, const f32x4& g) {      // g * relu'(z) with the dropout scale folded in
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act[e] > 0.f ? g[e] * d.mscale : 0.f;
        return o;
    };
    {
        float* pWh = w.pWh + pt * N * H;
        wg_lmm_wide<false>(N, H, nr, lbar, ldq, a(L - 1), ldH, [&](int m, int n, const f32x4& acc, int cnt) {
            wg_st4(pWh + m * H + n, acc, cnt);
        });
        float* pbh = w.pbh + pt * N;
        wg_lcolsum(nr, N, lbar, ldq, [&](int n, float s) { pbh[n] = s; });
    }
    wg_lds_barrier(); QSTAMP()
    {
        float* zl = a(L - 1);
        wg_lmm_wide<true>(nr, H, N, lbar, ldq, sm + y.Wh, ldH, [&](int m, int n, const f32x4& acc, int) {
            float* p = zl + m * ldH + n;
            *(f32x4*)p = relu_bwd4(*(const f32x4*)p, acc);
        });
    }
    wg_lds_barrier(); QSTAMP()
    for (int i = L - 1; i >= 1; --i) {
        const int hi = d.h[i], hp = d.h[i - 1];
        const float* zi = a(i); const int ldi = lda(i), ldp = lda(i - 1);
        float* pWi = w.pW[i] + pt * (long)hi * hp;
        wg_lmm_wide<false>(hi, hp, nr, zi, ldi, a(i - 1), ldp, [&](int m, int n, const f32x4& acc, int cnt) {
            wg_st4(pWi + (long)m * hp + n, acc, cnt);
        });
        float* pbi = w.pb[i] + pt * hi;
        wg_lcolsum(nr, hi, zi, ldi, [&](int n, float s) { pbi[n] = s; });
        wg_lds_barrier(); QSTAMP()
        float* zp = a(i - 1);
        wg_lmm_wide<true>(nr, hp, hi, zi, ldi, sm + y.W[i], wg_ld(hp), [&](int m, int n, const f32x4& acc, int) {
            float* p = zp + m * ldp + n;
            *(f32x4*)p = relu_bwd4(*(const f32x4*)p, acc);
        });
        wg_lds_barrier(); QSTAMP()
    }
    // layer 0: Abar0 rows of the query set, b0bar, and the adjoint of the low-rank factor D_T
    const float* z0 = a(0);
    float* A0bq = w.A0bar + ((long)b * (S + Qn) + S + r0) * h0;
    {
        const int c4n = (h0 + 3) >> 2;
        for (int i = tid; i < nr * c4n; i += nt) {
            const int m = i / c4n, n = (i - m * c4n) << 2;
            wg_st4(A0bq + (long)m * h0 + n, *(const f32x4*)(z0 + m * ld0 + n), min(4, h0 - n));
        }
    }
    float* pb0 = w.pb0 + pt * h0; float* pb0l = sm + y.pb0;
    wg_lcolsum(nr, h0, z0, ld0, [&](int n, float s) { pb0[n] = s; pb0l[n] = s; });
    wg_lds_barrier(); QSTAMP()
    float* pD = w.pD + pt * (long)S * h0;
    wg_lmm_wide<false>(S, h0, nr, sm + y.Gq, ldG, z0, ld0, [&](int m, int n, const f32x4& acc, int cnt) {
        wg_st4(pD + (long)m * h0 + n, -alpha * (acc + *(const f32x4*)(pb0l + n)), cnt);
    });
    (void)z4;
    QSTAMP()
#undef QSTAMP
}

// ------------------------------------------------------------------------------------------------------------
// reverse: sum the query slabs, then the second-order sweep back through the T inner steps
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void reverse_kernel(EpiDims d, EpiBuf w, float* loss_b, float* acc_b, float* head_bar) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int sm_cap = w.lds_reverse;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int S = d.S, N = d.N, L = d.L, H = d.H, h0 = d.h[0];
    const float alpha = d.alpha;
    const int ntile = w.ntile;

    if (tid == 0) {
        float ls = 0.f, cr = 0.f;
        for (int t = 0; t < ntile; ++t) { ls += w.ploss[(long)b * ntile + t]; cr += w.pcorr[(long)b * ntile + t]; }
        loss_b[b] = ls / (float)d.Qn;
        acc_b[b] = cr / (float)d.Qn;
    }
    if (!d.need_grad) return;

    auto Wb = [&](int i) { return w.Wb[i] + (long)b * ((long)d.h[i] * d.h[i - 1]); };
    auto bb = [&](int i) { return w.bb[i] + (long)b * d.h[i]; };
    float* Whb = w.Whb + (long)b * N * H;
    float* bhb = w.bhb + (long)b * N;
    float* b0b = w.b0b + (long)b * h0;
    float* Db = w.Db + (long)b * S * h0;
    float* A0bs = w.A0bar + (long)b * (S + d.Qn) * h0;
    auto sum_tiles = [&](float* dst, const float* src, long sz) {
        wg_sum_slabs(dst, src + (long)b * ntile * sz, ntile, sz, sz);
    };
    for (int i = 1; i < L; ++i) {
        const long sz = (long)d.h[i] * d.h[i - 1];
        sum_tiles(Wb(i), w.pW[i], sz);
        sum_tiles(bb(i), w.pb[i], d.h[i]);
    }
    sum_tiles(Whb, w.pWh, (long)N * H);
    sum_tiles(bhb, w.pbh, N);
    sum_tiles(b0b, w.pb0, h0);
    sum_tiles(Db, w.pD, (long)S * h0);
    for (int i = tid; i < S * h0; i += nt) A0bs[i] = 0.f;
    __syncthreads();

    if (d.second_order) {
        const float* Gss = w.G + (long)b * (S + d.Qn) * S;
        float* cs = w.cs + (long)b * h0;
        float* eb = w.eb + (long)b * S * N;
        float* lb = w.lb + (long)b * S * N;
        float* const Xa = w.X0 + (long)b * S * w.maxh;
        float* const Xb = w.X1 + (long)b * S * w.maxh;
        auto X = [&](int i) { return i ? Xb : Xa; };
        for (int t = d.T - 1; t >= 0; --t) {
            auto a = [&](int i) { return (const float*)(w.ta[i] + ((long)b * w.ntape + t) * S * d.h[i]); };
            auto dz = [&](int i) { return (const float*)(w.tdz[i] + ((long)b * w.ntape + t) * S * d.h[i]); };
            auto ab = [&](int i) { return w.abar[i] + (long)b * S * d.h[i]; };
            auto Wc = [&](int i) { return (const float*)(w.Wslot[i] + ((long)b * w.nslot + t) * ((long)d.h[i] * d.h[i - 1])); };
            const float* Whc = w.Whslot + ((long)b * w.nslot + t) * N * H;
            const float* p = w.tp + ((long)b * w.ntape + t) * S * N;
            const float* e = w.te + ((long)b * w.ntape + t) * S * N;

            // abar_i = 0 ; dab = Dbar * relu'(z0)          (dz0bar = adjoint of D_{t+1})
            for (int i = 0; i < L; ++i) {
                float* abi = ab(i);
                for (int j = tid; j < S * d.h[i]; j += nt) abi[j] = 0.f;
            }
            int cur = 0;
            {
                wg_ew((long)S * h0, Xa, a(0), Db, nullptr, [&](long, float m_, float v, float) { return m_ > 0.f ? v * d.mscale : 0.f; });
            }
            __syncthreads();
            // ---- reverse of the backward pass, bottom up
            for (int i = 1; i < L; ++i) {
                const int hi = d.h[i], hp = d.h[i - 1];
                const float* dab = X(cur); float* nxt = X(cur ^ 1);
                const float* bbi = bb(i); const float* Wbi = Wb(i); const float* ai = a(i);
                // dzbar_i = dab W_i^T - alpha bbar_i - alpha a_{i-1} Wbar_i^T ; next dab = dzbar_i * relu'(z_i)
                wg_mm2(sm, sm_cap, S, hi, hp, dab, hp, 1, Wc(i), 1, hp, [&](int m, int n) { return bbi[n]; },
                       [&](int m, int n, float acc, float pre) { nxt[(long)m * hi + n] = acc - alpha * pre; });
                wg_mm2(sm, sm_cap, S, hi, hp, a(i - 1), hp, 1, Wbi, 1, hp,                                   // same thread, same (m,n)
                       [&](int m, int n) { return f32pair{nxt[(long)m * hi + n], ai[(long)m * hi + n]}; },
                       [&](int m, int n, float acc, f32pair pre) {
                           nxt[(long)m * hi + n] = pre.y > 0.f ? (pre.x - alpha * acc) * d.mscale : 0.f;
                       });
                // abar_{i-1} += dz_i (-alpha Wbar_i)
                float* abp = ab(i - 1);
                wg_mm2(sm, sm_cap, S, hp, hi, dz(i), hi, 1, Wbi, hp, 1, [&](int m, int n) { return abp[(long)m * hp + n]; },
                       [&](int m, int n, float acc, float pre) { abp[(long)m * hp + n] = pre - alpha * acc; });
                __syncthreads();
                // Wbar_i += dz_i^T dab
                float* Wbw = Wb(i);
                wg_mm2(sm, sm_cap, hi, hp, S, dz(i), 1, hi, dab, hp, 1, [&](int m, int n) { return Wbw[(long)m * hp + n]; },
                       [&](int m, int n, float acc, float pre) { Wbw[(long)m * hp + n] = pre + acc; });
                __syncthreads();
                cur ^= 1;
            }
            {   // head: ebar = dab Wh^T - alpha bhbar - alpha a Whbar^T ; abar += e (-alpha Whbar) ; Whbar += e^T dab
                const float* dab = X(cur);
                wg_mm2(sm, sm_cap, S, N, H, dab, H, 1, Whc, 1, H, [&](int m, int n) { return bhb[n]; },
                       [&](int m, int n, float acc, float pre) { eb[m * N + n] = acc - alpha * pre; });
                wg_mm2(sm, sm_cap, S, N, H, a(L - 1), H, 1, Whb, 1, H, [&](int m, int n) { return eb[m * N + n]; },
                       [&](int m, int n, float acc, float pre) { eb[m * N + n] = pre - alpha * acc; });
                float* abl = ab(L - 1);
                wg_mm2(sm, sm_cap, S, H, N, e, N, 1, Whb, H, 1, [&](int m, int n) { return abl[(long)m * H + n]; },
                       [&](int m, int n, float acc, float pre) { abl[(long)m * H + n] = pre - alpha * acc; });
                __syncthreads();
                wg_mm2(sm, sm_cap, N, H, S, e, 1, N, dab, H, 1, [&](int m, int n) { return Whb[m * H + n]; },
                       [&](int m, int n, float acc, float pre) { Whb[m * H + n] = pre + acc; });
                // softmax-CE second derivative: lbar = p * (pbar - <p,pbar>),  pbar = ebar / S
                for (int s = tid; s < S; s += nt) {
                    float dot = 0.f;
                    for (int n = 0; n < N; ++n) dot += p[s * N + n] * eb[s * N + n];
                    for (int n = 0; n < N; ++n) lb[s * N + n] = p[s * N + n] * (eb[s * N + n] - dot) / (float)S;
                }
                __syncthreads();
                // ---- reverse of the forward pass
                wg_mm2(sm, sm_cap, S, H, N, lb, N, 1, Whc, H, 1, [&](int m, int n) { return abl[(long)m * H + n]; },
                       [&](int m, int n, float acc, float pre) { abl[(long)m * H + n] = pre + acc; });
                wg_mm2(sm, sm_cap, N, H, S, lb, 1, N, a(L - 1), H, 1, [&](int m, int n) { return Whb[m * H + n]; },
                       [&](int m, int n, float acc, float pre) { Whb[m * H + n] = pre + acc; });
                wg_colsum(sm, sm_cap, S, N, lb, N, [&](int n, float s) { bhb[n] += s; });
                __syncthreads();
            }
            for (int i = L - 1; i >= 1; --i) {
                const int hi = d.h[i], hp = d.h[i - 1];
                float* zb = ab(i);
                {
                    wg_ew((long)S * hi, zb, a(i), zb, nullptr, [&](long, float m_, float v, float) { return m_ > 0.f ? v * d.mscale : 0.f; });
                }
                __syncthreads();
                float* abp = ab(i - 1); float* Wbw = Wb(i); float* bbw = bb(i);
                wg_mm2(sm, sm_cap, S, hp, hi, zb, hi, 1, Wc(i), hp, 1, [&](int m, int n) { return abp[(long)m * hp + n]; },
                       [&](int m, int n, float acc, float pre) { abp[(long)m * hp + n] = pre + acc; });
                wg_mm2(sm, sm_cap, hi, hp, S, zb, 1, hi, a(i - 1), hp, 1, [&](int m, int n) { return Wbw[(long)m * hp + n]; },
                       [&](int m, int n, float acc, float pre) { Wbw[(long)m * hp + n] = pre + acc; });
                wg_colsum(sm, sm_cap, S, hi, zb, hi, [&](int n, float s) { bbw[n] += s; });
                __syncthreads();
            }
            // layer 0: z0bar -> Abar0 rows of the support set, b0bar, Dbar
            float* z0b = ab(0);
            wg_ew((long)S * h0, z0b, a(0), z0b, nullptr, [&](long, float m_, float v, float) { return m_ > 0.f ? v * d.mscale : 0.f; });
            __syncthreads();
            wg_ew((long)S * h0, A0bs, A0bs, z0b, nullptr, [&](long, float x, float y, float) { return x + y; });
            __syncthreads();
            wg_colsum(sm, sm_cap, S, h0, z0b, h0, [&](int n, float s) { cs[n] = s; b0b[n] += s; });
            __syncthreads();
            wg_mm2(sm, sm_cap, S, h0, S, Gss, S, 1, z0b, h0, 1,
                   [&](int m, int n) { return f32pair{Db[(long)m * h0 + n], cs[n]}; },
                   [&](int m, int n, float acc, f32pair pre) { Db[(long)m * h0 + n] = pre.x - alpha * (acc + pre.y); });
            __syncthreads();
        }
    }
    // d loss_b / d head_b = [Whbar | bhbar]
    float* hb = head_bar + (long)b * N * (H + 1);
    for (int j = tid; j < N * H; j += nt) hb[(j / H) * (H + 1) + (j % H)] = Whb[j];
    for (int j = tid; j < N; j += nt) hb[j * (H + 1) + H] = bhb[j];
}

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
__global__ void class_text_select_kernel(int N, int S, int Dt, const float* text_s, const int64_t* y_s, float* out,
                                         int* status) {
    // one wave per (episode, class): ballot for the FIRST support row of the class (fumi.py:207-210), then row copy
    const int b = blockIdx.y, n = blockIdx.x, lane = threadIdx.x;
    const int64_t* ys = y_s + (long)b * S;
    int first = S;
    for (int s0 = 0; s0 < S && first == S; s0 += 64) {
        const int s = s0 + lane;
        const bool hit = s < S && ys[s] == n;
        const unsigned long long m = __ballot(hit);
        if (m) first = s0 + __ffsll((long long)m) - 1;
    }
    float* o = out + ((long)b * N + n) * Dt;
    if (first == S) {
        if (lane == 0) atomicOr(status, FUMI_ST_CLASS_MISSING);
        for (int j = lane; j < Dt; j += 64) o[j] = __builtin_nanf("");
        return;
    }
    const float* src = text_s + ((long)b * S + first) * Dt;
    for (int j = lane; j < Dt; j += 64) o[j] = src[j];
}

__global__ void broadcast_head_kernel(int N, int H, const float* Wf, const float* bf, float* head) {
    float* hb = head + (long)blockIdx.x * N * (H + 1);
    for (int j = threadIdx.x; j < N * (H + 1); j += blockDim.x) {
        const int n = j / (H + 1), c = j % (H + 1);
        hb[j] = c < H ? Wf[n * H + c] : bf[n];
    }
}

__global__ void tanh_bwd_kernel(long n, const float* h, const float* hbar, float* out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = hbar[i] * (1.f - h[i] * h[i]);
}

__global__ void relu_mask_mul_kernel(long n, const float* u, float* g, float scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        g[i] = u[i] > 0.f ? g[i] * scale : 0.f;
}

__global__ void split_head_grad_kernel(int B, int N, int H, const float* head_bar, float scale, float* gW, float* gb) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N * (H + 1)) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += head_bar[(long)b * N * (H + 1) + j];
    const int n = j / (H + 1), c = j % (H + 1);
    if (c < H) gW[n * H + c] = scale * s; else gb[n] = scale * s;
}

inline int blocks_for(long n) { long b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

struct Carver {      // computes the layout twice: once to size the workspace, once to hand out pointers
    fumi_ws* ws; size_t bytes;
    float* take(size_t n) {
        bytes += ws_align(n * sizeof(float));
        return ws ? ws_f(ws, n) : nullptr;
    }
};

void carve(Carver& c, const EpisodeProblem& p, EpiBuf& w) {
    const size_t B = p.B, S = p.S, Qn = p.Qn, N = p.N, L = p.L, h0 = p.h[0], H = p.h[L - 1];
    const bool taped = p.need_grad && p.second_order && p.T > 0;
    w.nslot = taped ? p.T + 1 : 1;
    w.ntape = taped ? p.T : 1;
    w.ntile = (p.Qn + QR - 1) / QR;
    int maxh = 0;
    for (int i = 0; i < p.L; ++i) maxh = p.h[i] > maxh ? p.h[i] : maxh;
    w.maxh = maxh;
    {   // LDS for wg_mm's operand images: enough for the largest product of each kernel in one chunk, capped
        auto r16 = [](int x) { return (x + 15) & ~15; };
        auto need = [&](int M, int N_, int K) { return (long)(r16(M) + 4 + r16(N_) + 4) * (r16(K) + 4); };
        const int S_ = p.S, N_ = p.N, H_ = p.h[p.L - 1], h0_ = p.h[0], R_ = QR;
        long a = need(S_, h0_, S_), q = need(R_, h0_, S_), r = need(S_, h0_, S_);
        a = std::max({a, need(S_, N_, H_), need(S_, H_, N_), need(N_, H_, S_)});
        q = std::max({q, need(R_, N_, H_), need(N_, H_, R_), need(R_, H_, N_), need(S_, h0_, R_)});
        r = std::max({r, need(S_, N_, H_), need(S_, H_, N_), need(N_, H_, S_)});
        for (int i = 1; i < p.L; ++i) {
            const int hi = p.h[i], hp = p.h[i - 1];
            a = std::max({a, need(S_, hi, hp), need(S_, hp, hi), need(hi, hp, S_)});
            q = std::max({q, need(R_, hi, hp), need(R_, hp, hi), need(hi, hp, R_)});
            r = std::max({r, need(S_, hi, hp), need(S_, hp, hi), need(hi, hp, S_)});
        }
        const long cap = 38000;               // 152 KB of the CU's 160 KB
        w.lds_adapt = (int)std::min(a, cap); w.lds_query = (int)std::min(q, cap); w.lds_reverse = (int)std::min(r, cap);
    }
    const size_t nt = w.ntile;
    w.A0 = c.take(B * (S + Qn) * h0); w.G = c.take(B * (S + Qn) * S);
    w.D = c.take(B * S * h0); w.cs = c.take(B * h0);
    w.bh = c.take(B * N); w.Whslot = c.take(B * w.nslot * N * H);
    w.tp = c.take(B * w.ntape * S * N); w.te = c.take(B * w.ntape * S * N);
    w.lg = c.take(B * S * N);
    w.lbar = c.take(B * Qn * N); w.qcs = c.take(B * nt * h0);
    w.ploss = c.take(B * nt); w.pcorr = c.take(B * nt);
    for (size_t i = 0; i < L; ++i) {
        const size_t hi = p.h[i], hp = i ? p.h[i - 1] : 0;
        w.ta[i] = c.take(B * w.ntape * S * hi); w.tdz[i] = c.take(B * w.ntape * S * hi);
        w.aq[i] = c.take(B * Qn * hi); w.zq[i] = c.take(B * Qn * hi);
        if (i >= 1) { w.bcur[i] = c.take(B * hi); w.Wslot[i] = c.take(B * w.nslot * hi * hp); }
        else { w.bcur[i] = nullptr; w.Wslot[i] = nullptr; }
    }
    if (p.need_grad) {
        w.pWh = c.take(B * nt * N * H); w.pbh = c.take(B * nt * N); w.pb0 = c.take(B * nt * h0); w.pD = c.take(B * nt * S * h0);
        w.Whb = c.take(B * N * H); w.bhb = c.take(B * N); w.b0b = c.take(B * h0); w.Db = c.take(B * S * h0);
        w.X0 = c.take(B * S * maxh); w.X1 = c.take(B * S * maxh); w.eb = c.take(B * S * N); w.lb = c.take(B * S * N);
        w.A0bar = c.take(B * (S + Qn) * h0);
        for (size_t i = 0; i < L; ++i) {
            const size_t hi = p.h[i], hp = i ? p.h[i - 1] : 0;
            w.abar[i] = c.take(B * S * hi);
            if (i >= 1) {
                w.pW[i] = c.take(B * nt * hi * hp); w.pb[i] = c.take(B * nt * hi);
                w.Wb[i] = c.take(B * hi * hp); w.bb[i] = c.take(B * hi);
            } else { w.pW[i] = w.pb[i] = w.Wb[i] = w.bb[i] = nullptr; }
        }
    }
}


}  // namespace

extern "C" void fumi_dbg_set_epi_trace(void* p) { g_epi_trace = (unsigned long long*)p; }

size_t episode_workspace_bytes(const EpisodeProblem& p) {
    Carver c{nullptr, 0};
    EpiBuf w;
    w.trace = nullptr;
    carve(c, p, w);
    if (p.need_grad) {
        int kc;
        const size_t ns = (size_t)xpanel_bwd_nsplit(p.B, p.S, p.Qn, p.D, p.h[0], &kc);
        c.bytes += ws_align(ns * p.h[0] * (size_t)p.D * sizeof(float));
    }
    return c.bytes;
}

int run_episodes(fumi_ws* ws, hipStream_t st, const EpisodeProblem& p) {
    if (p.L < 1 || p.L > MAXL || p.B < 1 || p.N < 1 || p.S < 1 || p.Qn < 1 || p.D < 1 || p.T < 0) return FUMI_EINVAL;
    for (int i = 0; i < p.L; ++i) if (p.h[i] < 1) return FUMI_EINVAL;
    EpiBuf w;
    Carver c{ws, 0};
    carve(c, p, w);
    w.trace = g_epi_trace;
    EpiDims d;
    d.B = p.B; d.N = p.N; d.S = p.S; d.Qn = p.Qn; d.L = p.L; d.T = p.T; d.H = p.h[p.L - 1];
    for (int i = 0; i < MAXL; ++i) d.h[i] = i < p.L ? p.h[i] : 0;
    d.alpha = p.alpha; d.need_grad = p.need_grad; d.second_order = p.second_order;
    d.taped = (p.need_grad && p.second_order && p.T > 0) ? 1 : 0;
    d.drop_thr = 0; d.mscale = 1.f; d.seed_lo = (unsigned)(p.seed & 0xffffffffULL); d.seed_hi = (unsigned)(p.seed >> 32);
    if (p.dropout_p > 0.f) {
        if (!(p.dropout_p < 1.f)) return FUMI_EINVAL;
        d.drop_thr = (unsigned)((double)p.dropout_p * 4294967296.0);
        if (d.drop_thr == 0) d.drop_thr = 1;
        d.mscale = 1.f / (1.f - p.dropout_p);
    }
    const int h0 = p.h[0];
    int rc;

    // ---- shared pass 1 over X: [A0 | G] = [Xs;Xq] [W0;Xs]^T for every episode, one launch (xpanel.hip)
    {
        ProfScope ps(ws, st, FUMI_PH_XPANEL_FWD);
        if ((rc = launch_xpanel_fwd(st, p.B, p.S, p.Qn, p.D, h0, p.x_s, p.x_q, p.W[0], w.A0, w.G))) return rc;
    }
    // ---- per-episode phases
    EpiParams prm;
    for (int i = 0; i < MAXL; ++i) { prm.W[i] = i < p.L ? p.W[i] : nullptr; prm.b[i] = i < p.L ? p.b[i] : nullptr; }
    {
        ProfScope ps(ws, st, FUMI_PH_ADAPT);
        HIP_TRY(hipFuncSetAttribute((const void*)adapt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, w.lds_adapt * 4));
        hipLaunchKernelGGL(adapt_kernel, dim3(p.B), dim3(512), w.lds_adapt * 4, st, d, w, prm, p.y_s, p.head, ws->status);
        LAUNCH_CHECK();
    }
    {
        ProfScope ps(ws, st, FUMI_PH_QUERY);
        QLay ql; query_layout(ql, p.L, p.h, p.S, p.N);
        static const bool force_global = getenv("FUMI_EPI_GLOBAL") != nullptr;     // dev/test: take the generic kernels
        bool lds_form = ql.total <= QLDS_CAP && 7 + 2 * (p.L - 1) <= WG_MAXJOB && !force_global;
        StageTab tb; tb.njobs = 0; tb.nunits = 0;
        if (lds_form) {
            // staging plan: source = base + episode*sb + tile*st
            const long R = p.S + p.Qn, S = p.S, N = p.N, H = d.H;
            const long slot = d.taped ? p.T : 0;
            tb.add(w.A0 + S * h0, R * h0, (long)QR * h0, h0, -1, QR, h0, ql.a[0], wg_ld(h0));
            tb.add(w.D, S * h0, 0, h0, p.S, p.S, h0, ql.D, wg_ld(h0));
            for (int i = 1; i < p.L; ++i) {
                const long sz = (long)p.h[i] * p.h[i - 1];
                tb.add(w.Wslot[i] + slot * sz, w.nslot * sz, 0, p.h[i - 1], p.h[i], p.h[i], p.h[i - 1], ql.W[i], wg_ld(p.h[i - 1]));
            }
            tb.add(w.G + S * S, R * S, (long)QR * S, S, -1, QR, p.S, ql.Gq, wg_ld(p.S));
            tb.add(w.Whslot + slot * N * H, w.nslot * N * H, 0, H, p.N, p.N, (int)H, ql.Wh, wg_ld((int)H));
            tb.add(p.b[0], 0, 0, h0, 1, 1, h0, ql.b0, h0);
            tb.add(w.cs, h0, 0, h0, 1, 1, h0, ql.cs, h0);
            for (int i = 1; i < p.L; ++i) tb.add(w.bcur[i], p.h[i], 0, p.h[i], 1, 1, p.h[i], ql.bi[i], p.h[i]);
            tb.add(w.bh, N, 0, N, 1, 1, p.N, ql.bh, p.N);
            lds_form = tb.nunits <= 64 * 8;                 // wg_stage_rows: one unit per lane and wave
        }
        if (lds_form) {
            HIP_TRY(hipFuncSetAttribute((const void*)query_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ql.total * 4));
            hipLaunchKernelGGL(query_lds_kernel, dim3(w.ntile, p.B), dim3(512), ql.total * 4, st, tb, d, w, ql, p.y_q,
                               p.logits_q, p.preds_q, ws->status);
        } else {
            HIP_TRY(hipFuncSetAttribute((const void*)query_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, w.lds_query * 4));
            hipLaunchKernelGGL(query_kernel, dim3(w.ntile, p.B), dim3(512), w.lds_query * 4, st, d, w, p.b[0], p.y_q, p.logits_q,
                               p.preds_q, ws->status);
        }
        LAUNCH_CHECK();
    }
    {
        ProfScope ps(ws, st, FUMI_PH_REVERSE);
        HIP_TRY(hipFuncSetAttribute((const void*)reverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, w.lds_reverse * 4));
        hipLaunchKernelGGL(reverse_kernel, dim3(p.B), dim3(512), w.lds_reverse * 4, st, d, w, p.loss_b, p.acc_b, p.head_bar);
        LAUNCH_CHECK();
    }
    // ---- sums over episodes in one launch: meta-gradients of the hidden layers, layer-0 bias, loss/accuracy totals
    {
        ProfScope pr(ws, st, FUMI_PH_REDUCE);
        ReduceSegs sg; sg.n = 0; sg.scale = p.grad_scale;
        if (p.stats) { sg.add(p.loss_b, p.B, 1, 1, p.stats); sg.add(p.acc_b, p.B, 1, 1, p.stats + 1); }
        if (p.need_grad) {
            for (int i = 1; i < p.L; ++i) {
                const long sz = (long)p.h[i] * p.h[i - 1];
                sg.add(w.Wb[i], p.B, sz, sz, p.gW[i]);
                sg.add(w.bb[i], p.B, p.h[i], p.h[i], p.gb[i]);
            }
            sg.add(w.b0b, p.B, h0, h0, p.gb[0]);
        }
        if ((rc = launch_reduce_multi(st, sg))) return rc;
    }
    if (!p.need_grad) return FUMI_OK;
    // ---- shared pass 2 over X: gW0 = Abar0^T [Xs;Xq], contraction over all B*R rows split into slabs (xpanel.hip)
    {
        ProfScope pg(ws, st, FUMI_PH_XPANEL_BWD);
        int kc;
        const int ns = xpanel_bwd_nsplit(p.B, p.S, p.Qn, p.D, h0, &kc);
        const long slab = (long)h0 * p.D;
        float* slabs = ws_f(ws, (size_t)ns * slab);
        if ((rc = launch_xpanel_bwd(st, p.B, p.S, p.Qn, p.D, h0, p.x_s, p.x_q, w.A0bar, slabs, kc, ns))) return rc;
        if ((rc = launch_reduce_slabs(st, slabs, ns, slab, slab, p.grad_scale, p.gW[0]))) return rc;
    }
    return FUMI_OK;
}

int launch_class_text_select(hipStream_t st, int B, int N, int S, int Dt, const float* text_s, const int64_t* y_s,
                             float* out, int* status) {
    hipLaunchKernelGGL(class_text_select_kernel, dim3(N, B), dim3(64), 0, st, N, S, Dt, text_s, y_s, out, status);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_broadcast_head(hipStream_t st, int B, int N, int H, const float* Wf, const float* bf, float* head) {
    hipLaunchKernelGGL(broadcast_head_kernel, dim3(B), dim3(256), 0, st, N, H, Wf, bf, head);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_tanh_bwd(hipStream_t st, long n, const float* h, const float* hbar, float* out) {
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, h, hbar, out);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_relu_mask_mul(hipStream_t st, long n, const float* u, float* g, float scale) {
    hipLaunchKernelGGL(relu_mask_mul_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, u, g, scale);
    LAUNCH_CHECK();
    return FUMI_OK;
}
int launch_split_head_grad(hipStream_t st, int B, int N, int H, const float* head_bar, float scale, float* gW, float* gb) {
    hipLaunchKernelGGL(split_head_grad_kernel, dim3((N * (H + 1) + 255) / 256), dim3(256), 0, st, B, N, H, head_bar, scale, gW, gb);
    LAUNCH_CHECK();
    return FUMI_OK;
}
