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
int g_spin_limit = 1 << 22;                    // polls before a cross-workgroup wait gives up (fumi_hip_set_spin_limit; tests set 0)
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
    float *A0bar;                              // adjoint of A0 (support rows: sum over inner steps): episode b's support rows at
    float *A0bar_q; int ldsr, ldq;             //   A0bar + b ldsr h0, its query rows at A0bar_q + b ldq h0 -- interleaved [B,R,h0] (ldsr = ldq =
                                               //   R, A0bar_q = A0bar + S h0) or split [B,S,h0] | [B,Qn,h0] (ldsr = S, ldq = Qn: the two halves
                                               //   of the backward X-panel pass run as two launches, run_episodes)
    float *apart; int *acnt;                   // adapt_lds split over column parts: [B,2,8,S*h_1] layer-1 partial sums; [B] arrival counters (persistent, zero between steps)
    float *xpart; int *xcnt;                   // reverse_lds: [B,2,8,S*h_1] partial sums exchanged between the column parts; [B] arrival counters
    int *status; int spin_limit;               // status word (FUMI_ST_SYNC_TIMEOUT when a wait for the sibling parts expires); polls per wait
    int nslot, ntape, ntile, maxh;
    unsigned long long* trace;                 // dev: per-phase wall-clock stamps of block 0 (tests/dev/trace_adapt.py)
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
#undef STAMP
}

// ------------------------------------------------------------------------------------------------------------
// adapt, LDS-resident form: the episode's whole inner loop runs out of LDS (common.h "LDS-resident products").
// Staged once: G_ss, the fast weights of every deeper layer, the head.  D_t, colsum(D_t), the activations a_i and
// the pre-activation gradients dz_i (i >= 1) live in LDS; dz_0 only exists inside the epilogue that adds it to D_t.
// The constant part of layer 0, A0_s + b0, sits in REGISTERS: the 16 x 64 block a wave owns in the layer-0 products
// never changes (needs ceil(S/16)*ceil(h0/64) <= 8 blocks, one per wave).  Global memory only sees the tape (stores,
// never waited for) and the final state.  Taken when adapt_layout().total <= ALDS_CAP and the block count fits.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int q_r4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int q_r16(int x) { return (x + 15) & ~15; }
constexpr int ALDS_CAP = 40000;
struct ALay { int a[MAXL], dz[MAXL], W[MAXL], bi[MAXL], Wh, G, D, e, cs, bh, total; };
__host__ __device__ inline void adapt_layout(ALay& y, int L, const int* h, int S, int N) {
    int off = 0;
    const int RS = q_r4(S);
    for (int i = 0; i < MAXL; ++i) { y.a[i] = y.dz[i] = y.W[i] = y.bi[i] = 0; }
    for (int i = 0; i < L; ++i) { y.a[i] = off; off += RS * wg_ld(h[i]); }
    for (int i = 1; i < L; ++i) { y.dz[i] = off; off += RS * wg_ld(h[i]); }
    y.G = off; off += RS * wg_ld(S);
    y.e = off; off += RS * wg_ld(N);
    y.D = off; off += RS * wg_ld(h[0]);
    for (int i = 1; i < L; ++i) { y.W[i] = off; off += q_r16(h[i]) * wg_ld(h[i - 1]); y.bi[i] = off; off += q_r4(h[i]); }
    y.Wh = off; off += q_r16(N) * wg_ld(h[L - 1]);
    y.cs = off; off += q_r4(h[0]);
    y.bh = off; off += q_r4(N);
    y.total = off + 64;
}

__global__ __launch_bounds__(512) void adapt_lds_kernel(StageTab stg, EpiDims d, EpiBuf w_arg, ALay y, EpiParams prm,
                                                        const int64_t* y_s, int* status, int P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ StageTab s_stg;
    __shared__ int s_lab[128];                   // support labels (the LDS-resident form holds S <= 128 rows)
    // the buffer table (~100 pointers) is read from an LDS copy: as a kernel argument every pointer is a scalar load from the
    // argument block at its first use, each with its own wait, scattered over the phases (2.7 us of the step loop's preheader)
    __shared__ EpiBuf s_w;
    {
        constexpr int off = (int)((sizeof(StageTab) + sizeof(EpiDims) + alignof(EpiBuf) - 1) / alignof(EpiBuf) * alignof(EpiBuf));
        const int* src = (const int*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
        int* dst = (int*)&s_w;
        for (int i = threadIdx.x; i < (int)(sizeof(EpiBuf) / 4); i += blockDim.x) dst[i] = src[i];
    }
    const EpiBuf& w = s_w;
    // P > 1: the episode's layer-0 columns are split over P workgroups (ids 8 apart: same XCD); c = this part
    const int tid = threadIdx.x, nt = blockDim.x;
    const int b = P > 1 ? (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / P) : (int)blockIdx.x;
    const int c = P > 1 ? (int)((blockIdx.x >> 3) % P) : 0;
    if (b >= d.B) return;
    const bool lead = c == 0;                    // quantities every part computes identically are written by part 0
    int stamp_i = 0;
#define STAMP() if (w_arg.trace && tid == 0 && blockIdx.x == 0) w_arg.trace[stamp_i++] = __builtin_amdgcn_s_memrealtime();
    STAMP()
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = d.S, N = d.N, L = d.L, H = d.H, h0f = d.h[0], h0 = h0f / P, c0 = c * h0;   // h0: this part's columns
    const float alpha = d.alpha;
    const int64_t* ys = y_s + (long)b * S;
    const int ldS = wg_ld(S), ldN = wg_ld(N), ld0 = wg_ld(h0), ldH = wg_ld(H);
    auto a = [&](int i) { return sm + y.a[i]; };
    auto dz = [&](int i) { return sm + y.dz[i]; };
    auto hw = [&](int i) { return i == 0 ? h0 : d.h[i]; };       // width of layer i's LDS images
    auto ldh = [&](int i) { return wg_ld(hw(i)); };
    float* D = sm + y.D; float* cs = sm + y.cs; float* e_ = sm + y.e; float* Wh = sm + y.Wh; float* bh = sm + y.bh;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    wg_stage_tab_to_lds(&s_stg, 1, (int)(sizeof(StageTab) + sizeof(EpiDims) + sizeof(EpiBuf) + sizeof(ALay) + sizeof(EpiParams) + 64));
    for (int i = tid * 4, tot = y.total; i < tot; i += nt * 4) *(f32x4*)(sm + i) = z4;
    // labels and A0_s + b0 are loaded now: a load issued inside the step loop would wait for every older tape store
    // (vmcnt counts loads and stores together, in order)
    const int my_label = tid < S ? label(ys, tid, N, status) : 0;
    if (tid < S && tid < 128) s_lab[tid] = my_label;          // (read by the soft-max lanes of a row: published by the barriers before the loop)
    // A0_s + b0 of the lane's 4x4 block of the layer-0 products (block = wave)
    f32x4 preZ[4];
    {
        const int tn = (h0 + 63) >> 6;
        const int m0 = (wave / tn) << 4, n = ((wave % tn) << 6) + 4 * (lane & 15);
        const float* A0s = w_arg.A0 + (long)b * (S + d.Qn) * h0f + c0;      // (before the LDS copy of the table is visible)
        const float* b0p = prm.b[0] + c0;
        const bool vec = ((h0f & 3) == 0) && ((c0 & 3) == 0) && ((((uintptr_t)A0s) & 15) == 0) && ((((uintptr_t)b0p) & 15) == 0);
        if (vec) {                               // unconditional loads from clamped addresses: all eight in flight together
            const int nc = n < h0 ? n : 0;
            const f32x4 bv = *(const f32x4*)(b0p + nc);
            f32x4 av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int m = m0 + 4 * (lane >> 4) + e; av[e] = *(const f32x4*)(A0s + (long)(m < S ? m : 0) * h0f + nc); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int m = m0 + 4 * (lane >> 4) + e; preZ[e] = (m < S && n < h0) ? av[e] + bv : z4; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + 4 * (lane >> 4) + e;
                f32x4 v = z4;
                if (m < S && n < h0) for (int cc = 0; cc < 4 && n + cc < h0; ++cc) v[cc] = A0s[(long)m * h0f + n + cc] + b0p[n + cc];
                preZ[e] = v;
            }
        }
    }
    wg_lds_barrier(); STAMP()                   // (LDS only: the preZ loads stay in flight behind the staging)
    wg_stage_rows<12>(&s_stg, b, 0, c, 0, sm);
    wg_lds_barrier(); STAMP()
    // slot 0 of the tape = the initial fast weights (reverse needs W_t of every step)
    auto store_img = [&](float* dst, long drs, const float* img, int ld, int rows, int cols) {      // LDS image -> global rows of stride drs
        const int c4n = (cols + 3) >> 2;
        for (int i = tid; i < rows * c4n; i += nt) {
            const int r = i / c4n, cc = (i - r * c4n) << 2;
            wg_st4(dst + (long)r * drs + cc, *(const f32x4*)(img + r * ld + cc), min(4, cols - cc));
        }
    };
    // fast weights of layer i in the tape / final slot: layer 1 holds this part's columns only
    auto store_W = [&](int i, long slot) {
        const long sz = (long)d.h[i] * d.h[i - 1];
        float* dst = w.Wslot[i] + ((long)b * w.nslot + slot) * sz;
        if (i == 1) store_img(dst + c0, h0f, sm + y.W[1], ldh(0), d.h[1], h0);
        else if (lead) store_img(dst, d.h[i - 1], sm + y.W[i], ldh(i - 1), d.h[i], d.h[i - 1]);
    };
    auto store_slot0 = [&]() {                 // (the images are only modified by the updates at the end of a step)
        for (int i = 1; i < L; ++i) store_W(i, 0);
        if (lead) store_img(w.Whslot + (long)b * w.nslot * N * H, H, Wh, ldH, N, H);
    };
    if (d.taped && d.T == 0) store_slot0();

    for (int t = 0; t < d.T; ++t) {
        const long tp = (long)b * w.ntape + (d.taped ? t : 0);
        // 1. layer 0 through the low-rank form: a0 = relu(A0s + b0 - alpha (G D_t + colsum D_t))
        {
            float* a0 = a(0);
            float* ta0 = w.ta[0] + tp * S * h0f + c0;
            const unsigned key0 = drop_key(d, b, t, 0);
            auto fin = [&](int m, int n, const f32x4& pre, int cnt) {
                const f32x4 v = drop_relu4(d, key0, (long)m * h0f + c0 + n, pre);
                f32x4 o = v;
#pragma unroll
                for (int c = 1; c < 4; ++c) if (c >= cnt) o[c] = 0.f;
                *(f32x4*)(a0 + m * ld0 + n) = o;
                if (d.taped) wg_st4(ta0 + (long)m * h0f + n, o, cnt);
            };
            STAMP()
            if (t == 0) {                         // D_0 = 0
                const int tn = (h0 + 63) >> 6;
                const int m0 = (wave / tn) << 4, n = ((wave % tn) << 6) + 4 * (lane & 15);
                if (wave < ((S + 15) >> 4) * tn && n < h0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = m0 + 4 * (lane >> 4) + e;
                        if (m < S) fin(m, n, preZ[e], min(4, h0 - n));
                    }
                }
            } else {
                wg_lmm_wide<true>(S, h0, S, sm + y.G, ldS, D, ld0, [&](int m, int n, const f32x4& acc, int cnt, auto ec) {
                    fin(m, n, preZ[decltype(ec)::value] - alpha * (acc + *(const f32x4*)(cs + n)), cnt);
                });
            }
        }
        STAMP()
        if (t == 0 && d.taped) store_slot0();    // after the registers loaded at kernel start have been consumed
        STAMP()
        wg_lds_barrier(); STAMP()
        // 2. deeper layers with the episode's fast weights
        for (int i = 1; i < L; ++i) {
            const int hi = d.h[i], hp = hw(i - 1);
            float* ai = a(i); const int ldi = ldh(i);
            const float* bi = sm + y.bi[i];
            float* tai = w.ta[i] + tp * S * hi;
            const unsigned keyi = drop_key(d, b, t, i);
            auto act = [&](int m, int n, const f32x4& pre, int cnt) {
                f32x4 o = drop_relu4(d, keyi, (long)m * hi + n, pre + *(const f32x4*)(bi + n));
#pragma unroll
                for (int cc = 1; cc < 4; ++cc) if (cc >= cnt) o[cc] = 0.f;
                *(f32x4*)(ai + m * ldi + n) = o;
                if (d.taped && lead) wg_st4(tai + (long)m * hi + n, o, cnt);
            };
            if (i == 1 && P > 1) {
                // this part's share of a_0 W_1^T (contraction over its columns) -> exchange -> every part forms a_1
                float* Xp = dz(1);                                  // free until the backward half of the step
                wg_lmm<true, true>(S, hi, hp, a(0), ldh(0), sm + y.W[1], wg_ld(hp), [&](int m, int n, const f32x4& acc, int) {
                    *(f32x4*)(Xp + m * ldi + n) = acc;
                });
                wg_lds_barrier();
                float* xb = w.apart + (((long)b * 2 + (t & 1)) * 8) * (long)S * hi;
                float* mine = xb + (long)c * S * hi;
                for (int e2 = tid; e2 < S * hi; e2 += nt) {
                    const int m = e2 / hi, n = e2 - m * hi;
                    __hip_atomic_store(mine + e2, Xp[m * ldi + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                wg_drain_stores();                                  // every wave: its sc1 partial stores have completed ...
                __syncthreads();                                    // ... before the one lane that signals for all of them does
                if (tid == 0) {
                    __hip_atomic_fetch_add(w.acnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int target = P * (t + 1);
                    // bounded: the parts of an episode are dispatched together (see reverse_lds_kernel); a wait that expires
                    // is reported (FUMI_ST_SYNC_TIMEOUT -> the host raises), never silently carried on from
                    int spin = 0;
                    while (__hip_atomic_load(w.acnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        if (++spin > w.spin_limit) { atomicOr(w.status, FUMI_ST_SYNC_TIMEOUT); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                __syncthreads();
                const int w4 = q_r4(hi);
                for (int e2 = tid; e2 < S * (w4 >> 2); e2 += nt) {
                    const int m = e2 / (w4 >> 2), n = (e2 - m * (w4 >> 2)) << 2;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    const int cnt = min(4, hi - n);
                    for (int cc = 0; cc < P; ++cc)
                        for (int e3 = 0; e3 < cnt; ++e3)
                            v[e3] += __hip_atomic_load(xb + (long)cc * S * hi + m * hi + n + e3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    act(m, n, v, cnt);
                }
            } else {
                wg_lmm<true, true>(S, hi, hp, a(i - 1), ldh(i - 1), sm + y.W[i], wg_ld(hp), [&](int m, int n, const f32x4& acc, int cnt) { act(m, n, acc, cnt); });
            }
            wg_lds_barrier(); STAMP()
        }
        // 3. head logits (into e_), softmax, e = (p - onehot)/S
        wg_lmm<true, true>(S, N, H, a(L - 1), ldH, Wh, ldH, [&](int m, int n, const f32x4& acc, int) {
            *(f32x4*)(e_ + m * ldN + n) = acc + *(const f32x4*)(bh + n);          // bh zero-padded
        });
        wg_lds_barrier(); STAMP()
        {
            // one lane per (row, class): G = 8 / 16 / 32 / 64 lanes per row, every lane walks its row's N logits through shuffles in
            // class order -- the same max / sum, in the same order, as one thread per row computed, at 1 / N of the chain length
            // (one thread per row: S = 25 lanes of the workgroup busy for ~500 dependent instructions, 3.3 us per inner step)
            float* tpp = w.tp + tp * S * N; float* tee = w.te + tp * S * N;
            const int G = N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : 64;
            const int per = nt / G, gl = tid & (G - 1), base = (tid & 63) & ~(G - 1);
            for (int s0 = 0; s0 < S; s0 += per) {
                const int s_ = s0 + tid / G;
                const bool rok = s_ < S, live = rok && gl < N;
                float* row = e_ + (rok ? s_ : 0) * ldN;
                const float x = live ? row[gl] : -INFINITY;
                float mx = __shfl(x, base, 64);
                for (int n = 1; n < N; ++n) mx = fmaxf(mx, __shfl(x, base + n, 64));
                const float ex = live ? expf(x - mx) : 0.f;
                float sum = 0.f;
                for (int n = 0; n < N; ++n) sum += __shfl(ex, base + n, 64);
                const float inv = 1.f / sum;
                if (live) {
                    const int yy = s_lab[s_];
                    const float pv = ex * inv;
                    const float ev = (pv - (gl == yy ? 1.f : 0.f)) / (float)S;
                    row[gl] = ev;
                    if (d.taped && lead) { tpp[s_ * N + gl] = pv; tee[s_ * N + gl] = ev; }
                }
                if (rok) for (int n = gl; n < ldN - 4; n += G) if (n >= N) row[n] = 0.f;      // K padding of e
            }
        }
        wg_lds_barrier(); STAMP()
        // 4. backward through the head: dz_{L-1} = (e Wh) * relu'   (reads Wh before it is updated)
        {
            const float* al = a(L - 1);
            float* tdl = w.tdz[L - 1] + tp * S * H;
            auto put = [&](int m, int n, const f32x4& acc, int cnt) {
                const f32x4 act = *(const f32x4*)(al + m * ldH + n);
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (c < cnt && act[c] > 0.f) ? acc[c] * d.mscale : 0.f;
                if (L > 1) *(f32x4*)(dz(L - 1) + m * ldH + n) = o;
                else {                                                          // single hidden layer: this is dz_0
                    float* pd = D + m * ld0 + n; *(f32x4*)pd = *(const f32x4*)pd + o;
                }
                if (d.taped && lead) wg_st4(tdl + (long)m * H + n, o, cnt);
            };
            wg_lmm_wide<true>(S, H, N, e_, ldN, Wh, ldH, [&](int m, int n, const f32x4& acc, int cnt, auto) { put(m, n, acc, cnt); });
        }
        wg_lds_barrier(); STAMP()
        // head update: Wh <- Wh - alpha e^T a,  bh <- bh - alpha colsum(e); next slot of the tape
        {
            float* Whn = w.Whslot + ((long)b * w.nslot + (d.taped ? t + 1 : 0)) * N * H;
            wg_lmm_wide<false>(N, H, S, e_, ldN, a(L - 1), ldH, [&](int m, int n, const f32x4& acc, int cnt, auto) {
                float* pw = Wh + m * ldH + n;
                const f32x4 o = *(const f32x4*)pw - alpha * acc;
                *(f32x4*)pw = o;
                if (d.taped && lead) wg_st4(Whn + m * H + n, o, cnt);
            });
            wg_lcolsum(S, N, e_, ldN, [&](int n, float s_) { bh[n] -= alpha * s_; });
        }
        // 5. hidden layers, top down (the head update above touches neither dz_{L-1} nor W_i)
        for (int i = L - 1; i >= 1; --i) {
            const int hi = d.h[i], hp = hw(i - 1);
            const float* ap = a(i - 1); const int ldp = ldh(i - 1), ldi = ldh(i);
            const long gst = i == 1 ? h0f : hp, gc0 = i == 1 ? c0 : 0;          // global row stride / first column of this part
            const bool wr = i == 1 || lead;
            float* tdp = w.tdz[i - 1] + tp * S * gst + gc0;
            wg_lmm_wide<true>(S, hp, hi, dz(i), ldi, sm + y.W[i], wg_ld(hp), [&](int m, int n, const f32x4& acc, int cnt, auto) {
                const f32x4 act = *(const f32x4*)(ap + m * ldp + n);
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (c < cnt && act[c] > 0.f) ? acc[c] * d.mscale : 0.f;
                if (i > 1) *(f32x4*)(dz(i - 1) + m * ldp + n) = o;
                else { float* pd = D + m * ld0 + n; *(f32x4*)pd = *(const f32x4*)pd + o; }     // 6. D_{t+1} = D_t + dz_0
                if (d.taped && wr) wg_st4(tdp + (long)m * gst + n, o, cnt);
            });
            wg_lds_barrier(); STAMP()
            float* Wi = sm + y.W[i]; const int ldw = wg_ld(hp);
            float* Wn = w.Wslot[i] + ((long)b * w.nslot + (d.taped ? t + 1 : 0)) * ((long)hi * gst) + gc0;
            wg_lmm_wide<false>(hi, hp, S, dz(i), ldi, ap, ldp, [&](int m, int n, const f32x4& acc, int cnt, auto) {
                float* pw = Wi + m * ldw + n;
                const f32x4 o = *(const f32x4*)pw - alpha * acc;
                *(f32x4*)pw = o;
                if (d.taped && wr) wg_st4(Wn + (long)m * gst + n, o, cnt);
            });
            float* bi = sm + y.bi[i];
            wg_lcolsum(S, hi, dz(i), ldi, [&](int n, float s_) { bi[n] -= alpha * s_; });
            if (i == 1) wg_lcolsum(S, h0, D, ld0, [&](int n, float s_) { cs[n] = s_; });           // colsum(D_{t+1})
        }
        if (L == 1) { wg_lds_barrier(); wg_lcolsum(S, h0, D, ld0, [&](int n, float s_) { cs[n] = s_; }); }
        wg_lds_barrier(); STAMP()
    }
    // ---- final state for the query tiles / the reverse sweep
    store_img(w.D + (long)b * S * h0f + c0, h0f, D, ld0, S, h0);
    for (int n = tid; n < h0; n += nt) w.cs[(long)b * h0f + c0 + n] = cs[n];
    if (lead) {
        for (int n = tid; n < N; n += nt) w.bh[(long)b * N + n] = bh[n];
        for (int i = 1; i < L; ++i) {
            const float* bi = sm + y.bi[i];
            for (int n = tid; n < d.h[i]; n += nt) w.bcur[i][(long)b * d.h[i] + n] = bi[n];
        }
    }
    if (!d.taped) {
        for (int i = 1; i < L; ++i) store_W(i, 0);
        if (lead) store_img(w.Whslot + (long)b * w.nslot * N * H, H, Wh, ldH, N, H);
    }
    STAMP()
#undef STAMP
}

// ------------------------------------------------------------------------------------------------------------
// query: forward with the adapted weights, loss / argmax, first-order backward of the query loss (partial slabs)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void query_kernel(EpiDims d, EpiBuf w, const float* const b0, const int64_t* y_q,
                                                    float* logits_q, int64_t* preds_q, float* preds_f, int* status) {
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
        if (preds_f) preds_f[(long)b * Qn + r0 + m] = (float)arg;
        for (int n = 0; n < N; ++n)
            lbar[m * N + n] = (expf(lq[m * N + n] - mx) * inv - (n == y ? 1.f : 0.f)) / (float)Qn;
    }
    __syncthreads();
    if (tid == 0) {
        float ls = 0.f, cs_ = 0.f;
        for (int m = 0; m < nr; ++m) { ls += s_loss[m]; cs_ += s_corr[m]; }
        w.ploss[(long)b * w.ntile + tile] = ls;
        w.pcorr[(long)b * w.ntile + tile] = cs_;
        if (tile == 0 && d.need_grad) w.xcnt[b] = 0;          // arrival counter of reverse_lds_kernel's exchanges
        if (tile == 0 && w.acnt) w.acnt[b] = 0;               // ... and of the next step's split adapt kernel
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
    float* A0bq = w.A0bar_q + ((long)b * w.ldq + r0) * h0;
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
struct QLay { int a[MAXL], W[MAXL], bi[MAXL], Wh, Wh2, Gq, D, lq, b0, cs, pb0, bh, total; };
__host__ __device__ inline void query_layout(QLay& y, int L, const int* h, int S, int N, bool fused = false) {
    int off = 0;
    for (int i = 0; i < MAXL; ++i) { y.a[i] = y.W[i] = y.bi[i] = 0; }
    for (int i = 0; i < L; ++i) { y.a[i] = off; off += QR * wg_ld(h[i]); }
    for (int i = 1; i < L; ++i) { y.W[i] = off; off += q_r16(h[i]) * wg_ld(h[i - 1]); y.bi[i] = off; off += q_r4(h[i]); }
    y.Wh = off; off += q_r16(N) * wg_ld(h[L - 1]);
    y.Wh2 = off; if (fused) off += q_r16(N) * wg_ld(h[L - 1]);       // (fused inner step: the updated head next to the old one)
    y.Gq = off; off += QR * wg_ld(S);
    y.D = off; off += q_r4(S) * wg_ld(h[0]);
    y.lq = off; off += QR * wg_ld(N);
    y.b0 = off; off += q_r4(h[0]);
    y.cs = off; off += q_r4(h[0]);
    y.pb0 = off; off += q_r4(h[0]);
    y.bh = off; off += q_r4(N);
    y.total = off;
}

// FUSED (one inner step, two layers, S <= 32: the reference configuration): every query tile first runs the episode's inner step on
// the support rows ITSELF -- redundantly in each of the episode's tiles, which costs nothing on a chip that the 32 one-workgroup
// adapt launches left 7/8 idle -- so the adapt kernel, its launch and the round trip of the fast weights through memory go away.
// The support activations borrow the tile's own images (a_0, a_1, lq) before the query rows are staged into them; the updated head
// is built next to the old one (the backward through the head reads the old); tile 0 writes the tape the reverse sweep reads.
template <bool FUSED>
__global__ __launch_bounds__(512) void query_lds_kernel(StageTab stg, StageTab stg_sup, EpiDims d, EpiBuf w, QLay y, const int64_t* y_q,
                                                        float* logits_q, int64_t* preds_q, float* preds_f, int* status,
                                                        const int64_t* y_s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float s_loss[QR];
    __shared__ float s_corr[QR];
    __shared__ StageTab s_stg2[2];
    __shared__ int s_lab[QR];
    StageTab& s_stg = s_stg2[0];
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

    // the row's label is loaded now: inside the chain it would wait for every older store (vmcnt is in order)
    const int my_label = tid < nr ? label(y_q + (long)b * Qn + r0, tid, N, status) : 0;
    // ---- zero the arena (padding must read as zero), then stage everything this tile needs in one batch (plan: host)
    const int my_label_s = (FUSED && tid < S) ? label(y_s + (long)b * S, tid, N, status) : 0;
    if (FUSED && tid < S && tid < QR) s_lab[tid] = my_label_s;
    wg_stage_tab_to_lds(s_stg2, 2, (int)(2 * sizeof(StageTab) + sizeof(EpiDims) + sizeof(EpiBuf) + sizeof(QLay) + 64));
    for (int i = tid * 4, tot = y.total; i < tot; i += nt * 4) *(f32x4*)(sm + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads(); QSTAMP()
    int whq = y.Wh;                                                  // the head the query pass uses
    if constexpr (FUSED) {
        // the tape (identical in every tile of the episode) is written by three of them: a_0 by tile 0, the initial fast weights by
        // tile 1, the rest by tile 2 -- one tile writing all 105 KB was 1.5 us behind the others, and the launch lasts as long as it
        const int nt_ = (Qn + QR - 1) / QR;
        const bool lead = tile == 0 && d.taped;
        const bool lead_w = tile == (nt_ > 1 ? 1 : 0) && d.taped, lead_r = tile == (nt_ > 2 ? 2 : 0) && d.taped;
        const f32x4 z4s = {0.f, 0.f, 0.f, 0.f};
        // the tile's own A0 rows are requested NOW (4 float4 per thread: h0 <= 256) and written into a_0 when the support rows
        // are done with it: their latency disappears behind the inner step
        f32x4 qv[4]; bool qok[4];
        {
            const float* A0q = w.A0 + ((long)b * (S + Qn) + S + r0) * h0;
            const int c4n = h0 >> 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 512 * i, m = f / c4n, n = (f - m * c4n) << 2;
                qok[i] = m < nr;
                qv[i] = *(const f32x4*)(A0q + (long)(qok[i] ? m : 0) * h0 + (qok[i] ? n : 0));
            }
        }
        wg_stage_rows<20>(&s_stg2[1], b, tile, 0, nr, sm);           // A0_s rows -> a_0, W_1, b_1, the episode's head, b_0, the tile's G rows
        wg_lds_barrier();
        float* a0 = a(0); float* a1 = a(1); float* W1 = sm + y.W[1]; float* b1 = sm + y.bi[1];
        float* Wh = sm + y.Wh; float* Wh2 = sm + y.Wh2; float* bh = sm + y.bh; float* Dl = sm + y.D; float* csl = sm + y.cs;
        const float* b0l = sm + y.b0; float* e_ = sm + y.lq;
        const int h1 = d.h[1], ld1 = wg_ld(h1);
        const long tp = (long)b * w.ntape;                           // tape step 0
        auto store_img = [&](float* dst, long drs, const float* img, int ld, int rows, int cols) {      // LDS image -> global rows
            const int c4n = (cols + 3) >> 2;
            for (int i = tid; i < rows * c4n; i += nt) {
                const int r = i / c4n, cc = (i - r * c4n) << 2;
                wg_st4(dst + (long)r * drs + cc, *(const f32x4*)(img + r * ld + cc), min(4, cols - cc));
            }
        };
        if (lead_w) {                                                // slot 0 of the tape: the initial fast weights
            store_img(w.Wslot[1] + (long)b * w.nslot * ((long)h1 * h0), h0, W1, ld0, h1, h0);
            store_img(w.Whslot + (long)b * w.nslot * N * H, H, Wh, ldH, N, H);
        }
        {   // 1. a_0 = relu(A0_s + b_0)   (D_0 = 0)
            const unsigned key0 = drop_key(d, b, 0, 0);
            float* ta0 = w.ta[0] + tp * S * h0;
            const int c4n = h0 >> 2;
            for (int i = tid; i < S * c4n; i += nt) {
                const int m = i / c4n, n = (i - m * c4n) << 2;
                float* p = a0 + m * ld0 + n;
                const f32x4 v = drop_relu4(d, key0, (long)m * h0 + n, *(const f32x4*)p + *(const f32x4*)(b0l + n));
                *(f32x4*)p = v;
                if (lead) *(f32x4*)(ta0 + (long)m * h0 + n) = v;
            }
        }
        wg_lds_barrier();
        {   // 2. a_1
            const unsigned key1 = drop_key(d, b, 0, 1);
            float* ta1 = w.ta[1] + tp * S * h1;
            wg_lmm<true, true>(S, h1, h0, a0, ld0, W1, ld0, [&](int m, int n, const f32x4& acc, int cnt) {
                f32x4 o = drop_relu4(d, key1, (long)m * h1 + n, acc + *(const f32x4*)(b1 + n));
#pragma unroll
                for (int cc = 1; cc < 4; ++cc) if (cc >= cnt) o[cc] = 0.f;
                *(f32x4*)(a1 + m * ld1 + n) = o;
                if (lead_r) wg_st4(ta1 + (long)m * h1 + n, o, cnt);
            });
        }
        wg_lds_barrier();
        // 3. head logits (into e_), soft-max, e = (p - onehot) / S
        wg_lmm<true, true>(S, N, H, a1, ld1, Wh, ldH, [&](int m, int n, const f32x4& acc, int) {
            *(f32x4*)(e_ + m * ldq + n) = acc + *(const f32x4*)(bh + n);
        });
        wg_lds_barrier();
        {
            float* tpp = w.tp + tp * S * N; float* tee = w.te + tp * S * N;
            const int G = N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : 64;
            const int per = nt / G, gl = tid & (G - 1), base = (tid & 63) & ~(G - 1);
            for (int s0 = 0; s0 < S; s0 += per) {
                const int s_ = s0 + tid / G;
                const bool rok = s_ < S, live = rok && gl < N;
                float* row = e_ + (rok ? s_ : 0) * ldq;
                const float x = live ? row[gl] : -INFINITY;
                float mx = __shfl(x, base, 64);
                for (int n = 1; n < N; ++n) mx = fmaxf(mx, __shfl(x, base + n, 64));
                const float ex = live ? expf(x - mx) : 0.f;
                float sum = 0.f;
                for (int n = 0; n < N; ++n) sum += __shfl(ex, base + n, 64);
                const float inv = 1.f / sum;
                if (live) {
                    const float pv = ex * inv;
                    const float ev = (pv - (gl == s_lab[s_] ? 1.f : 0.f)) / (float)S;
                    row[gl] = ev;
                    if (lead_r) { tpp[s_ * N + gl] = pv; tee[s_ * N + gl] = ev; }
                }
                if (rok) for (int n = gl; n < ldq - 4; n += G) if (n >= N) row[n] = 0.f;      // K padding of e
            }
        }
        wg_lds_barrier();
        // 4. updated head next to the old one: Wh' = Wh - alpha e^T a_1,  bh' = bh - alpha colsum(e)
        wg_lmm_wide<false>(N, H, S, e_, ldq, a1, ld1, [&](int m, int n, const f32x4& acc, int, auto) {
            *(f32x4*)(Wh2 + m * ldH + n) = *(const f32x4*)(Wh + m * ldH + n) - alpha * acc;
        });
        wg_lcolsum(S, N, e_, ldq, [&](int n, float s_) { bh[n] -= alpha * s_; });
        wg_lds_barrier();
        {   // 5. dz_1 = (e Wh) * relu'(a_1), in place of a_1 (reads the OLD head)
            float* tdl = w.tdz[1] + tp * S * h1;
            wg_lmm_wide<true>(S, H, N, e_, ldq, Wh, ldH, [&](int m, int n, const f32x4& acc, int cnt, auto) {
                float* p = a1 + m * ld1 + n;
                const f32x4 act = *(const f32x4*)p;
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (c < cnt && act[c] > 0.f) ? acc[c] * d.mscale : 0.f;
                *(f32x4*)p = o;
                if (lead_r) wg_st4(tdl + (long)m * h1 + n, o, cnt);
            });
        }
        wg_lds_barrier();
        // 6. dz_0 = (dz_1 W_1) * relu'(a_0) = D_1
        wg_lmm_wide<true>(S, h0, h1, a1, ld1, W1, ld0, [&](int m, int n, const f32x4& acc, int cnt, auto) {
            const f32x4 act = *(const f32x4*)(a0 + m * ld0 + n);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (c < cnt && act[c] > 0.f) ? acc[c] * d.mscale : 0.f;
            *(f32x4*)(Dl + m * ld0 + n) = o;
        });
        wg_lds_barrier();
        // 7. W_1' = W_1 - alpha dz_1^T a_0 in place, b_1' = b_1 - alpha colsum(dz_1), colsum(D_1)
        wg_lmm_wide<false>(h1, h0, S, a1, ld1, a0, ld0, [&](int m, int n, const f32x4& acc, int, auto) {
            float* pw = W1 + m * ld0 + n;
            *(f32x4*)pw = *(const f32x4*)pw - alpha * acc;
        });
        wg_lcolsum(S, h1, a1, ld1, [&](int n, float s_) { b1[n] -= alpha * s_; });
        wg_lcolsum(S, h0, Dl, ld0, [&](int n, float s_) { csl[n] = s_; });
        wg_lds_barrier();
        whq = y.Wh2;
        // the query rows replace the support rows in a_0 (rows past the tile: zeros); a_1 and the logits image are cleared
        {
            const int c4n = h0 >> 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 512 * i, m = f / c4n, n = (f - m * c4n) << 2;
                if (m < QR) *(f32x4*)(a0 + m * ld0 + n) = qok[i] ? qv[i] : z4s;
            }
        }
        for (int i = tid; i < QR * (ld1 >> 2); i += nt) { const int m = i / (ld1 >> 2); *(f32x4*)(a1 + m * ld1 + ((i - m * (ld1 >> 2)) << 2)) = z4s; }
        for (int i = tid; i < QR * (ldq >> 2); i += nt) { const int m = i / (ldq >> 2); *(f32x4*)(e_ + m * ldq + ((i - m * (ldq >> 2)) << 2)) = z4s; }
        wg_lds_barrier(); QSTAMP()
    } else {
        wg_stage_rows<20>(&s_stg, b, tile, 0, nr, sm);
        wg_lds_barrier(); QSTAMP()
    }

    // ---- forward (epilogues handle 4 consecutive columns of one row: wg_lmm)
    {
        float* a0 = a(0);
        const float* b0l = sm + y.b0; const float* csl = sm + y.cs;
        const unsigned keyq0 = drop_key(d, b, QUERY_CALL, 0);
        wg_lmm_wide<true>(nr, h0, S, sm + y.Gq, ldG, sm + y.D, ld0, [&](int m, int n, const f32x4& acc, int, auto) {
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
        wg_lmm<true, true>(nr, N, H, a(L - 1), ldH, sm + whq, ldH, [&](int m, int n, const f32x4& acc, int cnt) {
            const f32x4 v = acc + *(const f32x4*)(bhl + n);        // bh is zero-padded: columns past N stay 0
            *(f32x4*)(lq + m * ldq + n) = v;
            wg_st4(lg + m * N + n, v, cnt);
        });
    }
    wg_lds_barrier(); QSTAMP()
    // per row: log-softmax loss, first arg-max (torch.max semantics, fumi.py:180), lbar = (p - onehot)/Qn in place
    for (int m = tid; m < nr; m += nt) {
        const int yy = my_label;                                   // nr <= QR <= blockDim: m == tid
        float* row = lq + m * ldq;
        float mx = row[0]; int arg = 0;
        for (int n = 1; n < N; ++n) { const float v = row[n]; if (v > mx) { mx = v; arg = n; } }
        float sum = 0.f;
        for (int n = 0; n < N; ++n) sum += expf(row[n] - mx);
        const float lse = mx + logf(sum), inv = 1.f / sum;
        s_loss[m] = lse - row[yy];
        s_corr[m] = (arg == yy) ? 1.f : 0.f;
        preds_q[(long)b * Qn + r0 + m] = arg;
        if (preds_f) preds_f[(long)b * Qn + r0 + m] = (float)arg;
        for (int n = 0; n < N; ++n) row[n] = (expf(row[n] - mx) * inv - (n == yy ? 1.f : 0.f)) / (float)Qn;
    }
    wg_lds_barrier(); QSTAMP()
    if (tid == 0) {
        float ls = 0.f, cs_ = 0.f;
        for (int m = 0; m < nr; ++m) { ls += s_loss[m]; cs_ += s_corr[m]; }
        w.ploss[(long)b * w.ntile + tile] = ls;
        w.pcorr[(long)b * w.ntile + tile] = cs_;
        if (tile == 0 && d.need_grad) w.xcnt[b] = 0;          // arrival counter of reverse_lds_kernel's exchanges
        if (tile == 0 && w.acnt) w.acnt[b] = 0;               // ... and of the next step's split adapt kernel
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
        wg_lmm_wide<false>(N, H, nr, lbar, ldq, a(L - 1), ldH, [&](int m, int n, const f32x4& acc, int cnt, auto) {
            wg_st4(pWh + m * H + n, acc, cnt);
        });
        float* pbh = w.pbh + pt * N;
        wg_lcolsum(nr, N, lbar, ldq, [&](int n, float s) { pbh[n] = s; });
    }
    wg_lds_barrier(); QSTAMP()
    {
        float* zl = a(L - 1);
        wg_lmm_wide<true>(nr, H, N, lbar, ldq, sm + whq, ldH, [&](int m, int n, const f32x4& acc, int, auto) {
            float* p = zl + m * ldH + n;
            *(f32x4*)p = relu_bwd4(*(const f32x4*)p, acc);
        });
    }
    wg_lds_barrier(); QSTAMP()
    for (int i = L - 1; i >= 1; --i) {
        const int hi = d.h[i], hp = d.h[i - 1];
        const float* zi = a(i); const int ldi = lda(i), ldp = lda(i - 1);
        float* pWi = w.pW[i] + pt * (long)hi * hp;
        wg_lmm_wide<false>(hi, hp, nr, zi, ldi, a(i - 1), ldp, [&](int m, int n, const f32x4& acc, int cnt, auto) {
            wg_st4(pWi + (long)m * hp + n, acc, cnt);
        });
        float* pbi = w.pb[i] + pt * hi;
        wg_lcolsum(nr, hi, zi, ldi, [&](int n, float s) { pbi[n] = s; });
        wg_lds_barrier(); QSTAMP()
        float* zp = a(i - 1);
        wg_lmm_wide<true>(nr, hp, hi, zi, ldi, sm + y.W[i], wg_ld(hp), [&](int m, int n, const f32x4& acc, int, auto) {
            float* p = zp + m * ldp + n;
            *(f32x4*)p = relu_bwd4(*(const f32x4*)p, acc);
        });
        wg_lds_barrier(); QSTAMP()
    }
    // layer 0: Abar0 rows of the query set, b0bar, and the adjoint of the low-rank factor D_T
    const float* z0 = a(0);
    float* A0bq = w.A0bar_q + ((long)b * w.ldq + r0) * h0;
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
    wg_lmm_wide<false>(S, h0, nr, sm + y.Gq, ldG, z0, ld0, [&](int m, int n, const f32x4& acc, int cnt, auto) {
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
    float* A0bs = w.A0bar + (long)b * w.ldsr * h0;
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
// reverse, LDS-resident form, split over P workgroups per episode along the columns of layer 0.
//
// Everything that is h0 wide -- Wbar_1, W_1, Dbar, a_0, the adjoints of a_0 / z_0 -- is a column block per workgroup
// (h0c = h0/P columns), which is what makes the sweep fit LDS (one workgroup would need Wbar_1 AND W_1 AND four
// [S,h0] matrices: > 160 KiB for the reference network) and puts P CUs on an episode instead of one.  All products that
// involve those matrices are column-local except one: dzbar_1 = dab_0 W_1^T - alpha a_0 Wbar_1^T contracts over h0.
// Each part computes its partial [S,h_1] sum, publishes it with agent-coherent (sc1) stores, bumps the episode's arrival
// counter and -- after doing its Wbar_1 update in the meantime -- waits for the other parts and adds the P partials in
// part order (so every part gets the same bits).  The deeper layers and the head are small and computed redundantly by
// every part; part 0 writes their adjoints.
// Workgroup ids: the parts of an episode are 8 ids apart (same XCD, dispatched together); a workgroup only ever waits
// for workgroups with neighbouring ids, so a grid larger than the chip cannot deadlock (in-order dispatch).
// ------------------------------------------------------------------------------------------------------------
constexpr int RLDS_CAP = 40000;
struct RLay {
    int Wb[MAXL], bb[MAXL], a[MAXL], dz[MAXL], W[MAXL], ab[MAXL];
    int Db, A0bs, b0b, Whb, bhb, Wh, p, e, eb, G, dab0, X0, X1, cs, ldX, total;
};
inline void reverse_layout(RLay& y, int L, const int* h, int S, int N, int P) {
    int off = 0;
    const int RS = q_r4(S), h0c = h[0] / P, H = h[L - 1];
    auto take = [&](int rows, int ld) { const int o = off; off += rows * ld; return o; };
    for (int i = 0; i < MAXL; ++i) y.Wb[i] = y.bb[i] = y.a[i] = y.dz[i] = y.W[i] = y.ab[i] = 0;
    int maxh1 = 4;
    for (int i = 1; i < L; ++i) maxh1 = h[i] > maxh1 ? h[i] : maxh1;
    y.ldX = wg_ld(maxh1);
    for (int i = 1; i < L; ++i) {
        const int kc = i == 1 ? h0c : h[i - 1];
        y.Wb[i] = take(q_r16(h[i]), wg_ld(kc)); y.W[i] = take(q_r16(h[i]), wg_ld(kc)); y.bb[i] = take(1, q_r4(h[i]));
        y.a[i] = take(RS, wg_ld(h[i])); y.dz[i] = take(RS, wg_ld(h[i])); y.ab[i] = take(RS, wg_ld(h[i]));
    }
    y.a[0] = take(RS, wg_ld(h0c)); y.ab[0] = take(RS, wg_ld(h0c)); y.dab0 = take(RS, wg_ld(h0c));
    y.Db = take(RS, wg_ld(h0c)); y.A0bs = take(RS, wg_ld(h0c));
    y.X0 = take(RS, y.ldX); y.X1 = take(RS, y.ldX);
    y.Whb = take(q_r16(N), wg_ld(H)); y.Wh = take(q_r16(N), wg_ld(H));
    y.p = take(RS, wg_ld(N)); y.e = take(RS, wg_ld(N)); y.eb = take(RS, wg_ld(N)); y.G = take(RS, wg_ld(S));
    y.b0b = take(1, q_r4(h0c)); y.cs = take(1, q_r4(h0c)); y.bhb = take(1, q_r4(N));
    y.total = off + 64;
}

__global__ __launch_bounds__(512) void reverse_lds_kernel(StageTab stg_init, StageTab stg_step, EpiDims d, EpiBuf w, RLay y,
                                                          int P, float* loss_b, float* acc_b, float* head_bar) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ StageTab s_stg[2];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int b = xcd + 8 * (jq / P), c = jq % P;
    if (b >= d.B) return;
    int ri = 0;
#define RSTAMP() if (w.trace && tid == 0 && blockIdx.x == 0) w.trace[128 + ri++] = __builtin_amdgcn_s_memrealtime();
    RSTAMP()
    const int S = d.S, N = d.N, L = d.L, H = d.H, h0 = d.h[0], h0c = h0 / P, c0 = c * h0c, h1 = d.h[1];
    const float alpha = d.alpha, ms = d.mscale;
    const int ntile = w.ntile;
    if (tid == 0 && c == 0) {
        float ls = 0.f, cr = 0.f;
        for (int t = 0; t < ntile; ++t) { ls += w.ploss[(long)b * ntile + t]; cr += w.pcorr[(long)b * ntile + t]; }
        loss_b[b] = ls / (float)d.Qn;
        acc_b[b] = cr / (float)d.Qn;
    }
    if (!d.need_grad) return;

    const int ldc = wg_ld(h0c), ldN = wg_ld(N), ldS = wg_ld(S), ldH = wg_ld(H), ldX = y.ldX;
    auto ldh = [&](int i) { return wg_ld(d.h[i]); };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto relu_bwd4 = [&](const f32x4& act, const f32x4& g) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act[e] > 0.f ? g[e] * ms : 0.f;
        return o;
    };
    // elementwise pass over the valid [S x cols] part of LDS images (4 columns per thread)
    auto ew = [&](int cols, auto&& f) {
        const int c4n = (cols + 3) >> 2;
        for (int i = tid; i < S * c4n; i += nt) { const int m = i / c4n; f(m, (i - m * c4n) << 2); }
    };
    auto store_img = [&](float* dst, long drs, const float* img, int ld, int rows, int cols) {   // LDS image -> global
        const int c4n = (cols + 3) >> 2;
        for (int i = tid; i < rows * c4n; i += nt) {
            const int r = i / c4n, cc = (i - r * c4n) << 2;
            wg_st4(dst + (long)r * drs + cc, *(const f32x4*)(img + r * ld + cc), min(4, cols - cc));
        }
    };

    wg_stage_tab_to_lds(s_stg, 2, (int)(2 * sizeof(StageTab) + sizeof(EpiDims) + sizeof(EpiBuf) + sizeof(RLay) + 64));
    for (int i = tid * 4, tot = y.total; i < tot; i += nt * 4) *(f32x4*)(sm + i) = z4;
    __syncthreads(); RSTAMP()
    // adjoints after the query pass = sums of the tiles' partial slabs (this part's columns); G_ss
    wg_stage_rows<3, 8>(&s_stg[0], b, 0, c, 0, sm);
    wg_lds_barrier(); RSTAMP()

    float* Wb1 = sm + y.Wb[1]; float* Db = sm + y.Db; float* A0bs = sm + y.A0bs; float* b0b = sm + y.b0b;
    float* Whb = sm + y.Whb; float* bhb = sm + y.bhb; float* cs = sm + y.cs;
    float* a0 = sm + y.a[0]; float* ab0 = sm + y.ab[0]; float* dab0 = sm + y.dab0; float* W1 = sm + y.W[1];
    float* Wh = sm + y.Wh; float* pp = sm + y.p; float* ee = sm + y.e; float* eb = sm + y.eb; float* G = sm + y.G;
    float* Xa = sm + y.X0; float* Xb = sm + y.X1;

    if (d.second_order) {
        for (int t = d.T - 1; t >= 0; --t) {
            const int round = d.T - 1 - t;
            wg_stage_rows<8, 1>(&s_stg[1], b, t, c, 0, sm);
            wg_lds_barrier(); RSTAMP()
            // abar_i = 0 ; dab_0 = Dbar * relu'(z_0)
            for (int i = 1; i < L; ++i) { float* abi = sm + y.ab[i]; const int ldi = ldh(i); ew(d.h[i], [&](int m, int n) { *(f32x4*)(abi + m * ldi + n) = z4; }); }
            ew(h0c, [&](int m, int n) {
                *(f32x4*)(ab0 + m * ldc + n) = z4;
                *(f32x4*)(dab0 + m * ldc + n) = relu_bwd4(*(const f32x4*)(a0 + m * ldc + n), *(const f32x4*)(Db + m * ldc + n));
            });
            wg_lds_barrier(); RSTAMP()
            // ---- layer 1 (split): this part's share of dzbar_1, abar_0 -= alpha dz_1 Wbar_1
            const float* dz1 = sm + y.dz[1]; const int ld1 = ldh(1);
            wg_lmm<true, true>(S, h1, h0c, dab0, ldc, W1, ldc, [&](int m, int n, const f32x4& acc, int) { *(f32x4*)(Xa + m * ldX + n) = acc; });
            wg_lmm<true, true>(S, h1, h0c, a0, ldc, Wb1, ldc, [&](int m, int n, const f32x4& acc, int) {
                float* px = Xa + m * ldX + n; *(f32x4*)px = *(const f32x4*)px - alpha * acc;
            });
            wg_lmm_wide<true>(S, h0c, h1, dz1, ld1, Wb1, ldc, [&](int m, int n, const f32x4& acc, int, auto) {
                float* pa = ab0 + m * ldc + n; *(f32x4*)pa = *(const f32x4*)pa - alpha * acc;
            });
            wg_lds_barrier(); RSTAMP()
            float* xb = w.xpart + (((long)b * 2 + (round & 1)) * 8) * (long)S * h1;
            if (P > 1) {
                float* mine = xb + (long)c * S * h1;
                for (int i = tid; i < S * h1; i += nt) {
                    const int m = i / h1, n = i - m * h1;
                    __hip_atomic_store(mine + i, Xa[m * ldX + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // Wbar_1 += dz_1^T dab_0   (column-local; runs while the partial sums travel)
            wg_lmm_wide<false>(h1, h0c, S, dz1, ld1, dab0, ldc, [&](int m, int n, const f32x4& acc, int, auto) {
                float* pw = Wb1 + m * ldc + n; *(f32x4*)pw = *(const f32x4*)pw + acc;
            });
            if (P > 1) {
                wg_drain_stores();                                  // every wave: its sc1 partial stores have completed ...
                __syncthreads();                                    // ... before the one lane that signals for all of them does
                if (tid == 0) {
                    __hip_atomic_fetch_add(w.xcnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int target = P * (round + 1);
                    // bounded: the parts of an episode are dispatched together, so this wait is microseconds; a grid that
                    // somehow lost a part must not hang the device -- the expired wait sets FUMI_ST_SYNC_TIMEOUT and the
                    // host raises when it reads the status word (the gradients of this step are not to be trusted)
                    int spin = 0;
                    while (__hip_atomic_load(w.xcnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        if (++spin > w.spin_limit) { atomicOr(w.status, FUMI_ST_SYNC_TIMEOUT); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                __syncthreads();
            } else {
                wg_lds_barrier();
            }
            RSTAMP()
            {   // dzbar_1 = sum of the parts - alpha bbar_1 ; dab_1 = dzbar_1 * relu'(z_1)  -> Xb
                const float* a1 = sm + y.a[1]; const float* bb1 = sm + y.bb[1];
                const int w4 = q_r4(h1);
                for (int i = tid; i < S * w4; i += nt) {
                    const int m = i / w4, n = i - m * w4;
                    float v = 0.f;
                    if (n < h1) {
                        if (P > 1) { for (int cc = 0; cc < P; ++cc) v += __hip_atomic_load(xb + (long)cc * S * h1 + m * h1 + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        else v = Xa[m * ldX + n];
                        v = a1[m * ld1 + n] > 0.f ? (v - alpha * bb1[n]) * ms : 0.f;
                    }
                    Xb[m * ldX + n] = v;
                }
            }
            wg_lds_barrier(); RSTAMP()
            int cur = 1;
            auto X = [&](int i) { return i ? Xb : Xa; };
            // ---- deeper layers (not split; every part computes them)
            for (int i = 2; i < L; ++i) {
                const int hi = d.h[i], hp = d.h[i - 1], ldi = ldh(i), ldp = ldh(i - 1);
                const float* dab = X(cur); float* nxt = X(cur ^ 1);
                const float* bbi = sm + y.bb[i]; float* Wbi = sm + y.Wb[i]; const float* ai = sm + y.a[i];
                const float* dzi = sm + y.dz[i]; float* abp = sm + y.ab[i - 1];
                wg_lmm<true, true>(S, hi, hp, dab, ldX, sm + y.W[i], ldp, [&](int m, int n, const f32x4& acc, int) {
                    *(f32x4*)(nxt + m * ldX + n) = acc - alpha * *(const f32x4*)(bbi + n);
                });
                wg_lmm<true, true>(S, hi, hp, sm + y.a[i - 1], ldp, Wbi, ldp, [&](int m, int n, const f32x4& acc, int) {
                    float* px = nxt + m * ldX + n;
                    *(f32x4*)px = relu_bwd4(*(const f32x4*)(ai + m * ldi + n), *(const f32x4*)px - alpha * acc);
                });
                wg_lmm_wide<true>(S, hp, hi, dzi, ldi, Wbi, ldp, [&](int m, int n, const f32x4& acc, int, auto) {
                    float* pa = abp + m * ldp + n; *(f32x4*)pa = *(const f32x4*)pa - alpha * acc;
                });
                wg_lds_barrier();
                wg_lmm_wide<false>(hi, hp, S, dzi, ldi, dab, ldX, [&](int m, int n, const f32x4& acc, int, auto) {
                    float* pw = Wbi + m * ldp + n; *(f32x4*)pw = *(const f32x4*)pw + acc;
                });
                wg_lds_barrier();
                cur ^= 1;
            }
            {   // ---- head
                const float* dab = X(cur); const float* aL = sm + y.a[L - 1]; float* abl = sm + y.ab[L - 1];
                wg_lmm<true, true>(S, N, H, dab, ldX, Wh, ldH, [&](int m, int n, const f32x4& acc, int) {
                    *(f32x4*)(eb + m * ldN + n) = acc - alpha * *(const f32x4*)(bhb + n);
                });
                wg_lmm<true, true>(S, N, H, aL, ldH, Whb, ldH, [&](int m, int n, const f32x4& acc, int) {
                    float* pe = eb + m * ldN + n; *(f32x4*)pe = *(const f32x4*)pe - alpha * acc;
                });
                wg_lmm_wide<true>(S, H, N, ee, ldN, Whb, ldH, [&](int m, int n, const f32x4& acc, int, auto) {
                    float* pa = abl + m * ldH + n; *(f32x4*)pa = *(const f32x4*)pa - alpha * acc;
                });
                wg_lds_barrier(); RSTAMP()
                wg_lmm_wide<false>(N, H, S, ee, ldN, dab, ldX, [&](int m, int n, const f32x4& acc, int, auto) {
                    float* pw = Whb + m * ldH + n; *(f32x4*)pw = *(const f32x4*)pw + acc;
                });
                // softmax-CE second derivative: lbar = p * (pbar - <p,pbar>),  pbar = ebar / S   (in place of ebar)
                for (int s_ = tid; s_ < S; s_ += nt) {
                    float* re = eb + s_ * ldN; const float* rp = pp + s_ * ldN;
                    float dot = 0.f;
                    for (int n = 0; n < N; ++n) dot += rp[n] * re[n];
                    for (int n = 0; n < N; ++n) re[n] = rp[n] * (re[n] - dot) / (float)S;
                    for (int n = N; n < q_r4(N); ++n) re[n] = 0.f;
                }
                wg_lds_barrier(); RSTAMP()
                const float* lb = eb;
                wg_lmm_wide<true>(S, H, N, lb, ldN, Wh, ldH, [&](int m, int n, const f32x4& acc, int, auto) {
                    float* pa = abl + m * ldH + n; *(f32x4*)pa = *(const f32x4*)pa + acc;
                });
                wg_lmm_wide<false>(N, H, S, lb, ldN, aL, ldH, [&](int m, int n, const f32x4& acc, int, auto) {
                    float* pw = Whb + m * ldH + n; *(f32x4*)pw = *(const f32x4*)pw + acc;
                });
                wg_lcolsum(S, N, lb, ldN, [&](int n, float s_) { bhb[n] += s_; });
                wg_lds_barrier(); RSTAMP()
            }
            // ---- reverse of the forward pass
            for (int i = L - 1; i >= 1; --i) {
                const int hi = d.h[i], ldi = ldh(i);
                float* zb = sm + y.ab[i]; const float* ai = sm + y.a[i];
                ew(hi, [&](int m, int n) { float* pz = zb + m * ldi + n; *(f32x4*)pz = relu_bwd4(*(const f32x4*)(ai + m * ldi + n), *(const f32x4*)pz); });
                wg_lds_barrier();
                float* bbw = sm + y.bb[i];
                if (i >= 2) {
                    const int hp = d.h[i - 1], ldp = ldh(i - 1);
                    float* abp = sm + y.ab[i - 1]; float* Wbi = sm + y.Wb[i];
                    wg_lmm_wide<true>(S, hp, hi, zb, ldi, sm + y.W[i], ldp, [&](int m, int n, const f32x4& acc, int, auto) {
                        float* pa = abp + m * ldp + n; *(f32x4*)pa = *(const f32x4*)pa + acc;
                    });
                    wg_lmm_wide<false>(hi, hp, S, zb, ldi, sm + y.a[i - 1], ldp, [&](int m, int n, const f32x4& acc, int, auto) {
                        float* pw = Wbi + m * ldp + n; *(f32x4*)pw = *(const f32x4*)pw + acc;
                    });
                } else {
                    wg_lmm_wide<true>(S, h0c, hi, zb, ldi, W1, ldc, [&](int m, int n, const f32x4& acc, int, auto) {
                        float* pa = ab0 + m * ldc + n; *(f32x4*)pa = *(const f32x4*)pa + acc;
                    });
                    wg_lmm_wide<false>(hi, h0c, S, zb, ldi, a0, ldc, [&](int m, int n, const f32x4& acc, int, auto) {
                        float* pw = Wb1 + m * ldc + n; *(f32x4*)pw = *(const f32x4*)pw + acc;
                    });
                }
                wg_lcolsum(S, hi, zb, ldi, [&](int n, float s_) { bbw[n] += s_; });
                wg_lds_barrier(); RSTAMP()
            }
            // ---- layer 0: z0bar -> Abar0 rows of the support set, b0bar, Dbar
            ew(h0c, [&](int m, int n) {
                float* pz = ab0 + m * ldc + n;
                const f32x4 z = relu_bwd4(*(const f32x4*)(a0 + m * ldc + n), *(const f32x4*)pz);
                *(f32x4*)pz = z;
                float* pA = A0bs + m * ldc + n; *(f32x4*)pA = *(const f32x4*)pA + z;
            });
            wg_lds_barrier();
            wg_lcolsum(S, h0c, ab0, ldc, [&](int n, float s_) { cs[n] = s_; b0b[n] += s_; });
            wg_lds_barrier();
            wg_lmm_wide<true>(S, h0c, S, G, ldS, ab0, ldc, [&](int m, int n, const f32x4& acc, int, auto) {
                float* pd = Db + m * ldc + n; *(f32x4*)pd = *(const f32x4*)pd - alpha * (acc + *(const f32x4*)(cs + n));
            });
            wg_lds_barrier(); RSTAMP()
        }
    }
    // ---- outputs: this part's columns of the layer-0 / layer-1 adjoints; part 0 also the unsplit ones
    store_img(w.Wb[1] + (long)b * h1 * h0 + c0, h0, Wb1, ldc, h1, h0c);
    store_img(w.A0bar + (long)b * w.ldsr * h0 + c0, h0, A0bs, ldc, S, h0c);
    for (int n = tid; n < h0c; n += nt) w.b0b[(long)b * h0 + c0 + n] = b0b[n];
    if (c == 0) {
        for (int i = 1; i < L; ++i) {
            const float* bbi = sm + y.bb[i];
            for (int n = tid; n < d.h[i]; n += nt) w.bb[i][(long)b * d.h[i] + n] = bbi[n];
            if (i >= 2) store_img(w.Wb[i] + (long)b * d.h[i] * d.h[i - 1], d.h[i - 1], sm + y.Wb[i], ldh(i - 1), d.h[i], d.h[i - 1]);
        }
        // d loss_b / d head_b = [Whbar | bhbar]
        float* hb = head_bar + (long)b * N * (H + 1);
        for (int j = tid; j < N * H; j += nt) { const int n = j / H, k = j - n * H; hb[n * (H + 1) + k] = Whb[n * ldH + k]; }
        for (int j = tid; j < N; j += nt) hb[j * (H + 1) + H] = bhb[j];
    }
    RSTAMP()
#undef RSTAMP
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
    w.apart = c.take(B * 2 * 8 * S * (L > 1 ? (size_t)p.h[1] : 1)); w.acnt = nullptr;       // (acnt: persistent, set by run_episodes)
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
        w.xpart = c.take(B * 2 * 8 * S * (L > 1 ? (size_t)p.h[1] : 1)); w.xcnt = (int*)c.take(B);
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

void set_xpanel_trace(void* p);                    // xpanel.hip
void set_rn12_trace(void* p);                      // rn12_conv.hip
extern "C" int fumi_hip_set_trace_buffer(int which, void* p) {
    if (which == 0) g_epi_trace = (unsigned long long*)p;
    else if (which == 1) set_xpanel_trace(p);
    else if (which == 2) set_rn12_trace(p);
    else return FUMI_EINVAL;
    return FUMI_OK;
}
extern "C" int fumi_hip_set_spin_limit(int polls) { const int old = g_spin_limit; g_spin_limit = polls < 0 ? 0 : polls; return old; }

size_t episode_workspace_bytes(const EpisodeProblem& p) {
    Carver c{nullptr, 0};
    EpiBuf w;
    w.trace = nullptr;
    carve(c, p, w);
    if (p.need_grad) {
        int kc;
        size_t ns = (size_t)xpanel_bwd_nsplit(p.B, p.S, p.Qn, p.D, p.h[0], &kc);
        if (xpanel_bwd_two_part_ok(p.D, p.h[0])) {       // (the two-launch form has more, shorter slabs)
            int nsq, kcq, nss, kcs;
            xpanel_bwd_two_part_split(p.B, p.S, p.Qn, p.D, p.h[0], &nsq, &kcq, &nss, &kcs);
            ns = std::max(ns, (size_t)(nsq + nss));
        }
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
    w.status = ws->status; w.spin_limit = g_spin_limit;
    w.acnt = ws->acnt;                           // persistent arrival counters of the split adapt kernel (zero between steps)
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
    // The backward X-panel pass in two launches (T >= 2): its query-row part beside the reverse sweep on the workspace's second
    // stream (the sweep is a chain of T x ~30 us on P workgroups per episode -- 128 of the 256 CUs at the reference sizes --, the pass
    // fills the others and is done before the sweep), its support-row part behind the sweep.  The adjoint array is then laid out
    // split ([B,S,h0] | [B,Qn,h0]) so that each part reads one contiguous panel.  FUMI_EPI_OVERLAP=0: one launch behind the sweep (2: the two-launch form at any size, for tests).
    static const int ovl_env = getenv("FUMI_EPI_OVERLAP") ? atoi(getenv("FUMI_EPI_OVERLAP")) : 1;
    // (measured: FuMI BERT T = 5, 32 episodes 0.436 -> 0.420 ms per step; a 4-episode MAML step, whose pass is 23 us, LOSES 18 us to
    // the fork / join and the extra launch -- only meta-batches whose pass is long enough to be worth hiding: >= 2048 query rows.
    // At T = 1 the sweep is one 42 us launch on half of the CUs and the query-row part takes the other half for about as long: the
    // headline step 0.2200 -> 0.2153 ms without phase timing; not adopted -- the pass is the bench's roofline kernel, timed as ONE
    // launch on the caller's stream, and +2 % does not pay for a second population of that kernel in every profile)
    const bool two_part = ovl_env && p.need_grad && p.T >= 2 && ws->side && ws->evx[0] && ws->evx[1] && !ws->profiling && !p.after_reverse &&
                          xpanel_bwd_two_part_ok(p.D, h0) && p.second_order && ((long)p.B * p.Qn >= 2048 || ovl_env == 2);      // (2: tests)
    if (p.need_grad) {
        w.ldsr = two_part ? p.S : p.S + p.Qn; w.ldq = two_part ? p.Qn : p.S + p.Qn;
        w.A0bar_q = two_part ? w.A0bar + (size_t)p.B * p.S * h0 : w.A0bar + (size_t)p.S * h0;
    } else { w.A0bar_q = nullptr; w.ldsr = w.ldq = 0; }
    float* bwd_slabs = nullptr; int nsq = 0, kcq = 0, nss = 0, kcs = 0;

    // ---- shared pass 1 over X: [A0 | G] = [Xs;Xq] [W0;Xs]^T for every episode, one launch (xpanel.hip)
    if (p.inputs_ready) HIP_TRY(hipEventRecord(p.inputs_ready, st));
    {
        ProfScope ps(ws, st, FUMI_PH_XPANEL_FWD);
        int rider_done = 0;
        if ((rc = launch_xpanel_fwd(st, p.B, p.S, p.Qn, p.D, h0, p.x_s, p.x_q, p.W[0], w.A0, w.G, p.rows.table ? &p.rows : nullptr,
                                    p.fwd_rider, &rider_done, nullptr, nullptr, xpanel_planes(ws, p.B, p.S, p.D, h0), p.glove))) return rc;
        if (p.fwd_rider && !rider_done && p.fwd_rider_fallback && (rc = p.fwd_rider_fallback(p.hook_ctx))) return rc;
    }
    if (p.after_xpanel_fwd && (rc = p.after_xpanel_fwd(p.hook_ctx))) return rc;
    // ---- per-episode phases
    if (p.head_ready) HIP_TRY(hipStreamWaitEvent(st, p.head_ready, 0));       // `head` was produced on another stream
    EpiParams prm;
    for (int i = 0; i < MAXL; ++i) { prm.W[i] = i < p.L ? p.W[i] : nullptr; prm.b[i] = i < p.L ? p.b[i] : nullptr; }
    // one inner step on a two-layer network with at most 32 support rows: every query tile runs the inner step itself
    // (query_lds_kernel<true>), no adapt launch
    static const int fuse_env = getenv("FUMI_EPI_FUSE") ? atoi(getenv("FUMI_EPI_FUSE")) : 1;
    QLay qlf; StageTab tbq, tbs;
    bool fuse_q = fuse_env && !getenv("FUMI_EPI_GLOBAL") && p.T == 1 && p.L == 2 && p.S <= QR && (h0 & 3) == 0 && p.head != nullptr;
    if (fuse_q) {
        query_layout(qlf, p.L, p.h, p.S, p.N, true);
        fuse_q = qlf.total <= QLDS_CAP;
    }
    if (fuse_q) {
        const long R = p.S + p.Qn, S = p.S, N = p.N, H = d.H;
        const int h1 = p.h[1];
        tbs.init(); tbq.init();
        tbs.add(w.A0, R * h0, 0, 0, h0, p.S, p.S, h0, qlf.a[0], wg_ld(h0));                     // support rows of A0
        tbs.add(p.W[1], 0, 0, 0, h0, h1, h1, h0, qlf.W[1], wg_ld(h0));                          // meta-parameters of layer 1
        tbs.add(p.b[1], 0, 0, 0, h1, 1, 1, h1, qlf.bi[1], h1);
        tbs.add(p.head, N * (H + 1), 0, 0, H + 1, p.N, p.N, (int)H, qlf.Wh, wg_ld((int)H));    // the episode's head [Wh | bh]
        tbs.add(p.head + H, N * (H + 1), 0, 0, H + 1, p.N, p.N, 1, qlf.bh, 1);
        tbs.add(p.b[0], 0, 0, 0, h0, 1, 1, h0, qlf.b0, h0);
        tbs.add(w.G + S * S, R * S, (long)QR * S, 0, S, -1, QR, p.S, qlf.Gq, wg_ld(p.S));      // the tile's rows of G (its A0 rows: direct loads)
        fuse_q = !tbs.bad && tbs.nunits <= 64 * 8 && h0 <= 256;
    }
    if (!fuse_q) {
        ProfScope ps(ws, st, FUMI_PH_ADAPT);
        static const bool force_global_a = getenv("FUMI_EPI_GLOBAL") != nullptr;   // dev/test: take the generic kernels
        // column parts per episode (adapt_lds_kernel): the layer-0 columns can be split over AP workgroups that exchange the
        // layer-1 partial sums once per inner step.  OFF by default: measured at the reference sizes (h0 = 256, AP = 4) the
        // exchange costs what the smaller products save (phase trace: layer 1 + exchange 3.2 -> 6.0 us, the two backward
        // products 10.3 -> 7.6 us; 0.3118 vs 0.3121 ms per step at T = 1, 0.515 vs 0.506 at T = 5): the inner loop is bound
        // by per-phase latency, not by the CU's matrix rate.  FUMI_ADAPT_P=n enables it (wider first layers).
        static const int ap_env = getenv("FUMI_ADAPT_P") ? atoi(getenv("FUMI_ADAPT_P")) : 1;
        int AP = 1;
        if (ap_env > 1 && p.L >= 2 && w.acnt && p.B <= FUMI_ACNT) {
            AP = ap_env > 8 ? 8 : ap_env;
            while (AP > 1 && (h0 % (64 * AP) != 0)) AP >>= 1;            // every part: a multiple of 64 columns
        }
        int hs[MAXL];
        for (int i = 0; i < MAXL; ++i) hs[i] = i < p.L ? p.h[i] : 0;
        hs[0] = h0 / AP;
        ALay al; adapt_layout(al, p.L, hs, p.S, p.N);
        const int H = d.H;
        bool lds_form = al.total <= ALDS_CAP && 4 + 2 * (p.L - 1) <= WG_MAXJOB && !force_global_a &&
                        ((p.S + 15) / 16) * ((hs[0] + 63) / 64) <= 8;      // one layer-0 block per wave (A0s + b0 in registers)
        StageTab tb; tb.init();
        if (lds_form) {
            const long R = p.S + p.Qn, S = p.S, N = p.N;
            tb.add(w.G, R * S, 0, 0, S, p.S, p.S, p.S, al.G, wg_ld(p.S));
            for (int i = 1; i < p.L; ++i) {
                // layer 1: this part's columns of every row (part stride hs[0]); deeper layers whole
                const int kc = i == 1 ? hs[0] : p.h[i - 1];
                tb.add(p.W[i], 0, 0, i == 1 ? hs[0] : 0, p.h[i - 1], p.h[i], p.h[i], kc, al.W[i], wg_ld(kc));
                tb.add(p.b[i], 0, 0, 0, p.h[i], 1, 1, p.h[i], al.bi[i], p.h[i]);
            }
            tb.add(p.head, N * (H + 1), 0, 0, H + 1, p.N, p.N, H, al.Wh, wg_ld(H));
            tb.add(p.head + H, N * (H + 1), 0, 0, H + 1, p.N, p.N, 1, al.bh, 1);
            lds_form = !tb.bad && tb.nunits <= 64 * 8;
        }
        if (lds_form) {
            FUMI_SET_DYN_LDS(adapt_lds_kernel, al.total * 4);
            const unsigned grid = AP > 1 ? 8u * ((p.B + 7) / 8) * AP : (unsigned)p.B;
            hipLaunchKernelGGL(adapt_lds_kernel, dim3(grid), dim3(512), al.total * 4, st, tb, d, w, al, prm, p.y_s, ws->status, AP);
        } else {
            FUMI_SET_DYN_LDS(adapt_kernel, w.lds_adapt * 4);
            hipLaunchKernelGGL(adapt_kernel, dim3(p.B), dim3(512), w.lds_adapt * 4, st, d, w, prm, p.y_s, p.head, ws->status);
        }
        LAUNCH_CHECK();
    }
    if (fuse_q) {
        ProfScope ps(ws, st, FUMI_PH_QUERY);
        FUMI_SET_DYN_LDS(query_lds_kernel<true>, qlf.total * 4);
        hipLaunchKernelGGL(query_lds_kernel<true>, dim3(w.ntile, p.B), dim3(512), qlf.total * 4, st, tbq, tbs, d, w, qlf, p.y_q,
                           p.logits_q, p.preds_q, p.preds_f, ws->status, p.y_s);
        LAUNCH_CHECK();
    } else {
        ProfScope ps(ws, st, FUMI_PH_QUERY);
        QLay ql; query_layout(ql, p.L, p.h, p.S, p.N);
        static const bool force_global = getenv("FUMI_EPI_GLOBAL") != nullptr;     // dev/test: take the generic kernels
        bool lds_form = ql.total <= QLDS_CAP && 7 + 2 * (p.L - 1) <= WG_MAXJOB && !force_global;
        StageTab tb; tb.init();
        if (lds_form) {
            // staging plan: source = base + episode*sb + tile*st
            const long R = p.S + p.Qn, S = p.S, N = p.N, H = d.H;
            const long slot = d.taped ? p.T : 0;
            tb.add(w.A0 + S * h0, R * h0, (long)QR * h0, 0, h0, -1, QR, h0, ql.a[0], wg_ld(h0));
            tb.add(w.D, S * h0, 0, 0, h0, p.S, p.S, h0, ql.D, wg_ld(h0));
            for (int i = 1; i < p.L; ++i) {
                const long sz = (long)p.h[i] * p.h[i - 1];
                tb.add(w.Wslot[i] + slot * sz, w.nslot * sz, 0, 0, p.h[i - 1], p.h[i], p.h[i], p.h[i - 1], ql.W[i], wg_ld(p.h[i - 1]));
            }
            tb.add(w.G + S * S, R * S, (long)QR * S, 0, S, -1, QR, p.S, ql.Gq, wg_ld(p.S));
            tb.add(w.Whslot + slot * N * H, w.nslot * N * H, 0, 0, H, p.N, p.N, (int)H, ql.Wh, wg_ld((int)H));
            tb.add(p.b[0], 0, 0, 0, h0, 1, 1, h0, ql.b0, h0);
            tb.add(w.cs, h0, 0, 0, h0, 1, 1, h0, ql.cs, h0);
            for (int i = 1; i < p.L; ++i) tb.add(w.bcur[i], p.h[i], 0, 0, p.h[i], 1, 1, p.h[i], ql.bi[i], p.h[i]);
            tb.add(w.bh, N, 0, 0, N, 1, 1, p.N, ql.bh, p.N);
            lds_form = !tb.bad && tb.nunits <= 64 * 8;      // wg_stage_rows: one unit per lane and wave
        }
        if (lds_form) {
            StageTab none; none.init();
            FUMI_SET_DYN_LDS(query_lds_kernel<false>, ql.total * 4);
            hipLaunchKernelGGL(query_lds_kernel<false>, dim3(w.ntile, p.B), dim3(512), ql.total * 4, st, tb, none, d, w, ql, p.y_q,
                               p.logits_q, p.preds_q, p.preds_f, ws->status, p.y_s);
        } else {
            FUMI_SET_DYN_LDS(query_kernel, w.lds_query * 4);
            hipLaunchKernelGGL(query_kernel, dim3(w.ntile, p.B), dim3(512), w.lds_query * 4, st, d, w, p.b[0], p.y_q, p.logits_q,
                               p.preds_q, p.preds_f, ws->status);
        }
        LAUNCH_CHECK();
    }
    if (two_part) {
        // query-row part of gW0 = Abar0^T X on the second stream, behind the query pass (whose adjoint rows it reads)
        xpanel_bwd_two_part_split(p.B, p.S, p.Qn, p.D, h0, &nsq, &kcq, &nss, &kcs);
        const long slab = (long)h0 * p.D;
        bwd_slabs = ws_f(ws, (size_t)(nsq + nss) * slab);
        HIP_TRY(hipEventRecord(ws->evx[0], st));
        HIP_TRY(hipStreamWaitEvent(ws->side, ws->evx[0], 0));
        if ((rc = launch_xpanel_bwd(ws->side, p.B, 0, p.Qn, p.D, h0, nullptr, p.x_q, w.A0bar_q, bwd_slabs, kcq, nsq,
                                    p.rows.table ? &p.rows : nullptr, nullptr, nullptr))) return rc;
        HIP_TRY(hipEventRecord(ws->evx[1], ws->side));
    }
    {
        ProfScope ps(ws, st, FUMI_PH_REVERSE);
        static const bool force_global_r = getenv("FUMI_EPI_GLOBAL") != nullptr;   // dev/test: take the generic kernels
        bool lds_form = p.L >= 2 && p.L <= 4 && w.ntile <= 8 && !force_global_r;
        int P = 0; RLay rl; StageTab ti, tsx;
        if (lds_form) {
            for (int cand = 1; cand <= 8 && !P; cand *= 2) {
                if (h0 % (4 * cand)) break;
                reverse_layout(rl, p.L, p.h, p.S, p.N, cand);
                if (rl.total <= RLDS_CAP) P = cand;
            }
            lds_form = P > 0;
        }
        if (lds_form) {
            const long S = p.S, N = p.N, H = d.H, R = p.S + p.Qn, nt = w.ntile, h0c = h0 / P, h1 = p.h[1];
            ti.init(); tsx.init();
            // adjoints after the query pass: sums over the tiles' partial slabs (+ G_ss)
            ti.add(w.pW[1], nt * h1 * h0, 0, h0c, h0, (int)h1, (int)h1, (int)h0c, rl.Wb[1], wg_ld((int)h0c), (int)nt, h1 * h0);
            ti.add(w.pD, nt * S * h0, 0, h0c, h0, p.S, p.S, (int)h0c, rl.Db, wg_ld((int)h0c), (int)nt, S * h0);
            ti.add(w.pb0, nt * h0, 0, h0c, h0, 1, 1, (int)h0c, rl.b0b, (int)h0c, (int)nt, h0);
            ti.add(w.pWh, nt * N * H, 0, 0, H, p.N, p.N, (int)H, rl.Whb, wg_ld((int)H), (int)nt, N * H);
            ti.add(w.pbh, nt * N, 0, 0, N, 1, 1, p.N, rl.bhb, p.N, (int)nt, N);
            for (int i = 1; i < p.L; ++i) {
                const long hi = p.h[i], hp = p.h[i - 1];
                ti.add(w.pb[i], nt * hi, 0, 0, hi, 1, 1, (int)hi, rl.bb[i], (int)hi, (int)nt, hi);
                if (i >= 2) ti.add(w.pW[i], nt * hi * hp, 0, 0, hp, (int)hi, (int)hi, (int)hp, rl.Wb[i], wg_ld((int)hp), (int)nt, hi * hp);
            }
            ti.add(w.G, R * S, 0, 0, S, p.S, p.S, p.S, rl.G, wg_ld(p.S));
            // the tape of step t (t rides in the "tile" slot of the plan)
            tsx.add(w.ta[0], w.ntape * S * h0, S * h0, h0c, h0, p.S, p.S, (int)h0c, rl.a[0], wg_ld((int)h0c));
            tsx.add(w.Wslot[1], w.nslot * h1 * h0, h1 * h0, h0c, h0, (int)h1, (int)h1, (int)h0c, rl.W[1], wg_ld((int)h0c));
            for (int i = 1; i < p.L; ++i) {
                const long hi = p.h[i], hp = p.h[i - 1];
                tsx.add(w.ta[i], w.ntape * S * hi, S * hi, 0, hi, p.S, p.S, (int)hi, rl.a[i], wg_ld((int)hi));
                tsx.add(w.tdz[i], w.ntape * S * hi, S * hi, 0, hi, p.S, p.S, (int)hi, rl.dz[i], wg_ld((int)hi));
                if (i >= 2) tsx.add(w.Wslot[i], w.nslot * hi * hp, hi * hp, 0, hp, (int)hi, (int)hi, (int)hp, rl.W[i], wg_ld((int)hp));
            }
            tsx.add(w.Whslot, w.nslot * N * H, N * H, 0, H, p.N, p.N, (int)H, rl.Wh, wg_ld((int)H));
            tsx.add(w.tp, w.ntape * S * N, S * N, 0, N, p.S, p.S, p.N, rl.p, wg_ld(p.N));
            tsx.add(w.te, w.ntape * S * N, S * N, 0, N, p.S, p.S, p.N, rl.e, wg_ld(p.N));
            lds_form = !ti.bad && !tsx.bad && ti.nunits <= 64 * 8 && tsx.nunits <= 64 * 8;
        }
        if (lds_form) {
            FUMI_SET_DYN_LDS(reverse_lds_kernel, rl.total * 4);
            hipLaunchKernelGGL(reverse_lds_kernel, dim3(8 * ((p.B + 7) / 8) * P), dim3(512), rl.total * 4, st, ti, tsx, d, w, rl, P,
                               p.loss_b, p.acc_b, p.head_bar);
        } else {
            FUMI_SET_DYN_LDS(reverse_kernel, w.lds_reverse * 4);
            hipLaunchKernelGGL(reverse_kernel, dim3(p.B), dim3(512), w.lds_reverse * 4, st, d, w, p.loss_b, p.acc_b, p.head_bar);
        }
        LAUNCH_CHECK();
    }
    if (p.after_reverse) HIP_TRY(hipEventRecord(p.after_reverse, st));        // head_bar is complete
    // ---- sums over episodes in one launch: meta-gradients of the hidden layers, layer-0 bias, loss/accuracy totals
    {
        ProfScope pr(ws, st, FUMI_PH_REDUCE);
        ReduceSegs own; own.n = 0; own.scale = p.grad_scale;
        const bool defer = p.defer_reduce && p.need_grad && p.defer_reduce->n + 4 + 2 * p.L <= 24;
        ReduceSegs& sg = defer ? *p.defer_reduce : own;
        if (defer) sg.scale = p.grad_scale;
        if (p.stats) { sg.add(p.loss_b, p.B, 1, 1, p.stats); sg.add(p.acc_b, p.B, 1, 1, p.stats + 1); }
        if (p.need_grad) {
            for (int i = 1; i < p.L; ++i) {
                const long sz = (long)p.h[i] * p.h[i - 1];
                sg.add(w.Wb[i], p.B, sz, sz, p.gW[i]);
                sg.add(w.bb[i], p.B, p.h[i], p.h[i], p.gb[i]);
            }
            sg.add(w.b0b, p.B, h0, h0, p.gb[0]);
        }
        if (!defer && (rc = launch_reduce_multi(st, own))) return rc;
    }
    if (!p.need_grad) return FUMI_OK;
    // ---- shared pass 2 over X: gW0 = Abar0^T [Xs;Xq], contraction over all B*R rows split into slabs (xpanel.hip)
    {
        ProfScope pg(ws, st, FUMI_PH_XPANEL_BWD);
        int kc;
        int ns = xpanel_bwd_nsplit(p.B, p.S, p.Qn, p.D, h0, &kc);
        const long slab = (long)h0 * p.D;
        float* slabs = two_part ? bwd_slabs : ws_f(ws, (size_t)ns * slab);
        int rider_done = 0;
        if (two_part) {
            // support-row part behind the sweep (with the hypernetwork backward as its rider), then the second stream's part joins
            if ((rc = launch_xpanel_bwd(st, p.B, p.S, 0, p.D, h0, p.x_s, nullptr, w.A0bar, bwd_slabs + (size_t)nsq * slab, kcs, nss,
                                        p.rows.table ? &p.rows : nullptr, p.bwd_rider, &rider_done))) return rc;
            HIP_TRY(hipStreamWaitEvent(st, ws->evx[1], 0));
            ns = nsq + nss;
        } else
        if ((rc = launch_xpanel_bwd(st, p.B, p.S, p.Qn, p.D, h0, p.x_s, p.x_q, w.A0bar, slabs, kc, ns, p.rows.table ? &p.rows : nullptr,
                                    p.bwd_rider, &rider_done))) return rc;
        if (p.bwd_rider && !rider_done && p.bwd_rider_fallback && (rc = p.bwd_rider_fallback(p.hook_ctx2))) return rc;
        if (p.defer_reduce && p.defer_reduce->n < 24 && p.defer_reduce->scale == p.grad_scale) p.defer_reduce->add(slabs, ns, slab, slab, p.gW[0]);
        else if ((rc = launch_reduce_slabs(st, slabs, ns, slab, slab, p.grad_scale, p.gW[0]))) return rc;
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
