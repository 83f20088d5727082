// AM3 step: prototypical network with text-gated convex combination (no inner loop).
//
// Replaces AM3.forward / AM3.evaluate (fumi/models/am3.py:90-126,160-200, dropout 0) and the head math of
// fumi/utils/utils.py: get_num_samples :379-387, get_prototypes :331-376, prototypical_loss :390-402, get_preds :302-328.
//
// The image encoder (one shared Linear on all B*(S+Qn) rows) and its weight gradient run on the X-panel kernels of the FuMI
// path (xpanel.hip: one pass over [Xs_b;Xq_b] forward, one backward, rows never copied); the g / h text MLPs on B*S rows and
// their backward run on the GEMM family (gemm.hip), ReLU masks fused into the epilogues, every bias gradient in one batched
// column-sum.  The per-episode head is one fused kernel (am3_head_kernel, 1 workgroup / episode):
//   * per-class prototype sums as wavefront-level segmented reductions: one wave per class, lanes over the prototype
//     dim, the class test is wave-uniform (no scatter atomics, unlike the reference's scatter_add_)
//   * squared distances query x class with one wave per query row (lanes over P, shuffle reduction), softmax-CE over
//     the N classes, first arg-min, and the backward to every embedding in the same pass (per-wave slabs in LDS for the
//     prototype adjoints: deterministic, no float atomics)
#include "common.h"
#include <optional>
#include <stdlib.h>
#include "hyper_fwd.h"
#include "hyper_bwd.h"

namespace {

// Wave-wide reductions on the DPP path (quad swaps, row shifts by 4 and 8, row_bcast15 / 31: the total lands in lane 63 and is read
// back as a scalar).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS round trip per step; the head kernel runs ~70 of them per
// query row.
template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ float dpp0(float x) {            // 0 where the pattern has no source
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWS, 0xf, true));
}
template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ float dpps(float x) {            // own value where it has none
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_sum63(float v) {                                          // total in lane 63 only
    v += dpp0<0xB1>(v); v += dpp0<0x4E>(v); v += dpp0<0x114>(v); v += dpp0<0x118>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum63(v)), 63));
}

// LDS layout (floats): ip[N*P] tp[N*P] pr[N*P] pb[nw][N*P] lamc[N] cnt[N] lbar[N] wl[nw] wc[nw]
__global__ __launch_bounds__(256) void am3_head_generic_kernel(int N, int S, int Qn, int P, int lamda_fixed, int need_grad,
                                                       float dscale,
                                                       const float* __restrict__ im_s, const float* __restrict__ tx,
                                                       float* __restrict__ lam_s, const int64_t* __restrict__ y_s,
                                                       const float* __restrict__ im_q, const int64_t* __restrict__ y_q,
                                                       int64_t* __restrict__ preds, float* __restrict__ loss_b,
                                                       float* __restrict__ corr_b, float* __restrict__ conf_b,
                                                       float* __restrict__ lam_b, float lscale,
                                                       float* __restrict__ im_s_bar, float* __restrict__ tx_bar,
                                                       float* __restrict__ zl_bar, float* __restrict__ im_q_bar,
                                                       long im_stride, const float* __restrict__ im_bias, int* status) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int NP = N * P;
    float* ip = sm; float* tp = ip + NP; float* pr = tp + NP; float* pb = pr + NP;
    float* lamc = pb + nw * NP; float* cnt = lamc + N; float* lbar = cnt + N; float* wl = lbar + N; float* wc = wl + nw;
    im_s += (long)b * im_stride; tx += (long)b * S * P; lam_s += (long)b * S; y_s += (long)b * S;
    im_q += (long)b * im_stride; y_q += (long)b * Qn; preds += (long)b * Qn;      // (im_bias: added by the caller for this kernel)

    if (lamda_fixed >= 0) for (int s = tid; s < S; s += blockDim.x) lam_s[s] = (float)lamda_fixed;   // am3.py:174-177
    __syncthreads();
    // ---- prototypes: wave per class, lanes over P; count clamped to >= 1 (utils.py:353-355)
    for (int c = wave; c < N; c += nw) {
        float n = 0.f, ls = 0.f;
        for (int s = 0; s < S; ++s) {
            const long y = y_s[s];
            if (y < 0 || y >= N) { if (lane == 0) atomicOr(status, FUMI_ST_LABEL_RANGE); continue; }
            if (y == c) { n += 1.f; ls += lam_s[s]; }
        }
        const float nn = fmaxf(n, 1.f);
        for (int j = lane; j < P; j += 64) {
            float si = 0.f, st = 0.f;
            for (int s = 0; s < S; ++s)
                if (y_s[s] == c) { si += im_s[(long)s * P + j]; st += tx[(long)s * P + j]; }
            si /= nn; st /= nn;
            const float lc = ls / nn;
            ip[c * P + j] = si; tp[c * P + j] = st;
            pr[c * P + j] = lc * si + (1.f - lc) * st;
        }
        if (lane == 0) { lamc[c] = ls / nn; cnt[c] = nn; }
    }
    for (int i = tid; i < nw * NP; i += blockDim.x) pb[i] = 0.f;
    __syncthreads();

    // ---- queries: wave per row; d[c] = |proto_c - x|^2 ; CE over -d ; first arg-min ; backward
    float lsum = 0.f, csum = 0.f;
    float* mypb = pb + wave * NP;
    for (int q = wave; q < Qn; q += nw) {
        long yq = y_q[q];
        if (yq < 0 || yq >= N) { if (lane == 0) atomicOr(status, FUMI_ST_LABEL_RANGE); yq = 0; }
        // pass 1: distances (kept for the <= 64 classes handled per register group; general N re-computes)
        float dmin = INFINITY; int amin = 0; float mx = -INFINITY;
        for (int c = 0; c < N; ++c) {
            float v = 0.f;
            for (int j = lane; j < P; j += 64) { const float df = pr[c * P + j] - im_q[(long)q * P + j]; v += df * df; }
            v = wave_sum(v);
            if (v < dmin) { dmin = v; amin = c; }
            mx = fmaxf(mx, -v);
        }
        float se = 0.f, dy = 0.f;
        for (int c = 0; c < N; ++c) {
            float v = 0.f;
            for (int j = lane; j < P; j += 64) { const float df = pr[c * P + j] - im_q[(long)q * P + j]; v += df * df; }
            v = wave_sum(v);
            se += expf(-v - mx);
            if (c == yq) dy = v;
        }
        const float lse = mx + logf(se);
        lsum += lse + dy;                                   // -log softmax(-d)[y] = lse - (-d_y)
        csum += (amin == yq) ? 1.f : 0.f;
        if (lane == 0) preds[q] = amin;
        if (lane == 0 && conf_b) atomicAdd(conf_b + ((long)b * N + yq) * N + amin, 1.f);   // integer-valued: exact in any order
        if (need_grad) {
            // dbar[c] = dL/dd[c] = -(p_c - onehot_c) * dscale ;  xbar = sum_c dbar[c] * (-2)(proto_c - x) ;  pbar_c += dbar[c]*2(proto_c - x)
            for (int j0 = 0; j0 < P; j0 += 64) {
                const int j = j0 + lane;
                float xb = 0.f;
                const float x = j < P ? im_q[(long)q * P + j] : 0.f;
                for (int c = 0; c < N; ++c) {
                    float v = 0.f;
                    for (int jj = lane; jj < P; jj += 64) { const float df = pr[c * P + jj] - im_q[(long)q * P + jj]; v += df * df; }
                    v = wave_sum(v);
                    const float pc = expf(-v - lse);
                    const float db = -(pc - (c == yq ? 1.f : 0.f)) * dscale;
                    if (j < P) {
                        const float df2 = 2.f * (pr[c * P + j] - x);
                        xb -= db * df2;
                        mypb[c * P + j] += db * df2;
                    }
                }
                if (j < P) im_q_bar[(long)b * im_stride + (long)q * P + j] = xb;
            }
        }
    }
    if (lane == 0) { wl[wave] = lsum; wc[wave] = csum; }
    __syncthreads();
    if (tid == 0) {
        float l = 0.f, c = 0.f;
        for (int w = 0; w < nw; ++w) { l += wl[w]; c += wc[w]; }
        loss_b[b] = l * dscale;
        corr_b[b] = c;
        if (lam_b) { float ls_ = 0.f; for (int s2 = 0; s2 < S; ++s2) ls_ += lam_s[s2]; lam_b[b] = ls_ * lscale; }
    }
    if (!need_grad) return;
    // ---- prototype adjoints -> per-sample gradients (utils.py:358-375 reversed)
    for (int i = tid; i < NP; i += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += pb[w * NP + i];
        pb[i] = s;                                          // slab 0 now holds pbar (only element i of slab 0 is touched by thread i)
    }
    __syncthreads();
    for (int c = wave; c < N; c += nw) {
        float v = 0.f;
        for (int j = lane; j < P; j += 64) v += pb[c * P + j] * (ip[c * P + j] - tp[c * P + j]);
        v = wave_sum(v);
        if (lane == 0) lbar[c] = v;
    }
    __syncthreads();
    im_s_bar += (long)b * im_stride; tx_bar += (long)b * S * P; zl_bar += (long)b * S;
    for (int i = tid; i < S * P; i += blockDim.x) {
        const int s = i / P, j = i % P;
        long c = y_s[s];
        if (c < 0 || c >= N) { im_s_bar[i] = 0.f; tx_bar[i] = 0.f; continue; }
        const float g = pb[c * P + j] / cnt[c];
        im_s_bar[i] = lamc[c] * g;
        tx_bar[i] = (1.f - lamc[c]) * g;
    }
    for (int s = tid; s < S; s += blockDim.x) {
        long c = y_s[s];
        float z = 0.f;
        if (lamda_fixed < 0 && c >= 0 && c < N) { const float l = lam_s[s]; z = lbar[c] / cnt[c] * l * (1.f - l); }
        zl_bar[s] = z;
    }
}


__global__ void am3_bias_rows_kernel(float* __restrict__ x, const float* __restrict__ bias, long n, int P) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] += bias[i % P];
}

// ---- the same head for N <= 64 classes and P <= 512 (every configuration of the reference): no global load inside a loop.
// Labels, support embeddings and text embeddings of the episode are staged in LDS once; a query row is loaded once into
// registers (next row prefetched), its N distances live one per LANE (lane c holds d_c), so soft-max, first arg-min and the
// backward weights are wave-level reductions / shuffles instead of N recomputations of every distance.
// LDS (floats): ip tp pr [N*P] | pb [nw][N*P] | ims txs [S*P] | lamc cnt lbar [N] | wl wc [nw] | ys [S] (ints) | lam [S]
constexpr int HPJ = 8;                   // 64-lane chunks of the prototype dimension held in registers (P <= 512)
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpps<0xB1>(v)); v = fmaxf(v, dpps<0x4E>(v)); v = fmaxf(v, dpps<0x114>(v)); v = fmaxf(v, dpps<0x118>(v));
    v = fmaxf(v, dpps<0x142, 0xa>(v)); v = fmaxf(v, dpps<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__global__ __launch_bounds__(1024) void am3_head_kernel(int N, int S, int Qn, int P, int lamda_fixed, int need_grad,
                                                       float dscale,
                                                       const float* __restrict__ im_s, const float* __restrict__ tx,
                                                       float* __restrict__ lam_s, const int64_t* __restrict__ y_s,
                                                       const float* __restrict__ im_q, const int64_t* __restrict__ y_q,
                                                       int64_t* __restrict__ preds, float* __restrict__ loss_b,
                                                       float* __restrict__ corr_b, float* __restrict__ conf_b,
                                                       float* __restrict__ lam_b, float lscale,
                                                       float* __restrict__ im_s_bar, float* __restrict__ tx_bar,
                                                       float* __restrict__ zl_bar, float* __restrict__ im_q_bar,
                                                       long im_stride, const float* __restrict__ im_bias, int* status,
                                                       int B, int GQ, float* part, int* arrive, int imparts, long impart_stride,
                                                       float* __restrict__ bias_bar) {
    // bias_bar [B][P] (optional, need_grad): the episode's column sums of all embedding adjoints (the image encoder's bias gradient),
    // so that no separate column-sum pass over imbar is needed.
    // imparts > 1: the image embeddings arrive as that many partial products (split contraction of the encoder pass, xpanel.hip),
    // im_s / im_q point at part 0 and the parts are added where a row is read.
    // GQ workgroups per episode (ids equal mod 8: one XCD): each forms the prototypes and takes a contiguous share of the query
    // rows; loss / correct counts / prototype adjoints of the shares meet in `part` ([B][GQ][N*P + 2], agent-scope stores) and
    // the workgroup that arrives last at the episode's counter adds them in share order (deterministic) and runs the support-
    // side epilogue.  One workgroup per episode walked 10 query rows per wave, each a chain of ~25 dependent cross-lane steps.
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, nt = blockDim.x, nw = blockDim.x >> 6;
    const int b = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / GQ), gq = (int)((blockIdx.x >> 3) % GQ);
    if (b >= B) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NP = N * P, SP = S * P;
    const int qper = (Qn + GQ - 1) / GQ, q0 = gq * qper, q1 = min(Qn, q0 + qper);
    float* ip = sm; float* tp = ip + NP; float* pr = tp + NP; float* pb = pr + NP;
    float* ims = pb + nw * NP; float* txs = ims + SP;
    float* lamc = txs + SP; float* cnt = lamc + N; float* lbar = cnt + N; float* wl = lbar + N; float* wc = wl + nw;
    int* ys = (int*)(wc + nw); float* lam = (float*)(ys + S);
    float* cf = lam + S;                                     // [N*N] this share's confusion counts (LDS adds: no global atomics, no memset)
    float* bw = cf + N * N;                                  // [nw][P] per-wave column sums of the query adjoints
    const int PS = NP + 2 + N * N + P;                       // floats of one share's record in `part`
    im_s += (long)b * im_stride; tx += (long)b * SP; lam_s += (long)b * S; y_s += (long)b * S;
    im_q += (long)b * im_stride; y_q += (long)b * Qn; preds += (long)b * Qn;

    // ---- stage the episode's support side: every load of the prologue (support + text embeddings, labels, lamda, the
    // wave's first query row) is issued before the first one is used
    const int npj = (P + 63) >> 6;
    float xn[HPJ], bq[HPJ];
#pragma unroll
    for (int k = 0; k < HPJ; ++k) { const int j = k * 64 + lane; bq[k] = (k < npj && j < P) ? im_bias[j] : 0.f; }
    auto qrow = [&](int q, int j) {                          // one element of query row q (all parts)
        float v = im_q[(long)q * P + j];
        for (int z = 1; z < imparts; ++z) v += im_q[z * impart_stride + (long)q * P + j];
        return v;
    };
#pragma unroll
    for (int k = 0; k < HPJ; ++k) { const int j = k * 64 + lane; xn[k] = (k < npj && j < P && q0 + wave < q1) ? qrow(q0 + wave, j) : 0.f; }
    long yn = q0 + wave < q1 ? y_q[q0 + wave] : 0;
    for (int i0 = tid; i0 < SP; i0 += 4 * nt) {
        float a[4], t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * nt, SP - 1);
            a[u] = im_s[i] + im_bias[i % P]; t[u] = tx[i];
            for (int z = 1; z < imparts; ++z) a[u] += im_s[z * impart_stride + i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nt; if (i < SP) { ims[i] = a[u]; txs[i] = t[u]; } }
    }
    for (int s_ = tid; s_ < S; s_ += nt) {
        long y = y_s[s_];
        if (y < 0 || y >= N) { atomicOr(status, FUMI_ST_LABEL_RANGE); y = -1; }
        ys[s_] = (int)y;
        float l = lamda_fixed >= 0 ? (float)lamda_fixed : lam_s[s_];
        if (lamda_fixed >= 0 && gq == 0) lam_s[s_] = l;                        // am3.py:174-177
        lam[s_] = l;
    }
    for (int i = tid; i < nw * NP; i += nt) pb[i] = 0.f;
    for (int i = tid; i < N * N; i += nt) cf[i] = 0.f;
    __syncthreads();
    // ---- prototypes: wave per class, lanes over P; count clamped to >= 1 (utils.py:353-355).  Membership is a weight, not a
    // branch: the LDS reads of all S rows are independent of each other
    for (int c = wave; c < N; c += nw) {
        float n = 0.f, ls = 0.f;
        for (int s_ = 0; s_ < S; ++s_) { const float m_ = ys[s_] == c ? 1.f : 0.f; n += m_; ls += m_ * lam[s_]; }
        const float nn = fmaxf(n, 1.f), lc = ls / nn;
        for (int j = lane; j < P; j += 64) {
            float si = 0.f, st = 0.f;
#pragma unroll 5
            for (int s_ = 0; s_ < S; ++s_) { const float m_ = ys[s_] == c ? 1.f : 0.f; si += m_ * ims[s_ * P + j]; st += m_ * txs[s_ * P + j]; }
            si /= nn; st /= nn;
            ip[c * P + j] = si; tp[c * P + j] = st;
            pr[c * P + j] = lc * si + (1.f - lc) * st;
        }
        if (lane == 0) { lamc[c] = lc; cnt[c] = nn; }
    }
    __syncthreads();

    // ---- queries: wave per row
    float lsum = 0.f, csum = 0.f;
    float* mypb = pb + wave * NP;
    float xbs[HPJ];
#pragma unroll
    for (int k = 0; k < HPJ; ++k) xbs[k] = 0.f;
    for (int q = q0 + wave; q < q1; q += nw) {
        float x[HPJ];
#pragma unroll
        for (int k = 0; k < HPJ; ++k) x[k] = xn[k] + bq[k];
        long yq = yn;
        const int qn = q + nw;                                                  // prefetch the wave's next row
#pragma unroll
        for (int k = 0; k < HPJ; ++k) { const int j = k * 64 + lane; xn[k] = (k < npj && j < P && qn < q1) ? qrow(qn, j) : 0.f; }
        yn = qn < q1 ? y_q[qn] : 0;
        if (yq < 0 || yq >= N) { if (lane == 0) atomicOr(status, FUMI_ST_LABEL_RANGE); yq = 0; }
        // lane c keeps d_c = |proto_c - x|^2; eight classes are reduced together (their butterflies interleave: one
        // dependent shuffle chain per class would cost ~600 cycles each)
        float myd = INFINITY;
        for (int c0 = 0; c0 < N; c0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = 0.f;
                const int c = min(c0 + u, N - 1);
#pragma unroll
                for (int k = 0; k < HPJ; ++k) { const int j = k * 64 + lane; if (k < npj && j < P) { const float df = pr[c * P + j] - x[k]; v[u] += df * df; } }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = wave_sum(v[u]);                    // (eight independent DPP chains: they interleave)
#pragma unroll
            for (int u = 0; u < 8; ++u) if (c0 + u < N && lane == c0 + u) myd = v[u];
        }
        const float dmin = -wave_max(-myd);                                      // lanes >= N hold +inf
        const unsigned long long at = __ballot(lane < N && myd == dmin);
        const int amin = at ? __ffsll((long long)at) - 1 : 0;                  // first arg-min (torch.min semantics, utils.py:316)
        const float mx = -dmin;
        const float ex = lane < N ? expf(-myd - mx) : 0.f;
        const float se = wave_sum(ex);
        const float lse = mx + logf(se);
        const float dy = __shfl(myd, (int)yq, 64);
        lsum += lse + dy;                                                       // -log softmax(-d)[y] = lse - (-d_y)
        csum += (amin == (int)yq) ? 1.f : 0.f;
        if (lane == 0) preds[q] = amin;
        if (lane == 0 && conf_b) atomicAdd(cf + yq * N + amin, 1.f);                       // integer-valued: exact in any order
        if (need_grad) {
            // dbar[c] = dL/dd[c] = -(p_c - onehot_c) * dscale ;  xbar = sum_c dbar[c] * (-2)(proto_c - x) ;  pbar_c += dbar[c]*2(proto_c - x)
            const float mydb = lane < N ? -(expf(-myd - lse) - (lane == (int)yq ? 1.f : 0.f)) * dscale : 0.f;
            float xb[HPJ];
#pragma unroll
            for (int k = 0; k < HPJ; ++k) xb[k] = 0.f;
            for (int c0 = 0; c0 < N; c0 += 8) {
                float dbv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) dbv[u] = __shfl(mydb, min(c0 + u, N - 1), 64);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u;
                    if (c < N) {
                        const float db = dbv[u];
#pragma unroll
                        for (int k = 0; k < HPJ; ++k) {
                            const int j = k * 64 + lane;
                            if (k < npj && j < P) {
                                const float df2 = 2.f * (pr[c * P + j] - x[k]);
                                xb[k] -= db * df2;
                                mypb[c * P + j] += db * df2;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < HPJ; ++k) { const int j = k * 64 + lane; if (k < npj && j < P) { im_q_bar[(long)b * im_stride + (long)q * P + j] = xb[k]; xbs[k] += xb[k]; } }
        }
    }
    if (lane == 0) { wl[wave] = lsum; wc[wave] = csum; }
    if (need_grad && bias_bar) {
#pragma unroll
        for (int k = 0; k < HPJ; ++k) { const int j = k * 64 + lane; if (k < npj && j < P) bw[wave * P + j] = xbs[k]; }
    }
    __syncthreads();
    // ---- this share's totals (waves in order), then the meeting of the shares
    float* mine = part + ((long)b * GQ + gq) * PS;
    if (tid == 0) {
        float l = 0.f, c = 0.f;
        for (int w_ = 0; w_ < nw; ++w_) { l += wl[w_]; c += wc[w_]; }
        if (GQ == 1) { wl[0] = l; wc[0] = c; }
        else {
            __hip_atomic_store(mine + NP, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + NP + 1, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (need_grad) {
        for (int i = tid; i < NP; i += nt) {
            float s_ = 0.f;
            for (int w_ = 0; w_ < nw; ++w_) s_ += pb[w_ * NP + i];
            if (GQ == 1) pb[i] = s_;                        // slab 0 now holds pbar (only element i of slab 0 is touched by thread i)
            else __hip_atomic_store(mine + i, s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (conf_b) {
        for (int i = tid; i < N * N; i += nt) {
            if (GQ == 1) conf_b[(long)b * N * N + i] = cf[i];
            else __hip_atomic_store(mine + NP + 2 + i, cf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (need_grad && bias_bar) {                             // query part of the bias gradient: waves in order
        for (int j = tid; j < P; j += nt) {
            float s_ = 0.f;
            for (int w_ = 0; w_ < nw; ++w_) s_ += bw[w_ * P + j];
            if (GQ == 1) bw[j] = s_;                         // (thread j only touches column j of wave 0's row)
            else __hip_atomic_store(mine + NP + 2 + N * N + j, s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    wg_drain_stores();                                      // every wave: its sc1 partial stores have completed ...
    __syncthreads();                                        // ... before lane 0 signals for all of them
    if (GQ > 1) {
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(arrive + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == GQ - 1;
            if (old == GQ - 1) __hip_atomic_store(arrive + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        if (!s_last) return;
        const float* all = part + (long)b * GQ * PS;
        if (tid == 0) {
            float l = 0.f, c = 0.f;
            for (int g_ = 0; g_ < GQ; ++g_) {
                l += __hip_atomic_load(all + (long)g_ * PS + NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c += __hip_atomic_load(all + (long)g_ * PS + NP + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wl[0] = l; wc[0] = c;
        }
        if (need_grad) {
            for (int i = tid; i < NP; i += nt) {
                float s_ = 0.f;
                for (int g_ = 0; g_ < GQ; ++g_) s_ += __hip_atomic_load(all + (long)g_ * PS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pb[i] = s_;
            }
        }
        if (conf_b) {
            for (int i = tid; i < N * N; i += nt) {
                float s_ = 0.f;
                for (int g_ = 0; g_ < GQ; ++g_) s_ += __hip_atomic_load(all + (long)g_ * PS + NP + 2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                conf_b[(long)b * N * N + i] = s_;
            }
        }
        if (need_grad && bias_bar) {
            for (int j = tid; j < P; j += nt) {
                float s_ = 0.f;
                for (int g_ = 0; g_ < GQ; ++g_) s_ += __hip_atomic_load(all + (long)g_ * PS + NP + 2 + N * N + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bw[j] = s_;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        loss_b[b] = wl[0] * dscale;
        corr_b[b] = wc[0];
        if (lam_b) { float ls_ = 0.f; for (int s2 = 0; s2 < S; ++s2) ls_ += lam[s2]; lam_b[b] = ls_ * lscale; }
    }
    if (!need_grad) return;
    // ---- prototype adjoints -> per-sample gradients (utils.py:358-375 reversed)
    for (int c = wave; c < N; c += nw) {
        float v = 0.f;
        for (int j = lane; j < P; j += 64) v += pb[c * P + j] * (ip[c * P + j] - tp[c * P + j]);
        v = wave_sum(v);
        if (lane == 0) lbar[c] = v;
    }
    __syncthreads();
    im_s_bar += (long)b * im_stride; tx_bar += (long)b * SP; zl_bar += (long)b * S;
    for (int i = tid; i < SP; i += nt) {
        const int s_ = i / P, j = i - s_ * P;
        const int c = ys[s_];
        if (c < 0) { im_s_bar[i] = 0.f; tx_bar[i] = 0.f; continue; }
        const float g = pb[c * P + j] / cnt[c];
        im_s_bar[i] = lamc[c] * g;
        tx_bar[i] = (1.f - lamc[c]) * g;
    }
    for (int s_ = tid; s_ < S; s_ += nt) {
        const int c = ys[s_];
        float z = 0.f;
        if (lamda_fixed < 0 && c >= 0) { const float l = lam[s_]; z = lbar[c] / cnt[c] * l * (1.f - l); }
        zl_bar[s_] = z;
    }
    if (bias_bar) {                                          // + the support rows' adjoints: row s adds lamc[c] pbar_c / cnt[c], c = y_s
        for (int j = tid; j < P; j += nt) {
            float s_ = bw[j];
            for (int r = 0; r < S; ++r) { const int c = ys[r]; if (c >= 0) s_ += lamc[c] * (pb[c * P + j] / cnt[c]); }
            bias_bar[(long)b * P + j] = s_;
        }
    }
}

}  // namespace

// dx_s [B,S,D] / dx_q [B,Qn,D] (optional, need_grad): adjoints of the image rows, imbar Wi -- what an image encoder in front of
// this step (the Conv4 backbone, fumi_hip_conv4_encode_bwd) continues from
static int am3_step_impl(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int Dt, int Ht, int P, int lamda_fixed, int need_grad, float grad_scale,
        float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q, const float* text_s,
        const float* const* w, float* loss, int64_t* preds_q, float* lamda_s, float* correct, float* const* g_w,
        float* stats, float* dx_s, float* dx_q) {
    if (!ws || !x_s || !y_s || !x_q || !y_q || !text_s || !w || !loss || !preds_q || !lamda_s || !correct) return FUMI_EINVAL;
    if (B < 1 || N < 1 || S < 1 || Qn < 1 || D < 1 || Dt < 1 || Ht < 1 || P < 1 || lamda_fixed < -1 || lamda_fixed > 1) return FUMI_EINVAL;
    for (int i = 0; i < 10; ++i) if (!w[i] || (need_grad && (!g_w || !g_w[i]))) return FUMI_EINVAL;
    if (dropout_p < 0.f || dropout_p >= 1.f) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    // train-mode Dropout after the ReLU of g and of h (am3.py:82,88): counter-based masks keyed by (seed, tag, element)
    unsigned thr = 0; float dsc = 1.f;
    if (dropout_p > 0.f) { thr = (unsigned)((double)dropout_p * 4294967296.0); if (!thr) thr = 1; dsc = 1.f / (1.f - dropout_p); }
    auto dkey = [&](unsigned tag) {
        auto mix = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
        return mix(mix((unsigned)(seed & 0xffffffffULL) ^ (0x9E3779B9U * tag)) ^ (unsigned)(seed >> 32));
    };
    const float *Wi = w[0], *bi = w[1], *G0 = w[2], *g0 = w[3], *G1 = w[4], *g1 = w[5], *H0 = w[6], *h0 = w[7], *H1 = w[8], *h1 = w[9];
    const long Rs = (long)B * S, Rq = (long)B * Qn;
    // fast head: N <= 64 classes, P <= 512, and the episode's support side fits LDS next to the per-wave adjoint slabs
    int nwaves = 16;
    auto fast_lds = [&](int nw_) { return ((size_t)(3 + nw_) * N * P + 2 * (size_t)S * P + 3 * N + 2 * nw_ + 2 * S + (size_t)N * N + (size_t)nw_ * P + 16) * sizeof(float); };
    bool fast_head = N <= 64 && P <= 64 * HPJ;
    if (fast_head && fast_lds(16) > 150 * 1024) nwaves = 8;
    if (fast_head && fast_lds(nwaves) > 150 * 1024) nwaves = 4;
    if (fast_head && fast_lds(nwaves) > 150 * 1024) fast_head = false;
    static const int head_generic = getenv("FUMI_AM3_GENERIC") ? atoi(getenv("FUMI_AM3_GENERIC")) : 0;
    if (head_generic) fast_head = false;
    if (!fast_head) nwaves = 4;
    const size_t lds = fast_head ? fast_lds(nwaves) : ((size_t)(3 + nwaves) * N * P + 3 * N + 2 * nwaves) * sizeof(float);
    if (lds > 160 * 1024) return FUMI_ENOTSUP;
    int xkc = 0;
    const int xns = xpanel_bwd_nsplit(B, S, Qn, D, P, &xkc);          // contraction slabs of gWi = sum_b imbar_b^T [Xs_b;Xq_b]

    size_t bytes = 0;
    auto A = [&](size_t n) { bytes += ws_align(n * sizeof(float)); };
    A((Rs + Rq) * P); A(Rs * Ht); A(Rs * P); A(Rs * Ht); A(2 * B); A((size_t)B * N * N + B);
    const size_t wslab_n = (size_t)((Rs + 127) / 128) * ((size_t)Ht + 2 * (size_t)Ht * P + (size_t)Ht * Dt) + 64;   // text weight-gradient slabs
    const size_t cpart_n = (size_t)((Rs + 127) / 128) * (size_t)(2 * ((Ht + 3) & ~3) + ((P + 3) & ~3) + 4)
                         + (size_t)((Rs + Rq + 127) / 128) * (size_t)((P + 3) & ~3) + 64;       // ColsumJobs partial sums
    if (need_grad) { A((Rs + Rq) * P); A(Rs * P); A(Rs); A(Rs * Ht); A(Rs * Ht); A((size_t)xns * P * D); A(cpart_n); A(wslab_n); }
    // the text MLPs g (text -> Ht -> P) and h (P -> Ht -> 1) on the hypernetwork kernels (hyper_fwd.h / hyper_bwd.h): both layers of
    // a forward in one launch, a whole backward in one launch of independent (row block, column chunk) workgroups
    static const int mlp_fused = getenv("FUMI_AM3_MLP") ? atoi(getenv("FUMI_AM3_MLP")) : 1;      // 0: one GEMM launch per product
    const size_t hfg_n = hyper_fwd_workspace_floats((int)Rs, Ht, P), hfh_n = hyper_fwd_workspace_floats((int)Rs, Ht, 1);
    const size_t hbh_n = hyper_bwd_fused_workspace_floats((int)Rs, P, Ht, 1), hbg_n = hyper_bwd_fused_workspace_floats((int)Rs, 0, Ht, P);
    const size_t txp_n = (need_grad && (Ht & 63) == 0) ? (size_t)((Rs + 15) / 16) * (Ht / 64) * 16 * P : 0;   // partials of txbar += l1bar H0
    if (mlp_fused) { A(hfg_n); A(hfh_n); if (need_grad) { A(hbh_n); A(hbg_n); A(txp_n); } }
    // query shares per episode of the head kernel: enough workgroups for the chip, >= 2 rows per wave, counters available
    static const int hgq_env = getenv("FUMI_AM3_GQ") ? atoi(getenv("FUMI_AM3_GQ")) : 0;
    int hgq = 1;
    if (fast_head && B <= FUMI_HCNT) {
        if (hgq_env > 0) hgq = hgq_env > 16 ? 16 : hgq_env;
        else while (hgq < 8 && B * hgq * 2 <= 256 && Qn / (hgq * 2) >= nwaves) hgq *= 2;      // (8 shares at 32 episodes x 160 rows: measured best)
    }
    if (hgq > 1) A((size_t)B * hgq * ((size_t)N * P + 2 + (size_t)N * N + P));
    if (need_grad) A((size_t)B * P);
    const int xks = xpanel_fwd_ksplit(B, S, Qn, D, P, 0);              // contraction parts of the image-encoder pass (narrow output)
    if (xks > 1) A((size_t)xks * (Rs + Rq) * P);
    int rc = ws_reserve(ws, bytes);
    if (rc) return rc;
    float* im = ws_f(ws, (Rs + Rq) * P);          // image embeddings, [B, S+Qn, P]: an episode's support rows, then its query rows
    float* t1 = ws_f(ws, Rs * Ht);
    float* tx = ws_f(ws, Rs * P);
    float* l1 = ws_f(ws, Rs * Ht);
    float* lc = ws_f(ws, 2 * B);                  // per-episode loss | correct
    float* confb = ws_f(ws, (size_t)B * N * N + B);       // per-episode confusion counts | lamda sums
    float* lamb = confb + (size_t)B * N * N;
    if (!stats) { confb = nullptr; lamb = nullptr; }
    else if (!fast_head) HIP_TRY(hipMemsetAsync(confb, 0, (size_t)B * N * N * sizeof(float), st));   // (the fast head stores its counts)
    const long imst = (long)(S + Qn) * P;         // episode stride of the panel
    float* imq = im + (long)S * P;

    GemmArgs g;
    ReduceSegs fin;
    int im_nparts = 0; const float* im_src = im; long im_pstride = 0;
    HyperFwdArgs fa;
    float* hfg = nullptr; bool g_split = false; int g_rode = 0;
    {
        // image encoder on every support and query row in ONE pass of the X-panel kernel (xpanel.hip: the panel [Xs_b;Xq_b]
        // times Wi^T per episode, rows never copied; G = NULL: no Gram block).  The bias is added where the head reads.
        ProfScope ps(ws, st, FUMI_PH_XPANEL_FWD);
        float* xparts = xks > 1 ? ws_f(ws, (size_t)xks * (Rs + Rq) * P) : nullptr;
        // the text MLP g (independent of the images) rides at the front of this launch when it can (hyper_fwd.h)
        hfg = mlp_fused ? ws_f(ws, hfg_n) : nullptr;
        g_split = mlp_fused && Rs < (1 << 30) / Ht &&
                  hyper_fwd_split_args((int)Rs, Dt, Ht, P, 0, text_s, G0, g0, G1, g1, t1, tx, hfg, ws->hcnt, &fa);
        if (g_split) { fa.d.drop_thr = thr; fa.d.drop_key = dkey(1); fa.d.drop_scale = dsc; }
        if ((rc = launch_xpanel_fwd(st, B, S, Qn, D, P, x_s, x_q, Wi, im, nullptr, nullptr, g_split ? &fa : nullptr, &g_rode, xparts,
                                    fast_head ? &im_nparts : nullptr))) return rc;
        if (im_nparts > 1) { im_src = xparts; im_pstride = (long)(Rs + Rq) * P; }        // the head adds the parts where it reads rows
    }
    {
        ProfScope ps(ws, st, FUMI_PH_HYPER_FWD);
        float* hfh = mlp_fused ? ws_f(ws, hfh_n) : nullptr;
        if (g_split) {
            if (!g_rode && (rc = launch_hyper_fwd_split(st, fa))) return rc;
        } else {
            g = gemm_args((int)Rs, Ht, Dt, text_s, Dt, G0, Dt, t1, Ht); g.bias = g0; g.act = 1;
            g.drop_thr = thr; g.drop_key = dkey(1); g.drop_scale = dsc;
            if ((rc = launch_gemm(st, g, 0, 0))) return rc;
            g = gemm_args((int)Rs, P, Ht, t1, Ht, G1, Ht, tx, P); g.bias = g1;
            if ((rc = launch_gemm(st, g, 0, 0))) return rc;
        }
        if (lamda_fixed < 0) {
            if (mlp_fused && Rs < (1 << 30) / Ht &&
                hyper_fwd_split_args((int)Rs, P, Ht, 1, 2 /* sigmoid */, tx, H0, h0, H1, h1, l1, lamda_s, hfh, ws->hcnt, &fa)) {
                fa.d.drop_thr = thr; fa.d.drop_key = dkey(2); fa.d.drop_scale = dsc;
                if ((rc = launch_hyper_fwd_split(st, fa))) return rc;
            } else {
                g = gemm_args((int)Rs, Ht, P, tx, P, H0, P, l1, Ht); g.bias = h0; g.act = 1;
                g.drop_thr = thr; g.drop_key = dkey(2); g.drop_scale = dsc;
                if ((rc = launch_gemm(st, g, 0, 0))) return rc;
                g = gemm_args((int)Rs, 1, Ht, l1, Ht, H1, Ht, lamda_s, 1); g.bias = h1; g.act = 3;
                if ((rc = launch_gemm(st, g, 0, 0))) return rc;
            }
        }
    }
    float *imb = nullptr, *txb = nullptr, *zlb = nullptr, *l1b = nullptr, *t1b = nullptr, *slabs = nullptr, *cpart = nullptr, *wslabs = nullptr;
    if (need_grad) {
        cpart = nullptr;
        imb = ws_f(ws, (Rs + Rq) * P); txb = ws_f(ws, Rs * P); zlb = ws_f(ws, Rs);
        l1b = ws_f(ws, Rs * Ht); t1b = ws_f(ws, Rs * Ht); slabs = ws_f(ws, (size_t)xns * P * D);
        cpart = ws_f(ws, cpart_n);
        wslabs = ws_f(ws, wslab_n);
    }
    float* hpart = hgq > 1 ? ws_f(ws, (size_t)B * hgq * ((size_t)N * P + 2 + (size_t)N * N + P)) : nullptr;
    float* bias_bar = (need_grad && fast_head) ? ws_f(ws, (size_t)B * P) : nullptr;
    {
        ProfScope ps(ws, st, FUMI_PH_AM3);
        const float dscale = grad_scale / (float)Qn;
        if (fast_head) {
            FUMI_SET_DYN_LDS(am3_head_kernel, lds);
            hipLaunchKernelGGL(am3_head_kernel, dim3(8 * ((B + 7) / 8) * hgq), dim3(64 * nwaves), lds, st, N, S, Qn, P, lamda_fixed,
                               need_grad ? 1 : 0, dscale, im_src, tx, lamda_s, y_s, im_src + (long)S * P, y_q, preds_q, lc, lc + B, confb, lamb,
                               grad_scale / (float)S, imb, txb, zlb, imb ? imb + (long)S * P : nullptr, imst, bi, ws->status,
                               B, hgq, hpart, ws->hcnt, im_nparts > 1 ? im_nparts : 1, im_pstride, bias_bar);
        } else {
            hipLaunchKernelGGL(am3_bias_rows_kernel, dim3(256), dim3(256), 0, st, im, bi, (long)(Rs + Rq) * P, P);
            FUMI_SET_DYN_LDS(am3_head_generic_kernel, lds);
            hipLaunchKernelGGL(am3_head_generic_kernel, dim3(B), dim3(64 * nwaves), lds, st, N, S, Qn, P, lamda_fixed,
                               need_grad ? 1 : 0, dscale, im, tx, lamda_s, y_s, imq, y_q, preds_q, lc, lc + B, confb, lamb,
                               grad_scale / (float)S, imb, txb, zlb, imb ? imb + (long)S * P : nullptr, imst, bi, ws->status);
        }
        LAUNCH_CHECK();
        fin.n = 0; fin.scale = 1.f;                              // per-episode loss / correct counts -> scalars: with gradients they
        fin.add(lc, B, 1, 1, loss);                              // join the step's final reduction, else one small launch here
        fin.add(lc + B, B, 1, 1, correct);
        if (stats) {        // [loss | correct count | this rank's share of the mean lamda | confusion counts]
            fin.add(lc, B, 1, 1, stats);
            fin.add(lc + B, B, 1, 1, stats + 1);
            fin.add(lamb, B, 1, 1, stats + 2);
            fin.add(confb, B, (long)N * N, (long)N * N, stats + 3);
        }
        if (!need_grad && (rc = launch_reduce_multi(st, fin))) return rc;
    }
    if (!need_grad) return FUMI_OK;
    if (dx_s && dx_q) {                     // one batched product per side: rows of episode b sit at imbar + b (S+Qn) P
        g = gemm_args(S, D, P, imb, P, Wi, D, dx_s, D); g.nbatch = B; g.sA = imst; g.sC = (long)S * D;
        if ((rc = launch_gemm(st, g, 0, 1))) return rc;
        g = gemm_args(Qn, D, P, imb + (long)S * P, P, Wi, D, dx_q, D); g.nbatch = B; g.sA = imst; g.sC = (long)Qn * D;
        if ((rc = launch_gemm(st, g, 0, 1))) return rc;
    }

    // (the phase ends before the image-encoder gradient starts: the text-MLP backward's time used to include that launch and the
    // step's final reductions)
    std::optional<ProfScope> pb; pb.emplace(ws, st, FUMI_PH_HYPER_BWD);
    ColsumJobs cj; cj.n = 0; cj.part_total = 0;
    // weight gradients of the text MLPs contract over all B*S rows with only a few output tiles: the contraction is cut into
    // 128-row slabs (one workgroup each) that the step's final reduction sums together with everything else
    ReduceSegs tail_; tail_.n = 0; tail_.scale = 1.f;
    const int WKC = 128, wns = (int)((Rs + WKC - 1) / WKC);
    float* wslab_next = wslabs;
    auto wgrad = [&](int M_, int N_, const float* A_, long lda_, const float* B_, long ldb_, float* out) -> int {
        GemmArgs q = gemm_args(M_, N_, (int)Rs, A_, lda_, B_, ldb_, wslab_next, N_);
        q.kchunk = WKC; q.nsplit = wns; q.sCsplit = (long)M_ * N_;
        int r_ = launch_gemm(st, q, 1, 1);
        if (r_) return r_;
        tail_.add(wslab_next, wns, (long)M_ * N_, (long)M_ * N_, out);
        wslab_next += (size_t)wns * M_ * N_;
        return FUMI_OK;
    };
    HyperBwdArgs ba;
    int tx_nparts = 0;
    // (the partials are only used when BOTH backward passes take the fused form: g's is checked first)
    HyperBwdArgs probe; ReduceSegs dummy; dummy.n = 0; dummy.scale = 1.f;
    float* txparts = nullptr;
    if (mlp_fused && lamda_fixed < 0 && txp_n) {
        float* cand = ws_f(ws, txp_n);
        if (hyper_bwd_fused_args((int)Rs, Dt, Ht, P, 0, dsc, text_s, t1, nullptr, txb, G1, cand /* any aligned pointer */, nullptr,
                                 g_w[3], g_w[4], g_w[5], &dummy, &probe, t1b)) txparts = cand;
    }
    float* hbh = (mlp_fused && need_grad) ? ws_f(ws, hbh_n) : nullptr;
    float* hbg = (mlp_fused && need_grad) ? ws_f(ws, hbg_n) : nullptr;
    if (lamda_fixed < 0) {
        // h network: lam = sigmoid(l1 H1^T + h1), l1 = relu(tx H0^T + h0)
        if (mlp_fused && hyper_bwd_fused_args((int)Rs, P, Ht, 1, 0, dsc, tx, l1, nullptr, zlb, H1, hbh, g_w[6], g_w[7], g_w[8], g_w[9],
                                              &tail_, &ba, l1b)) {
            // txbar += l1bar H0 leaves this launch as per-(row block, chunk) partials that g's backward adds while it stages txbar
            if (txparts && P <= HBW_XDT && (P & 3) == 0) { ba.A0 = H0; ba.xpart = txparts; tx_nparts = Ht / 64; }
            if ((rc = launch_hyper_bwd_fused(st, ba))) return rc;                      // gH1, gh1, l1bar, gh0, gH0 (row-block slabs)
        } else {
            if ((rc = wgrad(1, Ht, zlb, 1, l1, Ht, g_w[8]))) return rc;                // gH1 = zlbar^T l1
            cj.add(zlb, (int)Rs, 1, 1, g_w[9]);
            g = gemm_args((int)Rs, Ht, 1, zlb, 1, H1, Ht, l1b, Ht);                    // l1bar = (zlbar H1) * relu'(l1) * dropout scale
            g.mask = l1; g.alpha = dsc;
            if ((rc = launch_gemm(st, g, 0, 1))) return rc;
            if ((rc = wgrad(Ht, P, l1b, Ht, tx, P, g_w[6]))) return rc;                // gH0 = l1bar^T tx
            cj.add(l1b, (int)Rs, Ht, Ht, g_w[7]);
        }
        if (!tx_nparts) {
            g = gemm_args((int)Rs, P, Ht, l1b, Ht, H0, P, txb, P); g.accumulate = 1;   // txbar += l1bar H0
            if ((rc = launch_gemm(st, g, 0, 1))) return rc;
        }
    } else {
        HIP_TRY(hipMemsetAsync(g_w[6], 0, (size_t)Ht * P * 4, st)); HIP_TRY(hipMemsetAsync(g_w[7], 0, (size_t)Ht * 4, st));
        HIP_TRY(hipMemsetAsync(g_w[8], 0, (size_t)Ht * 4, st)); HIP_TRY(hipMemsetAsync(g_w[9], 0, 4, st));
    }
    // g network: tx = t1 G1^T + g1, t1 = relu(text G0^T + g0)
    if (mlp_fused && hyper_bwd_fused_args((int)Rs, Dt, Ht, P, 0, dsc, text_s, t1, nullptr, txb, G1, hbg, nullptr, g_w[3], g_w[4], g_w[5],
                                          &tail_, &ba, t1b)) {
        if (tx_nparts) { ba.hbar_parts = txparts; ba.hbar_nparts = tx_nparts; }
        if ((rc = launch_hyper_bwd_fused(st, ba))) return rc;                          // gG1, gg1, t1bar, gg0 (row-block slabs)
    } else {
        if (tx_nparts) return FUMI_EINVAL;                                             // (cannot happen: probed above)
        if ((rc = wgrad(P, Ht, txb, P, t1, Ht, g_w[4]))) return rc;                    // gG1 = txbar^T t1
        cj.add(txb, (int)Rs, P, P, g_w[5]);
        g = gemm_args((int)Rs, Ht, P, txb, P, G1, Ht, t1b, Ht);                        // t1bar = (txbar G1) * relu'(t1) * dropout scale
        g.mask = t1; g.alpha = dsc;
        if ((rc = launch_gemm(st, g, 0, 1))) return rc;
        cj.add(t1b, (int)Rs, Ht, Ht, g_w[3]);
    }
    if ((rc = wgrad(Ht, Dt, t1b, Ht, text_s, Dt, g_w[2]))) return rc;                  // gG0 = t1bar^T text (800 x 768: 128-row slabs)
    if (float* tg = ws->text_grad) {                  // armed by fumi_hip_want_text_grad: d loss / d text_s = t1bar G0  [B*S,Dt]
        ws->text_grad = nullptr;                      // (t1bar carries grad_scale and the dropout scale already)
        g = gemm_args((int)Rs, Dt, Ht, t1b, Ht, G0, Dt, tg, Dt);
        if ((rc = launch_gemm(st, g, 0, 1))) return rc;
    }
    pb.reset();
    // image encoder: gWi = imbar_s^T Xs + imbar_q^T Xq (split over the contraction), gbi = colsum(imbar)
    {
        ProfScope pg(ws, st, FUMI_PH_XPANEL_BWD);
        if ((rc = launch_xpanel_bwd(st, B, S, Qn, D, P, x_s, x_q, imb, slabs, xkc, xns))) return rc;
    }
    {
        ProfScope pr(ws, st, FUMI_PH_REDUCE);
        const long slab = (long)P * D;
        if (bias_bar) tail_.add(bias_bar, B, P, P, g_w[1]);        // the head kernel left the episodes' column sums of imbar
        else cj.add(imb, (int)(Rs + Rq), P, P, g_w[1]);
        // every bias gradient (column sums over all rows) and the image-encoder weight slabs: two launches in all
        tail_.add(slabs, xns, slab, slab, g_w[0]);
        if (!tail_.append(fin) && (rc = launch_reduce_multi(st, fin))) return rc;      // loss / correct / statistics: same final launch
        if ((rc = launch_colsum_multi(st, cj, cpart, &tail_))) return rc;
    }
    return FUMI_OK;
}

extern "C" int fumi_hip_am3_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int Dt, int Ht, int P, int lamda_fixed, int need_grad, float grad_scale,
        float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q, const float* text_s,
        const float* const* w, float* loss, int64_t* preds_q, float* lamda_s, float* correct, float* const* g_w,
        float* stats) {
    return am3_step_impl(ws, stream, B, N, S, Qn, D, Dt, Ht, P, lamda_fixed, need_grad, grad_scale, dropout_p, seed, x_s, y_s, x_q, y_q,
                         text_s, w, loss, preds_q, lamda_s, correct, g_w, stats, nullptr, nullptr);
}

extern "C" int fumi_hip_am3_step_dx(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int Dt, int Ht, int P, int lamda_fixed, int need_grad, float grad_scale,
        float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q, const float* text_s,
        const float* const* w, float* loss, int64_t* preds_q, float* lamda_s, float* correct, float* const* g_w,
        float* stats, float* dx_s, float* dx_q) {
    if (need_grad && (!dx_s || !dx_q)) return FUMI_EINVAL;
    return am3_step_impl(ws, stream, B, N, S, Qn, D, Dt, Ht, P, lamda_fixed, need_grad, grad_scale, dropout_p, seed, x_s, y_s, x_q, y_q,
                         text_s, w, loss, preds_q, lamda_s, correct, g_w, stats, dx_s, dx_q);
}

// ---- accuracy and macro precision / recall / F1 from the confusion matrix (what the reference asks sklearn for on the host
// every meta-batch, utils.py:319-326): lane c owns class c; averages run over the classes that occur among targets or
// predictions, an undefined ratio counts as 0 (zero_division).  stats = [loss, -, lamda, conf[N*N]] as fumi_hip_am3_step
// writes them (after the all-reduce over ranks, if any); out6 = [loss, acc, f1, prec, rec, lamda].
namespace {
__global__ __launch_bounds__(64) void am3_metrics_kernel(int N, const float* __restrict__ stats, float* __restrict__ out) {
    const int c = threadIdx.x;
    const float* cf = stats + 3;
    float tp = 0.f, pp = 0.f, tt = 0.f;
    if (c < N) {
        tp = cf[c * N + c];
        for (int k = 0; k < N; ++k) { pp += cf[k * N + c]; tt += cf[c * N + k]; }
    }
    const float prec = pp > 0.f ? tp / pp : 0.f, rec = tt > 0.f ? tp / tt : 0.f;
    const float f1 = prec + rec > 0.f ? 2.f * prec * rec / (prec + rec) : 0.f;
    const float present = (pp + tt) > 0.f ? 1.f : 0.f;
    const float nl = fmaxf(wave_sum(present), 1.f);
    const float sp = wave_sum(prec * present), sr = wave_sum(rec * present), sf = wave_sum(f1 * present);
    const float stp = wave_sum(tp), sall = wave_sum(tt);
    if (c == 0) {
        out[0] = stats[0]; out[1] = sall > 0.f ? stp / sall : 0.f; out[2] = sf / nl; out[3] = sp / nl; out[4] = sr / nl;
        out[5] = stats[2];
    }
}
}  // namespace

extern "C" int fumi_hip_am3_metrics(fumi_ws_t* ws, fumi_stream_t stream, int N, const float* stats, float* out6) {
    if (!ws || !stats || !out6 || N < 1) return FUMI_EINVAL;
    if (N > 64) return FUMI_ENOTSUP;
    HIP_TRY(hipSetDevice(ws->device));
    hipLaunchKernelGGL(am3_metrics_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, N, stats, out6);
    LAUNCH_CHECK();
    return FUMI_OK;
}

