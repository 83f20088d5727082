// MAML with hidden_dims=None (fumi/models/maml.py:15-33: the network is the single MetaLinear lin_final on the embeddings).
// The same low-rank algebra as the FuMI layer 0, with the head as the only layer: with e_t = (softmax(z_t) - onehot)/S the
// support residual of inner step t,
//     W_t = W - alpha E_t^T Xs,  E_t = sum_{tau<t} e_tau [S,N],  b_t = b - alpha colsum(E_t)
//     z_t(X) = A(X) - alpha G(X) E_t + b_t,    A = X W^T [R,N],  G = X Xs^T [R,S]      (one xpanel_fwd launch, h0 = N)
// so an episode is a few [S,N] / [Qn,N] matrices.  Second-order reverse sweep (derivation as for episode.hip):
//     zbar_q = (softmax(z_q) - onehot)/Qn ;  Ebar_T = -alpha (G_qs^T zbar_q + 1 colsum(zbar_q)^T) ;  bbar = colsum(zbar_q)
//     t = T-1..0:  zbar_t = p_t * (Ebar_{t+1}/S - <p_t, Ebar_{t+1}/S>) ;  Abar_s += zbar_t ;  bbar += colsum(zbar_t)
//                  Ebar_t = Ebar_{t+1} - alpha (G_ss^T zbar_t + 1 colsum(zbar_t)^T)
//     gW = sum_b [Abar_s ; zbar_q]_b^T [Xs ; Xq]_b   (one xpanel_bwd launch),  gb = sum_b bbar_b
// First-order MAML keeps only the query terms.  One workgroup per episode, plain loops (the matrices are tiny).
#include "common.h"

namespace {

struct LinHead {
    int B, N, S, Qn, T, need_grad, second_order;
    float alpha;
};

__global__ __launch_bounds__(256) void linhead_kernel(LinHead d, const float* __restrict__ A, const float* __restrict__ G,
                                                      const float* __restrict__ bias, const int64_t* __restrict__ y_s,
                                                      const int64_t* __restrict__ y_q, float* __restrict__ logits_q,
                                                      int64_t* __restrict__ preds_q, float* __restrict__ preds_f,
                                                      float* __restrict__ loss_b, float* __restrict__ acc_b,
                                                      float* __restrict__ Abar, float* __restrict__ bbar, int* status) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int N = d.N, S = d.S, Qn = d.Qn, R = S + Qn, T = d.T;
    const float alpha = d.alpha;
    const bool taped = d.need_grad && d.second_order;
    float* E = sm;                       // [S,N]
    float* Eb = E + S * N;               // [S,N] adjoint
    float* Z = Eb + S * N;               // [S,N] scratch (zbar_t)
    float* cs = Z + S * N;               // [N] colsum(E)
    float* cz = cs + N;                  // [N] colsum(zbar)
    float* bb = cz + N;                  // [N] bbar
    float* red = bb + N;                 // [2*nt/64] loss / correct partials
    float* lbar = red + 16;              // [Qn,N]
    float* tape = lbar + (Qn > S ? Qn : S) * N;   // [T,S,N] p_t (second order only); lbar's slot also serves zbar_t [S,N]
    const float* As = A + (long)b * R * N;  const float* Aq = As + (long)S * N;
    const float* Gss = G + (long)b * R * S; const float* Gqs = Gss + (long)S * S;
    const int64_t* ys = y_s + (long)b * S;  const int64_t* yq = y_q + (long)b * Qn;

    for (int i = tid; i < S * N; i += nt) { E[i] = 0.f; Eb[i] = 0.f; }
    for (int n = tid; n < N; n += nt) { cs[n] = 0.f; bb[n] = 0.f; }
    __syncthreads();
    // ---- inner loop on the support set
    for (int t = 0; t < T; ++t) {
        for (int s = tid; s < S; s += nt) {
            long yv = ys[s];
            if (yv < 0 || yv >= N) { atomicOr(status, FUMI_ST_LABEL_RANGE); yv = 0; }
            float mx = -INFINITY;
            for (int n = 0; n < N; ++n) {
                float z = As[s * N + n] + bias[n] - alpha * cs[n];
                float acc = 0.f;
                for (int k = 0; k < S; ++k) acc += Gss[s * S + k] * E[k * N + n];
                z -= alpha * acc;
                Z[s * N + n] = z; mx = fmaxf(mx, z);
            }
            float sum = 0.f;
            for (int n = 0; n < N; ++n) sum += expf(Z[s * N + n] - mx);
            const float inv = 1.f / sum;
            for (int n = 0; n < N; ++n) {
                const float pv = expf(Z[s * N + n] - mx) * inv;
                if (taped) tape[((long)t * S + s) * N + n] = pv;
                Z[s * N + n] = (pv - (n == (int)yv ? 1.f : 0.f)) / (float)S;          // e_t
            }
        }
        __syncthreads();
        for (int i = tid; i < S * N; i += nt) E[i] += Z[i];
        __syncthreads();
        for (int n = tid; n < N; n += nt) { float a = 0.f; for (int s = 0; s < S; ++s) a += E[s * N + n]; cs[n] = a; }
        __syncthreads();
    }
    // ---- query set
    float ls = 0.f, cr = 0.f;
    for (int r = tid; r < Qn; r += nt) {
        long yv = yq[r];
        if (yv < 0 || yv >= N) { atomicOr(status, FUMI_ST_LABEL_RANGE); yv = 0; }
        float* lq = logits_q + ((long)b * Qn + r) * N;
        float mx = -INFINITY; int arg = 0;
        for (int n = 0; n < N; ++n) {
            float z = Aq[r * N + n] + bias[n] - alpha * cs[n];
            float acc = 0.f;
            for (int k = 0; k < S; ++k) acc += Gqs[r * S + k] * E[k * N + n];
            z -= alpha * acc;
            lq[n] = z;
            if (z > mx) { mx = z; arg = n; }                                          // first arg-max (torch.max)
        }
        float sum = 0.f;
        for (int n = 0; n < N; ++n) sum += expf(lq[n] - mx);
        ls += mx + logf(sum) - lq[yv];
        cr += arg == (int)yv ? 1.f : 0.f;
        preds_q[(long)b * Qn + r] = arg;
        if (preds_f) preds_f[(long)b * Qn + r] = (float)arg;
        const float inv = 1.f / sum;
        for (int n = 0; n < N; ++n) lbar[r * N + n] = (expf(lq[n] - mx) * inv - (n == (int)yv ? 1.f : 0.f)) / (float)Qn;
    }
    for (int o = 32; o > 0; o >>= 1) { ls += __shfl_down(ls, o, 64); cr += __shfl_down(cr, o, 64); }
    if ((tid & 63) == 0) { red[tid >> 6] = ls; red[4 + (tid >> 6)] = cr; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, c = 0.f;
        for (int w_ = 0; w_ < (nt >> 6); ++w_) { a += red[w_]; c += red[4 + w_]; }
        loss_b[b] = a / (float)Qn; acc_b[b] = c / (float)Qn;
    }
    if (!d.need_grad) return;
    // ---- backward: query rows of Abar, bbar, Ebar_T
    float* Ab_s = Abar + (long)b * R * N; float* Ab_q = Ab_s + (long)S * N;
    for (int i = tid; i < Qn * N; i += nt) Ab_q[i] = lbar[i];
    for (int n = tid; n < N; n += nt) { float a = 0.f; for (int r = 0; r < Qn; ++r) a += lbar[r * N + n]; cz[n] = a; bb[n] = a; }
    for (int i = tid; i < S * N; i += nt) Z[i] = 0.f;                                // Abar_s accumulator
    __syncthreads();
    if (d.second_order) {
        for (int i = tid; i < S * N; i += nt) {
            const int s = i / N, n = i - s * N;
            float a = 0.f;
            for (int r = 0; r < Qn; ++r) a += Gqs[r * S + s] * lbar[r * N + n];
            Eb[i] = -alpha * (a + cz[n]);
        }
        __syncthreads();
        float* zb = lbar;                                                            // reuse: [S,N] fits (S <= Qn not needed: sized max)
        for (int t = T - 1; t >= 0; --t) {
            const float* p = tape + (long)t * S * N;
            for (int s = tid; s < S; s += nt) {
                float dot = 0.f;
                for (int n = 0; n < N; ++n) dot += p[s * N + n] * Eb[s * N + n];
                for (int n = 0; n < N; ++n) zb[s * N + n] = p[s * N + n] * (Eb[s * N + n] - dot) / (float)S;
            }
            __syncthreads();
            for (int n = tid; n < N; n += nt) { float a = 0.f; for (int s = 0; s < S; ++s) a += zb[s * N + n]; cz[n] = a; bb[n] += a; }
            for (int i = tid; i < S * N; i += nt) Z[i] += zb[i];
            __syncthreads();
            for (int i = tid; i < S * N; i += nt) {
                const int s = i / N, n = i - s * N;
                float a = 0.f;
                for (int k = 0; k < S; ++k) a += Gss[k * S + s] * zb[k * N + n];
                Eb[i] -= alpha * (a + cz[n]);
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < S * N; i += nt) Ab_s[i] = Z[i];
    for (int n = tid; n < N; n += nt) bbar[(long)b * N + n] = bb[n];
}

}  // namespace

size_t linhead_lds_floats(int N, int S, int Qn, int T, int taped) {
    const size_t sq = (size_t)(Qn > S ? Qn : S) * N;
    return 3 * (size_t)S * N + 3 * (size_t)N + 16 + sq + (taped ? (size_t)T * S * N : 0);
}

int launch_linhead(hipStream_t st, int B, int N, int S, int Qn, int T, float alpha, int need_grad, int second_order,
                   const float* A, const float* G, const float* bias, const int64_t* y_s, const int64_t* y_q, float* logits_q,
                   int64_t* preds_q, float* preds_f, float* loss_b, float* acc_b, float* Abar, float* bbar, int* status) {
    const size_t fl = linhead_lds_floats(N, S, Qn, T, need_grad && second_order);
    if (fl > 38000) return FUMI_ENOTSUP;
    LinHead d{B, N, S, Qn, T, need_grad, second_order, alpha};
    FUMI_SET_DYN_LDS(linhead_kernel, fl * 4);
    hipLaunchKernelGGL(linhead_kernel, dim3(B), dim3(256), fl * 4, st, d, A, G, bias, y_s, y_q, logits_q, preds_q, preds_f, loss_b,
                       acc_b, Abar, bbar, status);
    LAUNCH_CHECK();
    return FUMI_OK;
}
