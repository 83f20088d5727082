// Shared declarations of the gfx950 engine (host side + device helpers).  Written for MI355X only:
// wave64, MFMA f32 (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 = exact fp32 fma chains), 160 KiB LDS/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/fumi_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------------------
// workspace: one hipMalloc'd slab, carved per call by a bump allocator (all sizes known on the host).
// ------------------------------------------------------------------------------------------------------------
struct ProfRec { int phase; hipEvent_t a, b; };
struct fumi_ws {
    int device;
    char* base;
    size_t cap;
    size_t off;          // bump pointer (bytes), reset at the start of each step
    int* status;         // device status word (own small allocation)
    int* status_host;    // pinned
    int profiling;       // record HIP events around each phase (bench only)
    std::vector<ProfRec>* recs;
    std::vector<hipEvent_t>* pool;
};

// RAII phase timer: two hipEventRecords on the caller's stream when profiling is on, nothing otherwise
struct ProfScope {
    fumi_ws* ws; hipStream_t st; hipEvent_t b; bool on;
    ProfScope(fumi_ws* w, hipStream_t s, int phase);
    ~ProfScope();
};

void fumi_set_hip_error(hipError_t e, const char* where);

#define HIP_TRY(expr)                                                      \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) { fumi_set_hip_error(_e, #expr); return FUMI_EHIP; } \
    } while (0)

#define LAUNCH_CHECK()                                                     \
    do {                                                                   \
        hipError_t _e = hipGetLastError();                                 \
        if (_e != hipSuccess) { fumi_set_hip_error(_e, __func__); return FUMI_EHIP; } \
    } while (0)

// make sure the slab holds `bytes`; grows (synchronising) when it does not
int ws_reserve(fumi_ws* ws, size_t bytes);
static inline size_t ws_align(size_t b) { return (b + 255) & ~(size_t)255; }
// carve n floats (call only after ws_reserve of the total)
static inline float* ws_f(fumi_ws* ws, size_t n) {
    float* p = (float*)(ws->base + ws->off);
    ws->off += ws_align(n * sizeof(float));
    return p;
}

// ------------------------------------------------------------------------------------------------------------
// dense GEMM family (gemm.hip):  C = act(alpha * op(A) op(B) + bias) [+ C]
//   AL: 0 -> A(m,k) = A[m*lda + k]  (k contiguous)      1 -> A(m,k) = A[k*lda + m]  (m contiguous, "A^T stored")
//   BL: 0 -> B(k,n) = B[n*ldb + k]  (k contiguous, "NT") 1 -> B(k,n) = B[k*ldb + n]  (n contiguous, "NN")
//   batch z in [0,nbatch): pointers advance by sA/sB/sC elements; split s in [0,nsplit): contraction range
//   [s*kchunk, min(K,(s+1)*kchunk)) and the result goes to C + s*sCsplit (partial slabs, summed by reduce_slabs).
// ------------------------------------------------------------------------------------------------------------
struct GemmArgs {
    int M, N, K;
    int kchunk, nsplit, nbatch;
    const float* A; long lda; long sA;
    const float* B; long ldb; long sB;
    float* C; long ldc; long sC; long sCsplit;
    const float* bias;     // [N] or NULL
    int act;               // 0 none, 1 relu, 2 tanh, 3 sigmoid
    float alpha;
    int accumulate;        // C += result
};
static inline GemmArgs gemm_args(int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                                 float* C, long ldc) {
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.kchunk = K; g.nsplit = 1; g.nbatch = 1;
    g.A = A; g.lda = lda; g.sA = 0; g.B = B; g.ldb = ldb; g.sB = 0;
    g.C = C; g.ldc = ldc; g.sC = 0; g.sCsplit = 0; g.bias = nullptr; g.act = 0; g.alpha = 1.f; g.accumulate = 0;
    return g;
}
int launch_gemm(hipStream_t st, const GemmArgs& g, int AL, int BL);
// out[i] = scale * sum_s slabs[s*stride + i]  (+ optional second slab set), i < n
int launch_reduce_slabs(hipStream_t st, const float* slabs, int nslab, long stride, long n, float scale, float* out);
// out[n] = scale * sum_m X[m*ld + n]
int launch_colsum(hipStream_t st, const float* X, int M, int N, long ld, float scale, float* out);

// ------------------------------------------------------------------------------------------------------------
// the two shared passes over the wide inputs (xpanel.hip)
// ------------------------------------------------------------------------------------------------------------
int launch_xpanel_fwd(hipStream_t st, int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q,
                      const float* W0, float* A0 /*[B,S+Qn,h0]*/, float* G /*[B,S+Qn,S]*/);
int xpanel_bwd_nsplit(int B, int S, int Qn, int D, int h0, int* kchunk_out);
int launch_xpanel_bwd(hipStream_t st, int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q,
                      const float* Abar /*[B,S+Qn,h0]*/, float* slabs /*[nsplit,h0,D]*/, int kchunk, int nsplit);

// ------------------------------------------------------------------------------------------------------------
// episode engine (episode.hip): inner-loop adaptation, query pass, second-order reverse sweep
// ------------------------------------------------------------------------------------------------------------
struct EpisodeProblem {
    int B, N, S, Qn, D, L;          // L hidden layers (>= 1), each followed by ReLU
    int h[FUMI_MAX_HIDDEN];
    int T;
    float alpha;
    int need_grad, second_order;
    float grad_scale;
    const float* x_s; const int64_t* y_s; const float* x_q; const int64_t* y_q;
    const float* W[FUMI_MAX_HIDDEN]; const float* b[FUMI_MAX_HIDDEN];
    const float* head;              // [B,N,H+1] initial head per episode  ([Wh | bh])
    float* logits_q; int64_t* preds_q; float* loss_b; float* acc_b;
    float* gW[FUMI_MAX_HIDDEN]; float* gb[FUMI_MAX_HIDDEN];   // outputs (scaled sums over episodes)
    float* head_bar;                // [B,N,H+1] d loss_b / d head_b   (unscaled, per episode)
};
size_t episode_workspace_bytes(const EpisodeProblem& p);
int run_episodes(fumi_ws* ws, hipStream_t st, const EpisodeProblem& p);

// small kernels shared by the entry points (episode.hip)
int launch_class_text_select(hipStream_t st, int B, int N, int S, int Dt, const float* text_s, const int64_t* y_s,
                             float* out, int* status);
int launch_broadcast_head(hipStream_t st, int B, int N, int H, const float* Wf, const float* bf, float* head);
int launch_tanh_bwd(hipStream_t st, long n, const float* h, const float* hbar, float* out);
int launch_relu_mask_mul(hipStream_t st, long n, const float* u, float* g_inout);
int launch_split_head_grad(hipStream_t st, int B, int N, int H, const float* head_bar, float scale, float* gW, float* gb);

#ifdef __HIPCC__
// ------------------------------------------------------------------------------------------------------------
// wg_mm: workgroup-cooperative small matrix product straight from (L2-resident) memory on the f32 MFMA.
//   for m<M, n<N:  epi(m, n, sum_k A(m,k) * B(k,n)),   A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn]
// Every wave owns 16x16 output tiles (v_mfma_f32_16x16x4_f32: lane l feeds A[l&15][l>>4], B[l>>4][l&15]; result
// register r of lane l is row 4*(l>>4)+r, column l&15).  All lanes run the MFMAs (loop bounds are wave-uniform);
// out-of-range operands are fed as zeros.  The caller synchronises (__syncthreads) between dependent products.
// ------------------------------------------------------------------------------------------------------------
template <class Epi>
__device__ __forceinline__ void wg_mm(int M, int N, int K, const float* A, long sam, long sak,
                                      const float* B, long sbk, long sbn, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int tm = (M + 15) >> 4, tn = (N + 15) >> 4;
    const int r = lane & 15, q = lane >> 4;
    for (int t = wave; t < tm * tn; t += nw) {
        const int m0 = (t / tn) << 4, n0 = (t % tn) << 4;
        const int am = m0 + r, bn = n0 + r;
        const bool aok = am < M, bok = bn < N;
        const float* ap = A + (long)(aok ? am : 0) * sam;
        const float* bp = B + (long)(bok ? bn : 0) * sbn;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 16 <= K; k += 16) {
            float a0 = ap[(long)(k + q) * sak], a1 = ap[(long)(k + 4 + q) * sak];
            float a2 = ap[(long)(k + 8 + q) * sak], a3 = ap[(long)(k + 12 + q) * sak];
            float b0 = bp[(long)(k + q) * sbk], b1 = bp[(long)(k + 4 + q) * sbk];
            float b2 = bp[(long)(k + 8 + q) * sbk], b3 = bp[(long)(k + 12 + q) * sbk];
            if (!aok) { a0 = a1 = a2 = a3 = 0.f; }
            if (!bok) { b0 = b1 = b2 = b3 = 0.f; }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
        }
        for (; k < K; k += 4) {
            const int kk = k + q;
            const bool kok = kk < K;
            float a = ap[(long)(kok ? kk : 0) * sak], b = bp[(long)(kok ? kk : 0) * sbk];
            if (!(aok && kok)) a = 0.f;
            if (!(bok && kok)) b = 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + q * 4 + i, n = n0 + r;
            if (m < M && n < N) epi(m, n, acc[i]);
        }
    }
}

// column sums of X[M,N] (row stride ld): f(n, sum_m X[m,n]); one thread per column
template <class F>
__device__ __forceinline__ void wg_colsum(int M, int N, const float* X, long ld, F&& f) {
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += X[(long)m * ld + n];
        f(n, s);
    }
}
#endif
