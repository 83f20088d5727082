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
struct f32pair { float x, y; };      // two-float "pre" value of wg_mm2 epilogues
struct ProfRec { int phase; hipEvent_t a, b; };
struct fumi_ws {
    int device;
    char* base;
    size_t cap;
    size_t off;          // bump pointer (bytes), reset at the start of each step
    int* status;         // device status word (own small allocation)
    int* status_host;    // pinned
    const float* pub_src; float* pub_dst; int pub_n; unsigned long long pub_seq;   // deferred publication (api: publish_scalars_deferred)
    struct AdamPending* adam;   // deferred optimizer step (fumi_hip_adam_step_deferred): folded into the step's final reduction launch
    struct GlovePending* glove; // deferred embedding bag (fumi_hip_glove_bag_select_deferred): rider workgroups of the step's first launch
    float* text_grad;    // armed by fumi_hip_want_text_grad: the next FuMI step with need_grad also writes d loss / d class text rows here
    int* acnt;           // [FUMI_ACNT] arrival counters of the split adapt kernel (reset by the query kernel of the same step)
    int* hcnt;           // [FUMI_HCNT] arrival counters of hyper_fwd_split_kernel, zero between launches
    float* side_buf; size_t side_cap;   // small allocation that survives slab rewinds (ResNet-12 chunk loop: heads of the whole meta-batch)
    unsigned short* w0p; size_t w0p_cap;   // the layer-0 weight as three bf16 planes in MFMA fragment order (xpanel.hip), own allocation
    int profiling;       // bit p: record HIP events around phase p (bench only)
    int prof_every;      // ... at every prof_every-th occurrence of the phase
    unsigned prof_seen[32];
    hipStream_t side;    // second stream: the text path (hypernetwork fwd / bwd) runs beside the two X-panel passes
    hipStream_t lane;    // ResNet-12 / Conv4: the stream of the second lane of episodes (created on first use); = lanes[0]
    hipStream_t lanes[3]; hipEvent_t lane_ev[3];   // streams of lanes 1..3 and their join events (ws_lane_stream)
    hipEvent_t ev[4];    // fork / join points (timing disabled)
    hipEvent_t evx[2];   // fork / join of the query-row half of the backward X-panel pass beside the reverse sweep (episode.hip)
    std::vector<ProfRec>* recs;
    std::vector<hipEvent_t>* pool;
};

// stream of lane i (1..3) of a multi-lane step, created on first use together with its join event; nullptr when that fails
hipStream_t ws_lane_stream(fumi_ws* ws, int i);

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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size growth) instead of before every launch
#define FUMI_SET_DYN_LDS(kernel, bytes)                                                                   \
    do {                                                                                                  \
        static int _max_set[64];                                                                          \
        static bool _init = false;                                                                        \
        if (!_init) { for (int& _v : _max_set) _v = -1; _init = true; }                                   \
        int _dev = 0;                                                                                     \
        HIP_TRY(hipGetDevice(&_dev));                                                                     \
        if (_dev < 0 || _dev >= 64 || (int)(bytes) > _max_set[_dev]) {                                    \
            HIP_TRY(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            if (_dev >= 0 && _dev < 64) _max_set[_dev] = (int)(bytes);                                    \
        }                                                                                                 \
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
    unsigned drop_thr, drop_key; float drop_scale;   // act == 1 only: dropout after the ReLU (keep iff mix(key ^ (m*N+n)) >= thr)
    const float* mask;     // [M, ldc] or NULL: result kept where mask > 0, else 0 (a ReLU derivative fused into the product)
};
static inline GemmArgs gemm_args(int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                                 float* C, long ldc) {
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.kchunk = K; g.nsplit = 1; g.nbatch = 1;
    g.A = A; g.lda = lda; g.sA = 0; g.B = B; g.ldb = ldb; g.sB = 0;
    g.C = C; g.ldc = ldc; g.sC = 0; g.sCsplit = 0; g.bias = nullptr; g.act = 0; g.alpha = 1.f; g.accumulate = 0;
    g.drop_thr = 0; g.drop_key = 0; g.drop_scale = 1.f; g.mask = nullptr;
    return g;
}
int launch_gemm(hipStream_t st, const GemmArgs& g, int AL, int BL);
// out[i] = scale * sum_s slabs[s*stride + i]  (+ optional second slab set), i < n
int launch_reduce_slabs(hipStream_t st, const float* slabs, int nslab, long stride, long n, float scale, float* out);
// up to 24 small slab reductions in one launch
struct ReduceSegs {
    const float* src[24]; float* dst[24]; long end[24]; long stride[24]; int nslab[24]; int vec[24];
    int n; float scale;
    // a segment whose rows and pointers are 16-byte aligned is processed four elements per thread (end[] counts threads)
    void add(const float* s, int ns, long st, long cnt, float* d) {
        const bool v4 = (cnt % 4 == 0) && (st % 4 == 0) && ((uintptr_t)s % 16 == 0) && ((uintptr_t)d % 16 == 0);
        src[n] = s; nslab[n] = ns; stride[n] = st; dst[n] = d; vec[n] = v4 ? 1 : 0;
        end[n] = (n ? end[n - 1] : 0) + (v4 ? cnt / 4 : cnt); ++n;
    }
    // the segments of `o` (same scale) after this one's; false when they do not fit
    bool append(const ReduceSegs& o) {
        if (n + o.n > 24 || o.scale != scale) return false;
        for (int i = 0; i < o.n; ++i) {
            src[n] = o.src[i]; nslab[n] = o.nslab[i]; stride[n] = o.stride[i]; dst[n] = o.dst[i]; vec[n] = o.vec[i];
            end[n] = (n ? end[n - 1] : 0) + (o.end[i] - (i ? o.end[i - 1] : 0)); ++n;
        }
        return true;
    }
};
int launch_reduce_multi(hipStream_t st, ReduceSegs& sg);
#ifdef __HIPCC__
// torch.optim.Adam's update of one element (coupled L2 weight decay, bias correction folded into lr_over_bc1 / inv_sqrt_bc2), the
// operation order of torch's _single_tensor_adam.  FMA contraction is switched off for this body (HIP's default is
// -ffp-contract=fast, and __fmul_rn / __fadd_rn are plain operators that contract like any other): every operation rounds on its
// own, so the stand-alone Adam kernel (adam.hip) and the fold into the final reduction (gemm.hip) give the same bits whatever code
// surrounds the call.
__device__ __forceinline__ void adam_update1(float g, float& p, float& m, float& v, float lr_over_bc1, float inv_sqrt_bc2, float b1,
                                             float b2, float eps, float wd) {
#pragma clang fp contract(off)
    const float gr = g + wd * p;
    m = m + (1.f - b1) * (gr - m);                                       // lerp form, as torch does
    v = b2 * v + (1.f - b2) * gr * gr;
    p = p - lr_over_bc1 * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
}
#endif
// One embedding-bag request (glove.hip / glove_bag.h): out[r, :] = pool over the L tokens of row r (select form: of the first support
// row of class r % N of episode r / N) of table rows; a deferred one waits in the workspace for the next FuMI step (glove_flush
// launches it on its own)
struct GloveArgs {
    const int64_t* tok; int R, L; int64_t pad_id; const float* table; int V, E, mode; float* out; int* status;
    const int64_t* y_s; int N, S;
};
struct GlovePending { GloveArgs a; int on, vec; size_t lds; };
int glove_flush(fumi_ws* ws, hipStream_t st);
// A deferred Adam step (fumi_hip_adam_step_deferred): tensors by value, coefficients already folded (adam.hip)
struct AdamPending {
    int n, on;
    float* p[32]; const float* g[32]; float* m[32]; float* v[32]; long numel[32];
    float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd;
};
// The LAST launch of a training meta-step: the reduction `sg` (whose outputs are the gradients and the step's statistics) with,
// when the workspace holds them and every gradient tensor of the pending optimizer step is one whole segment of `sg`, the Adam
// update of each element right behind its gradient and the deferred publication of the statistics -- one launch instead of
// three, same arithmetic in the same order (bit-identical parameters).  Falls back to the separate launches otherwise.
int launch_reduce_multi_final(fumi_ws* ws, hipStream_t st, ReduceSegs& sg);
int launch_adam_pending(fumi_ws* ws, hipStream_t st);       // the plain Adam launch of a still pending deferred step (adam.hip)
// out[n] = scale * sum_m X[m*ld + n]
int launch_colsum(hipStream_t st, const float* X, int M, int N, long ld, float scale, float* out);
// several column sums (bias gradients) in TWO launches: partial sums of 128-row chunks for every job, then one
// reduce_multi over the chunks (fixed order: bit-reproducible)
struct ColsumJobs {
    const float* X[8]; float* out[8]; int M[8], N[8]; long ld[8]; int blk_end[8]; long part_off[8]; int n; long part_total;
    void add(const float* x, int m, int nn, long l, float* o) {
        X[n] = x; M[n] = m; N[n] = nn; ld[n] = l; out[n] = o;
        const int blocks = ((nn + 63) / 64) * ((m + 127) / 128);
        blk_end[n] = (n ? blk_end[n - 1] : 0) + blocks;
        part_off[n] = part_total; part_total += (long)((m + 127) / 128) * (((long)nn + 3) & ~3L);
        ++n;
    }
};
int launch_colsum_multi(hipStream_t st, const ColsumJobs& jobs, float* part /* jobs.part_total floats */,
                        const ReduceSegs* extra = nullptr);

// ------------------------------------------------------------------------------------------------------------
// the two shared passes over the wide inputs (xpanel.hip)
// ------------------------------------------------------------------------------------------------------------
// zero-copy episodes: rows of the meta-batch addressed through indices into an HBM-resident table (x_s / x_q unused)
struct XRows { const float* table; const int64_t* idx_s; const int64_t* idx_q; long n_rows; };
struct HyperFwdArgs;         // hyper_fwd.h: a split hypernetwork forward that can ride at the front of the forward X-panel launch
int launch_xpanel_fwd(hipStream_t st, int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q,
                      const float* W0, float* A0 /*[B,S+Qn,h0]*/, float* G /*[B,S+Qn,S]*/, const XRows* rows = nullptr,
                      const HyperFwdArgs* rider = nullptr, int* rider_done = nullptr /* set to 1 when the rider was launched */,
                      float* parts = nullptr /* [xpanel_fwd_ksplit(), B, S+Qn, h0] partial products of a split contraction */,
                      int* parts_unreduced = nullptr /* not NULL: the parts are NOT summed into A0; receives their number (0: A0 is final) */,
                      unsigned short* planes = nullptr /* xpanel_planes(): room for the column operand split once per launch (the fast forward needs it) */,
                      struct GlovePending* glove = nullptr /* a pending embedding bag: rides in the pre-split launch (or is launched first) */);
// true when launch_xpanel_fwd with these arguments takes the pre-split path (the launch that can carry a pending embedding bag)
bool xpanel_fwd_presplits(int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q, const float* W0, bool gram,
                          const struct XRows* rows, unsigned short* planes);
// room for W0 [h0, D] and the support rows [B, S, D] split into three bf16 planes each (grown on demand, owned by the workspace);
// NULL when the fast forward does not apply
unsigned short* xpanel_planes(fumi_ws* ws, int B, int S, int D, int h0);
int xpanel_fwd_ksplit(int B, int S, int Qn, int D, int h0, int with_gram);
int xpanel_bwd_nsplit(int B, int S, int Qn, int D, int h0, int* kchunk_out);
struct HyperBwdArgs;         // hyper_bwd.h: the hypernetwork backward, able to ride at the front of the backward X-panel launch
int launch_xpanel_bwd(hipStream_t st, int B, int S, int Qn, int D, int h0, const float* x_s, const float* x_q,
                      const float* Abar /*[B,S+Qn,h0]*/, float* slabs /*[nsplit,h0,D]*/, int kchunk, int nsplit,
                      const XRows* rows = nullptr, const HyperBwdArgs* rider = nullptr, int* rider_done = nullptr);
// sets FUMI_ST_LABEL_RANGE when an index is outside [0, n_rows) (the X-panel kernels clamp such an index to row 0)
int launch_index_range_check(hipStream_t st, const int64_t* idx, long n, long n_rows, int* status);

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
    XRows rows;                     // table != NULL: x_s / x_q are not used, rows come from the table (zero-copy episodes)
    float* logits_q; int64_t* preds_q; float* preds_f /* optional */; float* loss_b; float* acc_b;
    float* gW[FUMI_MAX_HIDDEN]; float* gb[FUMI_MAX_HIDDEN];   // outputs (scaled sums over episodes)
    float* head_bar;                // [B,N,H+1] d loss_b / d head_b   (unscaled, per episode)
    float* stats;                   // optional [2]: grad_scale * (sum_b loss_b, sum_b acc_b)
    float dropout_p;                // inner-loop dropout probability (0 = off) and the step's 64-bit seed
    unsigned long long seed;
    // optional cross-stream hooks (FuMI): `head` is produced on another stream -> wait for head_ready before the inner
    // loop; record after_reverse once head_bar is complete so its consumer can start beside the layer-0 gradient pass
    hipEvent_t head_ready, after_reverse;
    struct ReduceSegs* defer_reduce;            // not NULL: the final sums over episodes / slabs are appended here and the
                                                // caller launches them (one launch for the whole step) instead of run_episodes
    hipEvent_t inputs_ready;                    // recorded on the caller's stream BEFORE the first launch (fork point)
    const HyperFwdArgs* fwd_rider;              // not NULL: the producer of `head` as rider workgroups of xpanel_fwd (hyper_fwd.h);
    int (*fwd_rider_fallback)(void*);           // called (with hook_ctx) right after xpanel_fwd when that launch could not carry it
    const HyperBwdArgs* bwd_rider;              // not NULL: the consumer of `head_bar` as rider workgroups of xpanel_bwd (hyper_bwd.h);
    int (*bwd_rider_fallback)(void*);           // called (with hook_ctx2) right after xpanel_bwd when that launch could not carry it
    void* hook_ctx2;
    struct GlovePending* glove;                 // a pending embedding bag whose output the forward rider reads: carried by xpanel_fwd's first launch
    int (*after_xpanel_fwd)(void*); void* hook_ctx;   // host callback right after xpanel_fwd is enqueued: the producer of `head`
                                                // is launched there, so its host-side preparation does not delay the matrix pass
};
size_t episode_workspace_bytes(const EpisodeProblem& p);
int run_episodes(fumi_ws* ws, hipStream_t st, const EpisodeProblem& p);
// the two-launch form of the backward X-panel pass (query rows beside the reverse sweep, support rows behind it): slabs of each part
bool xpanel_bwd_two_part_ok(int D, int h0);
void xpanel_bwd_two_part_split(int B, int S, int Qn, int D, int h0, int* nsq, int* kcq, int* nss, int* kcs);

// MAML with a bare linear head (linhead.hip)
size_t linhead_lds_floats(int N, int S, int Qn, int T, int taped);
int launch_linhead(hipStream_t st, int B, int N, int S, int Qn, int T, float alpha, int need_grad, int second_order,
                   const float* A, const float* G, const float* bias, const int64_t* y_s, const int64_t* y_q, float* logits_q,
                   int64_t* preds_q, float* preds_f, float* loss_b, float* acc_b, float* Abar, float* bbar, int* status);

// hypernetwork as LDS-resident kernels (hyper.hip); FUMI_ENOTSUP when the shapes do not fit (callers fall back to GEMMs)
int hyper_lds_fits(int R, int Dt, int Ht, int H1);
size_t hyper_bwd_workspace_floats(int R, int Ht, int H1);
constexpr int FUMI_ACNT = 16384;       // episodes per call the split adapt kernel has counters for
constexpr int FUMI_HCNT = 1024;        // arrival counters of the split hypernetwork forward (one per 16-row block)
size_t hyper_fwd_workspace_floats(int R, int Ht, int H1);
int launch_hyper_fwd(hipStream_t st, int R, int Dt, int Ht, int H1, int tanh_head, const float* c, const float* A0,
                     const float* b0, const float* A1, const float* b1, float* u, float* h, float* hpart = nullptr,
                     int* cnt = nullptr);
int launch_hyper_bwd(hipStream_t st, int R, int Dt, int Ht, int H1, int tanh_head, float scale, const float* c,
                     const float* u, const float* h, const float* hbar, const float* A1, float* ub, float* part,
                     float* gA0, float* gb0, float* gA1, float* gb1, ReduceSegs* defer = nullptr);

// small kernels shared by the entry points (episode.hip)
int launch_class_text_select(hipStream_t st, int B, int N, int S, int Dt, const float* text_s, const int64_t* y_s,
                             float* out, int* status);
int launch_broadcast_head(hipStream_t st, int B, int N, int H, const float* Wf, const float* bf, float* head);
int launch_tanh_bwd(hipStream_t st, long n, const float* h, const float* hbar, float* out);
int launch_relu_mask_mul(hipStream_t st, long n, const float* u, float* g_inout, float scale = 1.f);
int launch_split_head_grad(hipStream_t st, int B, int N, int H, const float* head_bar, float scale, float* gW, float* gb);

// ------------------------------------------------------------------------------------------------------------
// Staging plan of an LDS-resident kernel: a list of dense [rows x cols] copies global -> LDS image, built on the host
// (all shapes and alignments are known there) and passed as a kernel argument.  A job's source is
// base + b*sb + tile*st + part*sc (episode / tile-or-step / column part of the workgroup; element strides); rows < 0
// means "the tile's row count" (last tile may be short).  nsum > 1 sums nsum source slabs sstride apart while
// copying (the per-tile partial sums of the query pass).  The copy unit is one wave-wide load instruction: 64 lanes x 16
// bytes (x 4 bytes when a row is not 16-byte aligned) covering one 64-lane segment of a wide row, or 64 >> lg whole rows of
// a narrow one (2^lg lanes per row); units are numbered job after job (ubase).
// ------------------------------------------------------------------------------------------------------------
constexpr int WG_MAXJOB = 24;
struct StageTab {
    const float* base[WG_MAXJOB];
    int sb[WG_MAXJOB], st[WG_MAXJOB], sc[WG_MAXJOB], rs[WG_MAXJOB], off[WG_MAXJOB], sstride[WG_MAXJOB];
    short rows[WG_MAXJOB], cols[WG_MAXJOB], ld[WG_MAXJOB];
    unsigned char vec[WG_MAXJOB], segs[WG_MAXJOB], nsum[WG_MAXJOB], lg[WG_MAXJOB];
    short ubase[WG_MAXJOB + 1];
    short njobs, nunits;
    int bad;                   // a field did not fit its type: the caller must take the generic kernel
    void init() { njobs = 0; nunits = 0; bad = 0; }
    // rows_: count, or -1 / -2 = the kernel's run-time nr / nr2 (at most max_rows); cols_: count, or -max = run-time nc (<= max)
    void add(const float* b_, long sb_, long st_, long sc_, long rs_, int rows_, int max_rows, int cols_, int off_, int ld_,
             int nsum_ = 1, long sstride_ = 0) {
        const int max_cols = cols_ < 0 ? -cols_ : cols_;
        const bool dyn_cols = cols_ < 0;
        if (njobs >= WG_MAXJOB) { bad = 1; return; }
        const int j = njobs++;
        auto fits = [](long v) { return v >= 0 && v < (1L << 31); };
        if (!fits(sb_) || !fits(st_) || !fits(sc_) || !fits(rs_) || !fits(sstride_) || rows_ > 32767 || max_cols > 32767 ||
            ld_ > 32767 || nsum_ > 255) bad = 1;
        base[j] = b_; sb[j] = (int)sb_; st[j] = (int)st_; sc[j] = (int)sc_; rs[j] = (int)rs_; off[j] = off_;
        sstride[j] = (int)sstride_; rows[j] = (short)rows_; cols[j] = (short)(dyn_cols ? -1 : cols_); ld[j] = (short)ld_; nsum[j] = (unsigned char)nsum_;
        vec[j] = ((rs_ & 3) == 0) && ((sb_ & 3) == 0) && ((st_ & 3) == 0) && ((sc_ & 3) == 0) && ((sstride_ & 3) == 0) &&
                 ((max_cols & 3) == 0) && ((((uintptr_t)b_) & 15) == 0);   // (run-time widths: the caller keeps them multiples of 4)
        // lanes a row needs (one float4 or one float each), rounded to a power of two: a unit is 64 / that many rows,
        // or one 64-lane segment of a row that needs more
        const int need = vec[j] ? (max_cols + 3) / 4 : max_cols;
        int lgv = 0; while ((1 << lgv) < need && lgv < 6) ++lgv;
        lg[j] = (unsigned char)lgv;
        const int sg = (need + 63) / 64;
        const int nu = sg > 1 ? max_rows * sg : (max_rows + (64 >> lgv) - 1) / (64 >> lgv);
        if (sg > 255 || nunits + nu > 32767) bad = 1;
        segs[j] = (unsigned char)sg;
        ubase[j] = nunits; nunits = (short)(nunits + nu); ubase[j + 1] = nunits;
    }
};

#ifdef __HIPCC__
// ------------------------------------------------------------------------------------------------------------
// wg_mm: workgroup-cooperative small matrix product on the f32 MFMA, operands staged through LDS.
//   for m<M, n<N:  epi(m, n, sum_k A(m,k) * B(k,n)),   A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn]
//
// The per-episode matrices ([S|32, <=256] activations, [64,256] fast weights) live in L2; reading MFMA fragments
// straight from there made every 4-MFMA step a dependent L2 round trip.  Instead the whole workgroup copies the two
// operands (or a K-chunk of them when they do not fit `lds_cap` floats) into LDS with coalesced 16-byte loads -- one
// L2 latency per product -- and the waves then run v_mfma_f32_16x16x4_f32 from LDS (lane l feeds A[l&15][k] and
// B[k][l&15] with k = kk + 4*(l>>4) + j for the j-th MFMA of a 16-deep step; result register r of lane l is row
// 4*(l>>4)+r, column l&15).  Images are zero-padded to multiples of 16 so no operand needs a bounds check.
//   operand contiguous along k  -> image [row][kp+4]: one ds_read_b128 gives the 4 k values of a step
//   operand contiguous along m/n-> image [k][dim+4] : 4 ds_read_b32, conflict-free (row stride = 4 mod 8 floats)
// Every thread of the block must call it (it contains __syncthreads); products whose outputs feed each other still
// need the caller's __syncthreads() in between, exactly as before.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wg_r16(int x) { return (x + 15) & ~15; }

// copy a [rows x cols] matrix whose cols are contiguous in memory (row stride rs) into img[r*ld + c], zero-padded to
// [rows_p x cols_p]
__device__ __forceinline__ void wg_stage(float* img, int ld, int rows, int rows_p, int cols, int cols_p,
                                         const float* src, long rs) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool vec = ((rs & 3) == 0) && ((cols & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
    if (vec) {
        // 4 independent 16-byte loads in flight per thread before the first LDS write: one L2 latency per 4 elements
        const int c4n = cols_p >> 2, tot = rows_p * c4n;
        for (int i0 = tid; i0 < tot; i0 += 4 * nt) {
            f32x4 v[4]; int off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nt;
                const int r = i / c4n, c = (i - r * c4n) << 2;
                off[u] = i < tot ? r * ld + c : -1;
                const bool ok = i < tot && r < rows && c < cols;
                const float* q = src + (long)(ok ? r : 0) * rs + (ok ? c : 0);
                const f32x4 t = *(const f32x4*)q;
                v[u] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (off[u] >= 0) *(f32x4*)(img + off[u]) = v[u];
        }
    } else {
        for (int i = tid; i < rows_p * cols_p; i += nt) {
            const int r = i / cols_p, c = i - r * cols_p;
            img[r * ld + c] = (r < rows && c < cols) ? src[(long)r * rs + c] : 0.f;
        }
    }
}

// fallback for operands with no unit stride: fragments straight from memory (correct for any strides, slow)
template <class Epi>
__device__ __forceinline__ void wg_mm_direct(int M, int N, int K, const float* A, long sam, long sak,
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
        for (int k = 0; k < K; k += 4) {
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

// wg_mm2: as wg_mm with a two-part epilogue: pre(m,n) -> small POD is evaluated for all of a lane's output elements BEFORE
// the MFMA loop (its global loads -- masks, old values, bias terms -- are all in flight together and hidden behind the
// product), post(m, n, acc, pre_value) consumes them afterwards.
template <class Pre, class Post>
__device__ __forceinline__ void wg_mm2(float* lds, int lds_cap, int M, int N, int K, const float* A, long sam, long sak,
                                       const float* B, long sbk, long sbn, Pre&& pre, Post&& post) {
    constexpr int TPW = 2;                       // output tiles a wave accumulates at once
    const bool a_kc = sak == 1, b_kc = sbk == 1;
    auto epi = [&](int m, int n, float acc) { post(m, n, acc, pre(m, n)); };
    if ((!a_kc && sam != 1) || (!b_kc && sbn != 1) || M <= 0 || N <= 0) {
        wg_mm_direct(M, N, K, A, sam, sak, B, sbk, sbn, epi);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int Mp = wg_r16(M), Np = wg_r16(N);
    const int tm = Mp >> 4, tn = Np >> 4, ntiles = tm * tn;
    // K-chunk that fits: A image + B image <= lds_cap floats
    auto need = [&](int kp) { return (a_kc ? Mp * (kp + 4) : kp * (Mp + 4)) + (b_kc ? Np * (kp + 4) : kp * (Np + 4)); };
    int kc = wg_r16(K > 0 ? K : 1);
    while (kc > 16 && need(kc) > lds_cap) kc = wg_r16(kc >> 1);
    if (need(kc) > lds_cap) { wg_mm_direct(M, N, K, A, sam, sak, B, sbk, sbn, epi); return; }
    const int lda = a_kc ? kc + 4 : Mp + 4, ldb = b_kc ? kc + 4 : Np + 4;
    float* Ai = lds;
    float* Bi = lds + (a_kc ? Mp * lda : kc * lda);

    // operands that fit in one chunk are staged once, before the passes over the output tiles
    const bool single = kc >= K;
    if (single && K > 0) {
        const int kp = wg_r16(K);
        __syncthreads();                                 // previous users of the LDS images are done
        if (a_kc) wg_stage(Ai, lda, M, Mp, K, kp, A, sam);
        else      wg_stage(Ai, lda, K, kp, M, Mp, A, sak);
        if (b_kc) wg_stage(Bi, ldb, N, Np, K, kp, B, sbn);
        else      wg_stage(Bi, ldb, K, kp, N, Np, B, sbk);
        __syncthreads();
    }
    for (int base = 0; base < ntiles; base += nw * TPW) {
        using PV = decltype(pre(0, 0));              // any small POD (float, or a pair of floats)
        f32x4 acc[TPW];
        PV pv[TPW][4];
        int tm0[TPW], tn0[TPW];
        // Branch-free on purpose: tiles past the end are clamped to the last tile (their results are dropped), so the
        // TPW*4 pre-loads sit in ONE basic block and are all in flight together.  hipcc otherwise gives every guarded
        // load its own block with a wait: dependent L2 round trips.
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = min(base + wave + nw * i, ntiles - 1);
            tm0[i] = (t / tn) << 4; tn0[i] = (t % tn) << 4;
            acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[i][e] = pre(min(tm0[i] + q * 4 + e, M - 1), min(tn0[i] + r, N - 1));
        for (int k0 = 0; k0 < K; k0 += kc) {
            const int kv = min(kc, K - k0);              // valid depth of this chunk
            const int kp = wg_r16(kv);
            if (!single) {
                __syncthreads();
                if (a_kc) wg_stage(Ai, lda, M, Mp, kv, kp, A + k0, sam);
                else      wg_stage(Ai, lda, kv, kp, M, Mp, A + (long)k0 * sak, sak);
                if (b_kc) wg_stage(Bi, ldb, N, Np, kv, kp, B + k0, sbn);
                else      wg_stage(Bi, ldb, kv, kp, N, Np, B + (long)k0 * sbk, sbk);
                __syncthreads();
            }
            for (int kk = 0; kk < kp; kk += 16) {        // the TPW tiles' fragment reads are independent: all in flight
                f32x4 av[TPW], bv[TPW];
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const float* ap = a_kc ? Ai + (tm0[i] + r) * lda + 4 * q + kk : Ai + (4 * q + kk) * lda + tm0[i] + r;
                    const float* bp = b_kc ? Bi + (tn0[i] + r) * ldb + 4 * q + kk : Bi + (4 * q + kk) * ldb + tn0[i] + r;
                    if (a_kc) av[i] = *(const f32x4*)ap;
                    else { av[i][0] = ap[0]; av[i][1] = ap[lda]; av[i][2] = ap[2 * lda]; av[i][3] = ap[3 * lda]; }
                    if (b_kc) bv[i] = *(const f32x4*)bp;
                    else { bv[i][0] = bp[0]; bv[i][1] = bp[ldb]; bv[i][2] = bp[2 * ldb]; bv[i][3] = bp[3 * ldb]; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][e], bv[i][e], acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (base + wave + nw * i < ntiles) {         // wave-uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = tm0[i] + q * 4 + e, n = tn0[i] + r;
                    if (m < M && n < N) post(m, n, acc[i][e], pv[i][e]);
                }
            }
        }
    }
}

template <class Epi>
__device__ __forceinline__ void wg_mm(float* lds, int lds_cap, int M, int N, int K, const float* A, long sam, long sak,
                                      const float* B, long sbk, long sbn, Epi&& epi) {
    wg_mm2(lds, lds_cap, M, N, K, A, sam, sak, B, sbk, sbn, [](int, int) { return 0.f; },
           [&](int m, int n, float acc, float) { epi(m, n, acc); });
}

// dst[i] = f(i, a[i], b[i], c[i]) for i < n (unused inputs may be nullptr).  float4 accesses, two per input in flight per
// thread before any use: an elementwise loop written naively is one dependent memory round trip per iteration.
template <class F>
__device__ __forceinline__ void wg_ew(long n, float* dst, const float* a, const float* b, const float* c, F&& f) {
    const int tid = threadIdx.x, nt = blockDim.x;
    auto al = [](const void* p) { return p == nullptr || ((((uintptr_t)p) & 15) == 0); };
    if ((n & 3) == 0 && al(dst) && al(a) && al(b) && al(c)) {
        const long n4 = n >> 2;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        for (long i0 = tid; i0 < n4; i0 += 2L * nt) {
            const long i1 = i0 + nt;
            const bool two = i1 < n4;
            const long j1 = two ? i1 : i0;
            const f32x4 a0 = a ? *(const f32x4*)(a + 4 * i0) : z4, a1 = a ? *(const f32x4*)(a + 4 * j1) : z4;
            const f32x4 b0 = b ? *(const f32x4*)(b + 4 * i0) : z4, b1 = b ? *(const f32x4*)(b + 4 * j1) : z4;
            const f32x4 c0 = c ? *(const f32x4*)(c + 4 * i0) : z4, c1 = c ? *(const f32x4*)(c + 4 * j1) : z4;
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = f(4 * i0 + e, a0[e], b0[e], c0[e]); o1[e] = f(4 * j1 + e, a1[e], b1[e], c1[e]); }
            *(f32x4*)(dst + 4 * i0) = o0;
            if (two) *(f32x4*)(dst + 4 * i1) = o1;
        }
    } else {
        for (long i = tid; i < n; i += nt) dst[i] = f(i, a ? a[i] : 0.f, b ? b[i] : 0.f, c ? c[i] : 0.f);
    }
}

// dst[i] = sum_t src[t*stride + i], i < n: several independent 16-byte loads in flight per thread (a plain loop is a
// chain of dependent-latency iterations)
__device__ __forceinline__ void wg_sum_slabs(float* dst, const float* src, int nslab, long stride, long n) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool vec = ((n & 3) == 0) && ((stride & 3) == 0) && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 15) == 0);
    if (vec) {
        const long n4 = n >> 2;
        for (long i0 = tid; i0 < n4; i0 += 2L * nt) {
            const long i1 = i0 + nt;
            const bool two = i1 < n4;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            int t = 0;
            for (; t + 1 < nslab; t += 2) {
                const f32x4 a0 = *(const f32x4*)(src + t * stride + 4 * i0);
                const f32x4 a1 = *(const f32x4*)(src + (t + 1) * stride + 4 * i0);
                const f32x4 b0 = *(const f32x4*)(src + t * stride + 4 * (two ? i1 : i0));
                const f32x4 b1 = *(const f32x4*)(src + (t + 1) * stride + 4 * (two ? i1 : i0));
                s0 += a0; s0 += a1; s1 += b0; s1 += b1;
            }
            if (t < nslab) {
                s0 += *(const f32x4*)(src + t * stride + 4 * i0);
                s1 += *(const f32x4*)(src + t * stride + 4 * (two ? i1 : i0));
            }
            *(f32x4*)(dst + 4 * i0) = s0;
            if (two) *(f32x4*)(dst + 4 * i1) = s1;
        }
    } else {
        for (long i = tid; i < n; i += nt) {
            float s_ = 0.f;
            for (int t = 0; t < nslab; ++t) s_ += src[t * stride + i];
            dst[i] = s_;
        }
    }
}

// dst[i] = src[i], i < n (4 independent 16-byte loads in flight per thread)
__device__ __forceinline__ void wg_copy(float* dst, const float* src, long n) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool vec = ((n & 3) == 0) && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 15) == 0);
    if (vec) {
        const long n4 = n >> 2;
        for (long i0 = tid; i0 < n4; i0 += 4L * nt) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long i = i0 + (long)u * nt; v[u] = *(const f32x4*)(src + 4 * (i < n4 ? i : i0)); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long i = i0 + (long)u * nt; if (i < n4) *(f32x4*)(dst + 4 * i) = v[u]; }
        }
    } else {
        for (long i = tid; i < n; i += nt) dst[i] = src[i];
    }
}

// column sums of X[M,N] (row stride ld): f(n, sum_m X[m,n]).  All threads take part: the rows are split over
// blockDim/N (at most 16) row groups whose partial sums meet in LDS, so a thread has only a few independent loads in
// flight instead of M dependent-latency iterations.  Contains __syncthreads; `lds` needs 16*N floats at most.
template <class F>
__device__ __forceinline__ void wg_colsum(float* lds, int lds_cap, int M, int N, const float* X, long ld, F&& f) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int parts = nt / (N > 0 ? N : 1);
    if (parts > 16) parts = 16;
    if (parts > M) parts = M;
    if (parts < 2 || parts * N > lds_cap) {
        for (int n = tid; n < N; n += nt) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int m = 0;
            for (; m + 3 < M; m += 4) {
                s0 += X[(long)m * ld + n]; s1 += X[(long)(m + 1) * ld + n];
                s2 += X[(long)(m + 2) * ld + n]; s3 += X[(long)(m + 3) * ld + n];
            }
            for (; m < M; ++m) s0 += X[(long)m * ld + n];
            f(n, (s0 + s1) + (s2 + s3));
        }
        return;
    }
    __syncthreads();
    if (tid < parts * N) {
        const int part = tid / N, n = tid - part * N;
        float s0 = 0.f, s1 = 0.f;
        int m = part;
        for (; m + parts < M; m += 2 * parts) { s0 += X[(long)m * ld + n]; s1 += X[(long)(m + parts) * ld + n]; }
        if (m < M) s0 += X[(long)m * ld + n];
        lds[part * N + n] = s0 + s1;
    }
    __syncthreads();
    for (int n = tid; n < N; n += nt) {
        float s = 0.f;
        for (int pp = 0; pp < parts; ++pp) s += lds[pp * N + n];
        f(n, s);
    }
}
// ------------------------------------------------------------------------------------------------------------
// LDS-resident products.  The per-episode kernels whose working set fits the CU's 160 KiB keep EVERY matrix of a
// phase chain in LDS: operands are staged once at kernel start, products read and write LDS images only, and global
// memory sees just the final outputs.  (With global intermediates every phase cost two dependent L2 round trips --
// operand staging and the drain of the epilogue's stores at the barrier -- about 4-5 us of a 32-row product that
// needs 1 us of MFMA.)
//   image: row-major [rows][ld], ld = wg_ld(cols) = 4 (mod 32) floats: a k-contiguous operand is read with one
//   ds_read_b128 per 16-deep step, a k-major one with 4 ds_read_b32, both conflict-free.  The arena is zeroed once at
//   kernel start: rows/cols past the logical shape stay zero (K padding must be), M/N padding is never written.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int wg_ld(int cols) { return ((cols + 31) & ~31) + 4; }

// executes a StageTab whose copy `Tl` sits in LDS (wg_stage_tab_to_lds).  The units go round-robin over the waves
// (unit u -> wave u % nw).  Step A: every LANE works out one of its wave's units (job search, row, source row pointer,
// LDS offset) -- 64 descriptors in parallel, one short dependent chain.  Step B: the wave walks its units, U wave-wide
// loads in flight before the first LDS write, fetching each unit's descriptor from the owning lane with v_readlane;
// loads are unconditional from clamped addresses (a guarded load is a basic block of its own with a full wait).
// Needs nunits <= 64 * (waves per workgroup).
__device__ __forceinline__ void wg_stage_tab_to_lds(StageTab* Tl, int ntab = 1, int kernarg_bytes = 0) {
    // the plans must be the kernel's FIRST arguments: they are read straight from the kernarg segment (taking the
    // address of a by-value argument would copy it to scratch)
    const int* src = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
    int* dst = (int*)Tl;
    for (int i = threadIdx.x; i < ntab * (int)(sizeof(StageTab) / 4); i += blockDim.x) dst[i] = src[i];
    // The rest of the argument block (dimension / buffer / layout structs, ~1-2 KB after the plans) is read by scalar loads
    // scattered over the kernel's phases.  The block is a fresh piece of device memory for every launch, so each 64-byte
    // line costs a memory round trip where it is first touched (the ISA shows 18 s_load + s_waitcnt pairs in one 4 us phase
    // of the adapt kernel).  Touch every line now, in parallel with the plan copy: the scalar loads then hit L2.
    if (kernarg_bytes > 0) {
        const int first = ntab * (int)sizeof(StageTab) / 64, last = (kernarg_bytes + 63) / 64;
        const int ln = first + (int)threadIdx.x;
        if (ln < last) {
            int dummy;
            asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"((const char*)src + (long)ln * 64) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(dummy));
        }
    }
}
static_assert(sizeof(StageTab) % 8 == 0, "consecutive StageTab kernel arguments must be contiguous");
// U units in flight per wave, each the sum of up to NS slabs (NS = 1: plain copy)
template <int U, int NS = 1>
__device__ __forceinline__ void wg_stage_rows(const StageTab* Tl, long b, long tile, long part, int nr, float* lds,
                                              int nr2 = 0, int nc = 0) {
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nunits = Tl->nunits, njobs = Tl->njobs;
    // ---- A: lane l describes unit wave + nw*l
    const int myu = wave + nw * lane;
    const bool live = myu < nunits;
    const int u = live ? myu : nunits - 1;
    int j = 0;
    for (int jj = 1; jj < njobs; ++jj) if (u >= Tl->ubase[jj]) j = jj;
    const int ru = u - Tl->ubase[j], segs = Tl->segs[j], lgv = Tl->lg[j];
    const int r0 = segs > 1 ? ru / segs : ru << (6 - lgv);            // first row of the unit
    const int sg = segs > 1 ? ru - r0 * segs : 0;
    const int rows = Tl->rows[j] == -1 ? nr : Tl->rows[j] == -2 ? nr2 : Tl->rows[j];     // run-time row / column counts
    const bool rok = live && r0 < rows;
    const float* d_src = Tl->base[j] + b * Tl->sb[j] + tile * Tl->st[j] + part * Tl->sc[j] + (long)(rok ? r0 : 0) * Tl->rs[j];
    const int d_off = Tl->off[j] + r0 * Tl->ld[j];
    const int d_left = rok ? rows - r0 : 0;                           // rows from r0 on (a unit takes at most 64 >> lg)
    const int d_cols = Tl->cols[j] < 0 ? nc : Tl->cols[j];
    const int d_misc = (sg << 6) | (lgv << 24) | (Tl->vec[j] ? 1 << 30 : 0);
    const int d_rs = Tl->rs[j], d_ld = Tl->ld[j];
    const int d_ns = Tl->nsum[j], d_ss = Tl->sstride[j];
    const unsigned d_lo = (unsigned)(uintptr_t)d_src, d_hi = (unsigned)((uintptr_t)d_src >> 32);
    // ---- B
    const int mine = (nunits - wave + nw - 1) / nw;                  // units of this wave (uniform)
    for (int x0 = 0; x0 < mine; x0 += U) {
        f32x4 v[U]; int off[U]; int md[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const int l = min(x0 + x, 63);
            const unsigned lo = __builtin_amdgcn_readlane(d_lo, l), hi = __builtin_amdgcn_readlane(d_hi, l);
            const int o = __builtin_amdgcn_readlane(d_off, l), cols = __builtin_amdgcn_readlane(d_cols, l);
            const int left = __builtin_amdgcn_readlane(d_left, l), misc = __builtin_amdgcn_readlane(d_misc, l);
            const int rs = __builtin_amdgcn_readlane(d_rs, l), ld = __builtin_amdgcn_readlane(d_ld, l);
            const bool isv = (misc >> 30) & 1;
            const int lgu = (misc >> 24) & 7;
            const int i = lane >> lgu;                               // row of the unit this lane copies
            const int cb = (misc & 0xffffff) + (lane & ((1 << lgu) - 1));
            const int c = isv ? cb << 2 : cb;
            const bool ok = (x0 + x < mine) && i < left && c < cols;
            const float* p = (const float*)(((uintptr_t)hi << 32) | lo) + (ok ? (long)i * rs + c : 0);
            const float* p4 = (const float*)((uintptr_t)p & ~(uintptr_t)15);   // aligned float4 holding *p (vec: p itself)
            const int sub = (int)(p - p4);
            f32x4 t = *(const f32x4*)p4;
            if (NS > 1) {
                const int ns = __builtin_amdgcn_readlane(d_ns, l), ss = __builtin_amdgcn_readlane(d_ss, l);
                f32x4 tk[NS];
#pragma unroll
                for (int k = 1; k < NS; ++k) tk[k] = *(const f32x4*)(p4 + (long)(k < ns ? k : 0) * ss);
#pragma unroll
                for (int k = 1; k < NS; ++k) if (k < ns) t += tk[k];          // (scalar jobs: sstride keeps the sub-offset)
            }
            v[x] = t;
            if (!isv) v[x][0] = sub == 0 ? t[0] : sub == 1 ? t[1] : sub == 2 ? t[2] : t[3];
            off[x] = o + i * ld + c;
            md[x] = ok ? (isv ? 2 : 1) : 0;
        }
#pragma unroll
        for (int x = 0; x < U; ++x) {
            if (md[x] == 2) *(f32x4*)(lds + off[x]) = v[x];
            else if (md[x] == 1) lds[off[x]] = v[x][0];
        }
    }
}

// barrier between two phases that only exchange LDS data: waits for the wave's LDS traffic, NOT for its global stores
// (__syncthreads drains vmcnt too, i.e. every phase that writes results to memory would pay the store latency)
__device__ __forceinline__ void wg_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Producer side of a cross-workgroup hand-off through sc1 (agent-scope, write-through) stores: EVERY storing wave waits for
// its own stores before the workgroup barrier behind which one lane bumps the arrival counter.  A __syncthreads() alone is a
// workgroup-scope release: on gfx950 it waits for lgkmcnt only, and the counter's atomic can overtake stores that are still
// in flight to another L2 channel (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility", valid forms).
// Inline asm, because hipcc may drop a wait it believes redundant.  Readers use sc1 loads after the counter add has returned.
__device__ __forceinline__ void wg_drain_stores() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// store the first cnt (1..4) elements of v at p: one 16-byte store when complete and aligned
__device__ __forceinline__ void wg_st4(float* p, const f32x4& v, int cnt) {
    if (cnt == 4 && ((((uintptr_t)p) & 15) == 0)) *(f32x4*)p = v;
    else { for (int e = 0; e < cnt; ++e) p[e] = v[e]; }
}

// epi(m, n, acc4, cnt) for m < M and n = 0, 4, 8, ... < N:  acc4[e] = sum_k A(m,k) B(k,n+e), cnt = min(4, N-n) valid
// elements (the others are 0).  Both operands are LDS images:
//   AKC: A(m,k) = A[m*lda + k]   else A(m,k) = A[k*lda + m]      BKC: B(k,n) = B[n*ldb + k]   else B(k,n) = B[k*ldb + n]
// 16x16 output tiles round-robin over the waves, computed TRANSPOSED (the MFMA's row operand is B, its column operand
// A): lane l then holds row m0 + (l&15), columns n0 + 4*(l>>4) .. +3 -- four consecutive floats of one output row, so
// an epilogue is one 16-byte LDS or global access per lane instead of four scalar ones.  Every wave keeps two
// accumulation chains going: two tiles when there are enough, otherwise the even / odd 16-deep steps of one tile.
// K is walked in 16-deep steps (lane l feeds k = kk + 4*(l>>4) + j to the j-th MFMA; the next step's fragments are read
// before the current step's MFMAs) and a 4-deep tail (k = kk + (l>>4)); images must be zero for k in [K, K+3].
template <bool AKC, bool BKC, class Epi>
__device__ __forceinline__ void wg_lmm(int M, int N, int K, const float* A, int lda, const float* B, int ldb, Epi&& epi) {
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform on purpose: scalar branches
    const int r = lane & 15, q = lane >> 4;
    const int tn = (N + 15) >> 4, ntiles = ((M + 15) >> 4) * tn;
    const int nsteps = K >> 4;
    const bool ksplit = ntiles <= nw;
    const int kst = ksplit ? 32 : 16, kofs1 = ksplit ? 16 : 0;
    const int n_it = ksplit ? nsteps >> 1 : nsteps;
    const int sa = AKC ? 1 : lda, sb = BKC ? 1 : ldb;            // address step per k
    struct Frag { f32x4 a0, b0, a1, b1; };
    for (int t0 = wave; t0 < ntiles; t0 += ksplit ? nw : 2 * nw) {
        const bool two = !ksplit && t0 + nw < ntiles;
        const int t1 = two ? t0 + nw : t0;
        const int m00 = (t0 / tn) << 4, n00 = (t0 % tn) << 4, m01 = (t1 / tn) << 4, n01 = (t1 % tn) << 4;
        const float* ap0 = AKC ? A + (m00 + r) * lda + 4 * q : A + (4 * q) * lda + m00 + r;
        const float* bp0 = BKC ? B + (n00 + r) * ldb + 4 * q : B + (4 * q) * ldb + n00 + r;
        const float* ap1 = (AKC ? A + (m01 + r) * lda + 4 * q : A + (4 * q) * lda + m01 + r) + kofs1 * sa;
        const float* bp1 = (BKC ? B + (n01 + r) * ldb + 4 * q : B + (4 * q) * ldb + n01 + r) + kofs1 * sb;
        // no conditionals inside: a chain without a tile of its own repeats tile 0 and is dropped afterwards
        auto load = [&](int k) {
            Frag f;
            const float* pa0 = ap0 + k * sa; const float* pb0 = bp0 + k * sb;
            const float* pa1 = ap1 + k * sa; const float* pb1 = bp1 + k * sb;
            if (AKC) { f.a0 = *(const f32x4*)pa0; f.a1 = *(const f32x4*)pa1; }
            else { f.a0[0] = pa0[0]; f.a0[1] = pa0[lda]; f.a0[2] = pa0[2 * lda]; f.a0[3] = pa0[3 * lda];
                   f.a1[0] = pa1[0]; f.a1[1] = pa1[lda]; f.a1[2] = pa1[2 * lda]; f.a1[3] = pa1[3 * lda]; }
            if (BKC) { f.b0 = *(const f32x4*)pb0; f.b1 = *(const f32x4*)pb1; }
            else { f.b0[0] = pb0[0]; f.b0[1] = pb0[ldb]; f.b0[2] = pb0[2 * ldb]; f.b0[3] = pb0[3 * ldb];
                   f.b1[0] = pb1[0]; f.b1[1] = pb1[ldb]; f.b1[2] = pb1[2 * ldb]; f.b1[3] = pb1[3 * ldb]; }
            return f;
        };
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        auto mma = [&](const Frag& f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b0[e], f.a0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b1[e], f.a1[e], acc1, 0, 0, 0);
            }
        };
        if (n_it > 0) {
            // two register sets alternate; the next step's fragments are read before the current step's MFMAs issue
            Frag fa = load(0);
            int it = 0;
            for (; it + 1 < n_it; it += 2) {              // sched_barrier: keep the reads ahead of the MFMAs they overlap
                const Frag fb = load((it + 1) * kst);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa);
                __builtin_amdgcn_sched_barrier(0);
                fa = load(min(it + 2, n_it - 1) * kst);
                __builtin_amdgcn_sched_barrier(0);
                mma(fb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it < n_it) mma(fa);
        }
        if (ksplit && (nsteps & 1)) {                            // odd 16-deep step left over by the even/odd split
            const int k = (nsteps - 1) << 4;
            f32x4 av, bv;
            const float* pa = ap0 + k * sa; const float* pb = bp0 + k * sb;
            if (AKC) av = *(const f32x4*)pa; else { av[0] = pa[0]; av[1] = pa[lda]; av[2] = pa[2 * lda]; av[3] = pa[3 * lda]; }
            if (BKC) bv = *(const f32x4*)pb; else { bv[0] = pb[0]; bv[1] = pb[ldb]; bv[2] = pb[2 * ldb]; bv[3] = pb[3 * ldb]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[e], av[e], acc0, 0, 0, 0);
        }
        for (int kk = nsteps << 4; kk < K; kk += 4) {            // 4-deep tail, plain k = kk + q
            const float a0 = AKC ? A[(m00 + r) * lda + kk + q] : A[(kk + q) * lda + m00 + r];
            const float b0 = BKC ? B[(n00 + r) * ldb + kk + q] : B[(kk + q) * ldb + n00 + r];
            const float a1 = AKC ? A[(m01 + r) * lda + kk + q] : A[(kk + q) * lda + m01 + r];
            const float b1 = BKC ? B[(n01 + r) * ldb + kk + q] : B[(kk + q) * ldb + n01 + r];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, a0, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, a1, acc1, 0, 0, 0);
        }
        if (ksplit) acc0 += acc1;
        {
            const int m = m00 + r, n = n00 + 4 * q;
            if (m < M && n < N) {
                const int cnt = min(4, N - n);
#pragma unroll
                for (int e = 1; e < 4; ++e) if (e >= cnt) acc0[e] = 0.f;
                epi(m, n, acc0, cnt);
            }
        }
        if (two) {
            const int m = m01 + r, n = n01 + 4 * q;
            if (m < M && n < N) {
                const int cnt = min(4, N - n);
#pragma unroll
                for (int e = 1; e < 4; ++e) if (e >= cnt) acc1[e] = 0.f;
                epi(m, n, acc1, cnt);
            }
        }
    }
}

template <int V> struct WgInt { static constexpr int value = V; };

// The same product for a k-major B image (B(k,n) = B[k*ldb + n], rows of B contiguous along n), organised so that one
// ds_read_b128 of a B row feeds FOUR MFMAs: a wave owns a 16 x 64 output block made of four column tiles
// c = 0..3 = columns {n0 + 4*j + c}; the read of row k at columns n0 + 4*(l&15) .. +3 gives lane l its B value for each
// of them.  Lane l ends up with the 4x4 block rows m0 + 4*(l>>4) .. +3, columns n0 + 4*(l&15) .. +3, i.e. four 16-byte
// epilogue accesses.  Per 16-deep step: 1 (AKC) or 4 reads for A, 4 for B, 16 MFMAs -- a third of the LDS instructions
// wg_lmm needs for this layout.  epi(m, n, acc4, cnt, WgInt<e>) as in wg_lmm; e = m & 3 says which of the lane's four
// rows this is (a compile-time tag, so per-lane register state can be indexed with it).  Block t goes to wave t % nw.
template <bool AKC, class Epi>
__device__ __forceinline__ void wg_lmm_wide(int M, int N, int K, const float* A, int lda, const float* B, int ldb, Epi&& epi) {
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int tn = (N + 63) >> 6, nblk = ((M + 15) >> 4) * tn;
    const int nsteps = K >> 4;
    struct Frag { f32x4 a, b0, b1, b2, b3; };
    for (int t = wave; t < nblk; t += nw) {
        const int m0 = (t / tn) << 4, n0 = (t % tn) << 6;
        const float* ap = AKC ? A + (m0 + r) * lda + 4 * q : A + (4 * q) * lda + m0 + r;
        const float* bp = B + (4 * q) * ldb + n0 + 4 * r;
        auto load = [&](int k) {
            Frag f;
            const float* pa = ap + k * (AKC ? 1 : lda);
            if (AKC) f.a = *(const f32x4*)pa;
            else { f.a[0] = pa[0]; f.a[1] = pa[lda]; f.a[2] = pa[2 * lda]; f.a[3] = pa[3 * lda]; }
            const float* pb = bp + k * ldb;
            f.b0 = *(const f32x4*)pb; f.b1 = *(const f32x4*)(pb + ldb);
            f.b2 = *(const f32x4*)(pb + 2 * ldb); f.b3 = *(const f32x4*)(pb + 3 * ldb);
            return f;
        };
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto mma = [&](const Frag& f) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[0], f.b0[c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[1], f.b1[c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[2], f.b2[c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[3], f.b3[c], acc[c], 0, 0, 0);
            }
        };
        if (nsteps > 0) {
            Frag fa = load(0);
            int it = 0;
            for (; it + 1 < nsteps; it += 2) {
                const Frag fb = load((it + 1) << 4);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa);
                __builtin_amdgcn_sched_barrier(0);
                fa = load(min(it + 2, nsteps - 1) << 4);
                __builtin_amdgcn_sched_barrier(0);
                mma(fb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it < nsteps) mma(fa);
        }
        for (int kk = nsteps << 4; kk < K; kk += 4) {            // 4-deep tail, plain k = kk + q
            const float a = AKC ? A[(m0 + r) * lda + kk + q] : A[(kk + q) * lda + m0 + r];
            const f32x4 bv = *(const f32x4*)(B + (kk + q) * ldb + n0 + 4 * r);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[c], acc[c], 0, 0, 0);
        }
        const int n = n0 + 4 * r;
        if (n < N) {
            const int cnt = min(4, N - n);
            auto out = [&](auto ec) {                            // ec: which of the lane's 4 rows (compile-time)
                constexpr int e = decltype(ec)::value;
                const int m = m0 + 4 * q + e;
                if (m < M) {
                    f32x4 v = {acc[0][e], cnt > 1 ? acc[1][e] : 0.f, cnt > 2 ? acc[2][e] : 0.f, cnt > 3 ? acc[3][e] : 0.f};
                    epi(m, n, v, cnt, ec);
                }
            };
            out(WgInt<0>{}); out(WgInt<1>{}); out(WgInt<2>{}); out(WgInt<3>{});
        }
    }
}

// f(n, sum_{m<M} X[m*ld + n]) for n < N on an LDS image: one thread per column, 4 independent chains
template <class F>
__device__ __forceinline__ void wg_lcolsum(int M, int N, const float* X, int ld, F&& f) {
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int m = 0;
        for (; m + 3 < M; m += 4) { s0 += X[m * ld + n]; s1 += X[(m + 1) * ld + n]; s2 += X[(m + 2) * ld + n]; s3 += X[(m + 3) * ld + n]; }
        for (; m < M; ++m) s0 += X[m * ld + n];
        f(n, (s0 + s1) + (s2 + s3));
    }
}
#endif
