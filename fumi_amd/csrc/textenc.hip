// the two pieces of the reference's model zoo beside the episodic hot path.
//   * CLIP baseline (fumi/models/clip.py:11-41 forward, :96-108 training step): two Linear.ReLU.Linear towers, cosine
//     similarity of every (text, image) pair, symmetric cross-entropy against the diagonal.  Linears on the GEMM family
//     (gemm.hip); the normalisation, the two soft-maxes and their backward through the norms are the kernels here.
//   * bi-LSTM text encoders RNN / RnnHid (fumi/models/common.py:44-161): input projections of all tokens as one GEMM per
//     direction, then L dependent steps of [R,H] x [H,4H] + gates.  Frozen encoders (the default, fumi.py:65-67) use the
//     forward-only entry; --fine_tune uses the tape-keeping forward and back-propagation through time below it.
#include "common.h"
#include <string.h>

namespace {

// ---- CLIP -----------------------------------------------------------------------------------------------------------------
// row norms of latents [n, P]: one wave per row
__global__ void row_norm_kernel(int n, int P, const float* x, float* out) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    float s = 0.f;
    for (int k = lane; k < P; k += 64) { const float v = x[(long)row * P + k]; s += v * v; }
#pragma unroll
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = sqrtf(s);
}

// sim[i][j] = raw[i][j] / (a[i] b[j])
__global__ void clip_sim_kernel(int nt, int ni, const float* raw, const float* a, const float* b, float* sim) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nt * ni) return;
    sim[i] = raw[i] / a[i / ni] / b[i % ni];
}

// symmetric cross-entropy of sim [n, n] against the diagonal (clip.py:101-105), one workgroup:
//   loss = (mean_i (lse_row_i - s_ii) + mean_j (lse_col_j - s_jj)) / 2;   dsim = (P_row + P_col - 2 I) / (2 n)
// then, for the backward through sim = raw / (a b^T):  draw = dsim / (a b^T),  rt[i] = sum_j dsim_ij sim_ij / a_i^2,
// rc[j] = sum_i dsim_ij sim_ij / b_j^2  (d tl_i = sum_j draw_ij il_j - rt_i tl_i, d il_j likewise)
__global__ __launch_bounds__(256) void clip_loss_kernel(int n, const float* sim, const float* a, const float* b, float* loss,
                                                        float* dsim, float* draw, float* rt, float* rc, float* lse_r, float* lse_c) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int i = tid; i < n; i += 256) {                                  // row i and column i
        float mr = -1e30f, mc = -1e30f;
        for (int j = 0; j < n; ++j) { mr = fmaxf(mr, sim[(long)i * n + j]); mc = fmaxf(mc, sim[(long)j * n + i]); }
        float sr = 0.f, sc = 0.f;
        for (int j = 0; j < n; ++j) { sr += __expf(sim[(long)i * n + j] - mr); sc += __expf(sim[(long)j * n + i] - mc); }
        const float lr = mr + __logf(sr), lc = mc + __logf(sc);
        lse_r[i] = lr; lse_c[i] = lc;
        acc += (lr - sim[(long)i * n + i]) + (lc - sim[(long)i * n + i]);
    }
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) loss[0] = red[0] / (2.f * n);
    if (!dsim) return;
    __syncthreads();
    for (long e = tid; e < (long)n * n; e += 256) {
        const int i = (int)(e / n), j = (int)(e % n);
        const float s = sim[e];
        const float d = (__expf(s - lse_r[i]) + __expf(s - lse_c[j]) - (i == j ? 2.f : 0.f)) / (2.f * n);
        dsim[e] = d;
        draw[e] = d / (a[i] * b[j]);
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        float st = 0.f, sc = 0.f;
        for (int j = 0; j < n; ++j) { st += dsim[(long)i * n + j] * sim[(long)i * n + j]; sc += dsim[(long)j * n + i] * sim[(long)j * n + i]; }
        rt[i] = st / (a[i] * a[i]); rc[i] = sc / (b[i] * b[i]);
    }
}

// d[i][k] -= r[i] * x[i][k]
__global__ void row_axpy_kernel(int n, int P, const float* r, const float* x, float* d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * P) return;
    d[i] -= r[i / P] * x[i];
}

// ---- LSTM -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
// one time step of one direction for every row: gates = pre[r][t] (x W_ih^T + b_ih, gate order i f g o) + rec[r] (h W_hh^T) + b_hh
__global__ void lstm_gate_kernel(int R, int H, int L, int t, const int64_t* tok, int64_t pad, const float* pre, const float* rec,
                                 const float* b_hh, float* c, float* h) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)R * H) return;
    const int r = (int)(id / H), j = (int)(id - (long)r * H);
    int len = 0;
    for (int k = 0; k < L; ++k) len += tok[(long)r * L + k] != pad;        // non-PAD count (common.py:90-91)
    if (t >= len) return;                                                  // packed sequence: the row has ended / not begun
    const float* p = pre + ((long)r * L + t) * 4 * H;
    const float* q = rec + (long)r * 4 * H;
    const float gi = p[j] + q[j] + b_hh[j], gf = p[H + j] + q[H + j] + b_hh[H + j];
    const float gg = p[2 * H + j] + q[2 * H + j] + b_hh[2 * H + j], go = p[3 * H + j] + q[3 * H + j] + b_hh[3 * H + j];
    const float cn = sigm(gf) * c[id] + sigm(gi) * tanhf(gg);
    c[id] = cn;
    h[id] = sigm(go) * tanhf(cn);
}

// the same step keeping what back-propagation needs, all indexed [r][t]: the four gate activations, the incoming h and c, tanh(c_t)
__global__ void lstm_gate_tape_kernel(int R, int H, int L, int t, const int64_t* tok, int64_t pad, const float* pre, const float* rec,
                                      const float* b_hh, float* c, float* h, float* gates, float* hprev, float* cprev, float* tcs) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)R * H) return;
    const int r = (int)(id / H), j = (int)(id - (long)r * H);
    int len = 0;
    for (int k = 0; k < L; ++k) len += tok[(long)r * L + k] != pad;
    if (t >= len) return;                                                  // (hprev of such steps stays at the memset's zeros)
    const long rt = (long)r * L + t;
    const float* p = pre + rt * 4 * H;
    const float* q = rec + (long)r * 4 * H;
    const float gi = sigm(p[j] + q[j] + b_hh[j]), gf = sigm(p[H + j] + q[H + j] + b_hh[H + j]);
    const float gg = tanhf(p[2 * H + j] + q[2 * H + j] + b_hh[2 * H + j]), go = sigm(p[3 * H + j] + q[3 * H + j] + b_hh[3 * H + j]);
    const float cp = c[id], cn = gf * cp + gi * gg, tc = tanhf(cn);
    float* gt = gates + rt * 4 * H;
    gt[j] = gi; gt[H + j] = gf; gt[2 * H + j] = gg; gt[3 * H + j] = go;
    hprev[rt * H + j] = h[id]; cprev[rt * H + j] = cp; tcs[rt * H + j] = tc;
    c[id] = cn;
    h[id] = go * tc;
}

// one step of back-propagation through time for one direction (dir 0 walks t = L-1..0, dir 1 walks t = 0..L-1).  A row is active
// at t < len; the adjoint of its state comes from the step processed just before (rec = dgates W_hh, and dc) or, at the row's
// first step of this walk, from the encoder's output adjoint d_out[r][dir*H + j] (on h for RNN, on c for RnnHid).
__global__ void lstm_gate_bwd_kernel(int R, int H, int L, int t, int dir, const int64_t* tok, int64_t pad, const float* gates,
                                     const float* cprev, const float* tcs, const float* d_out, int use_cell, const float* rec,
                                     float* dc, float* dgates) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)R * H) return;
    const int r = (int)(id / H), j = (int)(id - (long)r * H);
    int len = 0;
    for (int k = 0; k < L; ++k) len += tok[(long)r * L + k] != pad;
    const long rt = (long)r * L + t;
    float* dg = dgates + rt * 4 * H;
    if (t >= len) { dg[j] = 0.f; dg[H + j] = 0.f; dg[2 * H + j] = 0.f; dg[3 * H + j] = 0.f; return; }
    const bool chained = dir == 0 ? (t + 1 < len) : (t >= 1);
    const float seed = d_out[(long)r * 2 * H + dir * H + j];
    const float dh = chained ? rec[id] : (use_cell ? 0.f : seed);
    const float dcin = chained ? dc[id] : (use_cell ? seed : 0.f);
    const float* gt = gates + rt * 4 * H;
    const float gi = gt[j], gf = gt[H + j], gg = gt[2 * H + j], go = gt[3 * H + j];
    const float tc = tcs[rt * H + j], cp = cprev[rt * H + j];
    const float dct = dcin + dh * go * (1.f - tc * tc);
    dg[j] = dct * gg * gi * (1.f - gi);
    dg[H + j] = dct * cp * gf * (1.f - gf);
    dg[2 * H + j] = dct * gi * (1.f - gg * gg);
    dg[3 * H + j] = dh * tc * go * (1.f - go);
    dc[id] = dct * gf;
}

// out[r][d*H + j] = state_d[r][j]
__global__ void lstm_out_kernel(int R, int H, const float* s0, const float* s1, float* out) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)R * 2 * H) return;
    const int r = (int)(id / (2 * H)), k = (int)(id - (long)r * 2 * H);
    out[id] = k < H ? s0[(long)r * H + k] : s1[(long)r * H + k - H];
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

extern "C" {

int fumi_hip_clip_step(fumi_ws_t* ws, fumi_stream_t stream, int nt, int ni, int Dt, int D, int P,
        const float* text, const float* image, const float* const* w, int need_grad,
        float* sim, float* loss, float* const* g_w) {
    if (!ws || !text || !image || !w || !sim || nt < 1 || ni < 1 || Dt < 1 || D < 1 || P < 1) return FUMI_EINVAL;
    for (int i = 0; i < 8; ++i) if (!w[i] || (need_grad && (!g_w || !g_w[i]))) return FUMI_EINVAL;
    if ((need_grad || loss) && nt != ni) return FUMI_EINVAL;                // the loss pairs row i with column i
    if (need_grad && !loss) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    const size_t nn = (size_t)nt * ni;
    size_t bytes = 4 * ws_align((size_t)nt * P * 4) + 4 * ws_align((size_t)ni * P * 4) + 3 * ws_align(nn * 4) +
                   6 * ws_align((size_t)(nt > ni ? nt : ni) * 4);
    int rc = ws_reserve(ws, bytes);
    if (rc) return rc;
    float* t1 = ws_f(ws, (size_t)nt * P); float* tl = ws_f(ws, (size_t)nt * P); float* dtl = ws_f(ws, (size_t)nt * P); float* dt1 = ws_f(ws, (size_t)nt * P);
    float* i1 = ws_f(ws, (size_t)ni * P); float* il = ws_f(ws, (size_t)ni * P); float* dil = ws_f(ws, (size_t)ni * P); float* di1 = ws_f(ws, (size_t)ni * P);
    float* raw = ws_f(ws, nn); float* dsim = ws_f(ws, nn); float* draw = ws_f(ws, nn);
    const size_t nm = nt > ni ? nt : ni;
    float* a = ws_f(ws, nm); float* b = ws_f(ws, nm); float* rt = ws_f(ws, nm); float* rcc = ws_f(ws, nm); float* lr = ws_f(ws, nm); float* lc = ws_f(ws, nm);
    // towers (clip.py:29-30)
    GemmArgs g = gemm_args(nt, P, Dt, text, Dt, w[0], Dt, t1, P); g.bias = w[1]; g.act = 1;
    TRY(launch_gemm(st, g, 0, 0));
    g = gemm_args(nt, P, P, t1, P, w[2], P, tl, P); g.bias = w[3];
    TRY(launch_gemm(st, g, 0, 0));
    g = gemm_args(ni, P, D, image, D, w[4], D, i1, P); g.bias = w[5]; g.act = 1;
    TRY(launch_gemm(st, g, 0, 0));
    g = gemm_args(ni, P, P, i1, P, w[6], P, il, P); g.bias = w[7];
    TRY(launch_gemm(st, g, 0, 0));
    // cosine similarity (clip.py:32-41)
    hipLaunchKernelGGL(row_norm_kernel, dim3((nt + 3) / 4), dim3(256), 0, st, nt, P, tl, a); LAUNCH_CHECK();
    hipLaunchKernelGGL(row_norm_kernel, dim3((ni + 3) / 4), dim3(256), 0, st, ni, P, il, b); LAUNCH_CHECK();
    g = gemm_args(nt, ni, P, tl, P, il, P, raw, ni);
    TRY(launch_gemm(st, g, 0, 0));
    hipLaunchKernelGGL(clip_sim_kernel, dim3(nblk(nn)), dim3(256), 0, st, nt, ni, raw, a, b, sim); LAUNCH_CHECK();
    if (!loss) return FUMI_OK;
    hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(256), 0, st, nt, sim, a, b, loss, need_grad ? dsim : nullptr, draw, rt, rcc, lr, lc);
    LAUNCH_CHECK();
    if (!need_grad) return FUMI_OK;
    // through the normalisation: d tl = draw il - rt * tl,  d il = draw^T tl - rc * il
    g = gemm_args(nt, P, ni, draw, ni, il, P, dtl, P);
    TRY(launch_gemm(st, g, 0, 1));
    hipLaunchKernelGGL(row_axpy_kernel, dim3(nblk((long)nt * P)), dim3(256), 0, st, nt, P, rt, tl, dtl); LAUNCH_CHECK();
    g = gemm_args(ni, P, nt, draw, ni, tl, P, dil, P);
    TRY(launch_gemm(st, g, 1, 1));
    hipLaunchKernelGGL(row_axpy_kernel, dim3(nblk((long)ni * P)), dim3(256), 0, st, ni, P, rcc, il, dil); LAUNCH_CHECK();
    // the two towers' backward
    auto tower = [&](int n, int Din, const float* x, const float* h1, const float* dl, float* dh1, const float* W2,
                     float* gW0, float* gb0, float* gW2, float* gb2) -> int {
        GemmArgs q = gemm_args(P, P, n, dl, P, h1, P, gW2, P);                 // gW2 = dl^T h1
        TRY(launch_gemm(st, q, 1, 1));
        TRY(launch_colsum(st, dl, n, P, P, 1.f, gb2));
        q = gemm_args(n, P, P, dl, P, W2, P, dh1, P); q.mask = h1;              // dh1 = (dl W2) * relu'(h1)
        TRY(launch_gemm(st, q, 0, 1));
        q = gemm_args(P, Din, n, dh1, P, x, Din, gW0, Din);                     // gW0 = dh1^T x
        TRY(launch_gemm(st, q, 1, 1));
        return launch_colsum(st, dh1, n, P, P, 1.f, gb0);
    };
    TRY(tower(nt, Dt, text, t1, dtl, dt1, w[2], g_w[0], g_w[1], g_w[2], g_w[3]));
    return tower(ni, D, image, i1, dil, di1, w[6], g_w[4], g_w[5], g_w[6], g_w[7]);
}

int fumi_hip_lstm_bidir(fumi_ws_t* ws, fumi_stream_t stream, int R, int L, int E, int H,
        const int64_t* tokens, int64_t pad_id, const float* table, int64_t V, const float* const* w, int use_cell, float* out) {
    if (!ws || !tokens || !table || !w || !out || R < 1 || L < 1 || E < 1 || H < 1 || V < 1) return FUMI_EINVAL;
    for (int i = 0; i < 8; ++i) if (!w[i]) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    const size_t RL = (size_t)R * L;
    size_t bytes = ws_align(RL * E * 4) + 2 * ws_align(RL * 4 * H * 4) + ws_align((size_t)R * 4 * H * 4) + 4 * ws_align((size_t)R * H * 4);
    int rc = ws_reserve(ws, bytes);
    if (rc) return rc;
    float* x = ws_f(ws, RL * E);
    float* pre[2] = {ws_f(ws, RL * 4 * H), ws_f(ws, RL * 4 * H)};
    float* rec = ws_f(ws, (size_t)R * 4 * H);
    float* hs[2] = {ws_f(ws, (size_t)R * H), ws_f(ws, (size_t)R * H)};
    float* cs[2] = {ws_f(ws, (size_t)R * H), ws_f(ws, (size_t)R * H)};
    // embedding rows of every token (common.py:94), then both directions' input projections as two GEMMs
    TRY(fumi_hip_gather_rows(ws, stream, table, V, (int64_t)E * 4, tokens, (int64_t)RL, x));
    for (int d = 0; d < 2; ++d) {
        GemmArgs g = gemm_args((int)RL, 4 * H, E, x, E, w[4 * d], E, pre[d], 4 * H);
        g.bias = w[4 * d + 2];
        TRY(launch_gemm(st, g, 0, 0));
        HIP_TRY(hipMemsetAsync(hs[d], 0, (size_t)R * H * 4, st));
        HIP_TRY(hipMemsetAsync(cs[d], 0, (size_t)R * H * 4, st));
    }
    for (int d = 0; d < 2; ++d)
        for (int s = 0; s < L; ++s) {
            const int t = d == 0 ? s : L - 1 - s;
            GemmArgs g = gemm_args(R, 4 * H, H, hs[d], H, w[4 * d + 1], H, rec, 4 * H);
            TRY(launch_gemm(st, g, 0, 0));
            hipLaunchKernelGGL(lstm_gate_kernel, dim3(nblk((long)R * H)), dim3(256), 0, st, R, H, L, t, tokens, pad_id, pre[d], rec,
                               w[4 * d + 3], cs[d], hs[d]);
            LAUNCH_CHECK();
        }
    float** fin = use_cell ? cs : hs;
    hipLaunchKernelGGL(lstm_out_kernel, dim3(nblk((long)R * 2 * H)), dim3(256), 0, st, R, H, fin[0], fin[1], out);
    LAUNCH_CHECK();
    return FUMI_OK;
}

// tape of a training-mode forward, floats: x [R,L,E] | per direction: gates [R,L,4H], hprev [R,L,H], cprev [R,L,H], tanh_c [R,L,H]
int64_t fumi_hip_lstm_tape_floats(int R, int L, int E, int H) {
    if (R < 1 || L < 1 || E < 1 || H < 1) return 0;
    return (int64_t)R * L * E + 2 * (int64_t)R * L * 7 * H;
}

namespace {
struct LstmTape { float* x; float* gates[2]; float* hprev[2]; float* cprev[2]; float* tc[2]; };
LstmTape lstm_tape(float* tape, size_t RL, int E, int H) {
    LstmTape t;
    t.x = tape; tape += RL * E;
    for (int d = 0; d < 2; ++d) {
        t.gates[d] = tape; tape += RL * 4 * H;
        t.hprev[d] = tape; tape += RL * H;
        t.cprev[d] = tape; tape += RL * H;
        t.tc[d] = tape; tape += RL * H;
    }
    return t;
}
}  // namespace

int fumi_hip_lstm_bidir_train(fumi_ws_t* ws, fumi_stream_t stream, int R, int L, int E, int H,
        const int64_t* tokens, int64_t pad_id, const float* table, int64_t V, const float* const* w, int use_cell, float* out,
        float* tape) {
    if (!ws || !tokens || !table || !w || !out || !tape || R < 1 || L < 1 || E < 1 || H < 1 || V < 1) return FUMI_EINVAL;
    for (int i = 0; i < 8; ++i) if (!w[i]) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    const size_t RL = (size_t)R * L;
    size_t bytes = 2 * ws_align(RL * 4 * H * 4) + ws_align((size_t)R * 4 * H * 4) + 4 * ws_align((size_t)R * H * 4);
    int rc = ws_reserve(ws, bytes);
    if (rc) return rc;
    LstmTape tp = lstm_tape(tape, RL, E, H);
    float* pre[2] = {ws_f(ws, RL * 4 * H), ws_f(ws, RL * 4 * H)};
    float* rec = ws_f(ws, (size_t)R * 4 * H);
    float* hs[2] = {ws_f(ws, (size_t)R * H), ws_f(ws, (size_t)R * H)};
    float* cs[2] = {ws_f(ws, (size_t)R * H), ws_f(ws, (size_t)R * H)};
    TRY(fumi_hip_gather_rows(ws, stream, table, V, (int64_t)E * 4, tokens, (int64_t)RL, tp.x));
    for (int d = 0; d < 2; ++d) {
        GemmArgs g = gemm_args((int)RL, 4 * H, E, tp.x, E, w[4 * d], E, pre[d], 4 * H);
        g.bias = w[4 * d + 2];
        TRY(launch_gemm(st, g, 0, 0));
        HIP_TRY(hipMemsetAsync(hs[d], 0, (size_t)R * H * 4, st));
        HIP_TRY(hipMemsetAsync(cs[d], 0, (size_t)R * H * 4, st));
        HIP_TRY(hipMemsetAsync(tp.hprev[d], 0, RL * H * 4, st));          // steps past a row's length: a finite factor of their zero dgates
    }
    for (int d = 0; d < 2; ++d)
        for (int s = 0; s < L; ++s) {
            const int t = d == 0 ? s : L - 1 - s;
            GemmArgs g = gemm_args(R, 4 * H, H, hs[d], H, w[4 * d + 1], H, rec, 4 * H);
            TRY(launch_gemm(st, g, 0, 0));
            hipLaunchKernelGGL(lstm_gate_tape_kernel, dim3(nblk((long)R * H)), dim3(256), 0, st, R, H, L, t, tokens, pad_id, pre[d], rec,
                               w[4 * d + 3], cs[d], hs[d], tp.gates[d], tp.hprev[d], tp.cprev[d], tp.tc[d]);
            LAUNCH_CHECK();
        }
    float** fin = use_cell ? cs : hs;
    hipLaunchKernelGGL(lstm_out_kernel, dim3(nblk((long)R * 2 * H)), dim3(256), 0, st, R, H, fin[0], fin[1], out);
    LAUNCH_CHECK();
    return FUMI_OK;
}

int fumi_hip_lstm_bidir_bwd(fumi_ws_t* ws, fumi_stream_t stream, int R, int L, int E, int H,
        const int64_t* tokens, int64_t pad_id, const float* const* w, int use_cell, const float* tape, const float* d_out,
        float* const* g_w) {
    if (!ws || !tokens || !w || !tape || !d_out || !g_w || R < 1 || L < 1 || E < 1 || H < 1) return FUMI_EINVAL;
    for (int i = 0; i < 8; ++i) if (!w[i] || !g_w[i]) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    const size_t RL = (size_t)R * L;
    size_t bytes = ws_align(RL * 4 * H * 4) + 2 * ws_align((size_t)R * H * 4);
    int rc = ws_reserve(ws, bytes);
    if (rc) return rc;
    LstmTape tp = lstm_tape(const_cast<float*>(tape), RL, E, H);
    float* dg = ws_f(ws, RL * 4 * H);
    float* rec = ws_f(ws, (size_t)R * H);
    float* dc = ws_f(ws, (size_t)R * H);
    for (int d = 0; d < 2; ++d) {
        for (int s = 0; s < L; ++s) {
            const int t = d == 0 ? L - 1 - s : s;
            hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3(nblk((long)R * H)), dim3(256), 0, st, R, H, L, t, d, tokens, pad_id, tp.gates[d],
                               tp.cprev[d], tp.tc[d], d_out, use_cell, rec, dc, dg);
            LAUNCH_CHECK();
            if (s + 1 == L) break;
            GemmArgs g = gemm_args(R, H, 4 * H, dg + (size_t)t * 4 * H, L * 4 * H, w[4 * d + 1], H, rec, H);   // dh_prev = dgates_t W_hh
            TRY(launch_gemm(st, g, 0, 1));
        }
        GemmArgs g = gemm_args(4 * H, E, (int)RL, dg, 4 * H, tp.x, E, g_w[4 * d], E);                  // dW_ih = dgates^T x
        TRY(launch_gemm(st, g, 1, 1));
        g = gemm_args(4 * H, H, (int)RL, dg, 4 * H, tp.hprev[d], H, g_w[4 * d + 1], H);                // dW_hh = dgates^T h_prev
        TRY(launch_gemm(st, g, 1, 1));
        TRY(launch_colsum(st, dg, (int)RL, 4 * H, 4 * H, 1.f, g_w[4 * d + 2]));                        // both biases add to the same gate
        TRY(launch_colsum(st, dg, (int)RL, 4 * H, 4 * H, 1.f, g_w[4 * d + 3]));
    }
    return FUMI_OK;
}

}  // extern "C"
