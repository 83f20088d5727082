// C-ABI entry points (include/fumi_hip.h): workspace management and the orchestration of one meta-step.
#include "common.h"
#include "hyper_fwd.h"
#include "hyper_bwd.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_hip_err[512] = "";

void fumi_set_hip_error(hipError_t e, const char* where) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}

static hipEvent_t prof_event(fumi_ws* ws) {
    if (!ws->pool->empty()) { hipEvent_t e = ws->pool->back(); ws->pool->pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(fumi_ws* w, hipStream_t s, int phase) : ws(w), st(s), b(nullptr), on(w && ((w->profiling >> phase) & 1)) {
    if (on && ws->prof_every > 1) on = (ws->prof_seen[phase & 31]++ % (unsigned)ws->prof_every) == 0;
    if (!on) return;
    hipEvent_t a = prof_event(ws);
    b = prof_event(ws);
    (void)hipEventRecord(a, st);
    ws->recs->push_back(ProfRec{phase, a, b});
}
ProfScope::~ProfScope() { if (on) (void)hipEventRecord(b, st); }

int ws_reserve(fumi_ws* ws, size_t bytes) {
    ws->off = 0;
    if (bytes <= ws->cap) return FUMI_OK;
    // growth happens only on the first call of a new shape: synchronise, free, allocate 25 % headroom
    HIP_TRY(hipDeviceSynchronize());
    if (ws->base) HIP_TRY(hipFree(ws->base));
    ws->base = nullptr; ws->cap = 0;
    size_t head = bytes / 4;                                   // headroom, capped: ResNet-12 slabs are tens of GB
    if (head > ((size_t)1 << 30)) head = (size_t)1 << 30;
    size_t want = bytes + head + (1u << 20);
    hipError_t e = hipMalloc((void**)&ws->base, want);
    if (e != hipSuccess) { fumi_set_hip_error(e, "hipMalloc(workspace)"); ws->base = nullptr; return FUMI_ENOMEM; }
    ws->cap = want;
    return FUMI_OK;
}

hipStream_t ws_lane_stream(fumi_ws* ws, int i) {
    if (!ws || i < 1 || i > 3) return nullptr;
    hipStream_t& s = ws->lanes[i - 1];
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); s = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&ws->lane_ev[i - 1], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError(); (void)hipStreamDestroy(s); s = nullptr; ws->lane_ev[i - 1] = nullptr; return nullptr;
        }
        if (i == 1) ws->lane = s;
    }
    return s;
}

extern "C" {

int fumi_hip_version(void) { return 100; }

const char* fumi_hip_strerror(int code) {
    switch (code) {
        case FUMI_OK: return "ok";
        case FUMI_EINVAL: return "invalid argument";
        case FUMI_ENOMEM: return "workspace allocation failed";
        case FUMI_EHIP: return "HIP runtime error";
        case FUMI_ENOTSUP: return "not supported by this build";
        default: return "unknown error";
    }
}

const char* fumi_hip_last_hip_error(void) { return g_hip_err; }

int fumi_hip_workspace_create(int device, size_t bytes_hint, fumi_ws_t** out) {
    if (!out) return FUMI_EINVAL;
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(device));
    fumi_ws* ws = new fumi_ws();
    ws->device = device; ws->base = nullptr; ws->cap = 0; ws->off = 0; ws->status = nullptr; ws->status_host = nullptr; ws->hcnt = nullptr; ws->acnt = nullptr; ws->w0p = nullptr; ws->w0p_cap = 0; ws->side_buf = nullptr; ws->side_cap = 0; ws->pub_src = nullptr; ws->pub_dst = nullptr; ws->pub_n = 0; ws->pub_seq = 0; ws->adam = nullptr; ws->glove = nullptr; ws->text_grad = nullptr;
    ws->profiling = 0; ws->prof_every = 1; memset(ws->prof_seen, 0, sizeof(ws->prof_seen)); ws->recs = new std::vector<ProfRec>(); ws->pool = new std::vector<hipEvent_t>();
    ws->side = nullptr; ws->lane = nullptr;
    for (int i = 0; i < 3; ++i) { ws->lanes[i] = nullptr; ws->lane_ev[i] = nullptr; }
    for (auto& e : ws->ev) e = nullptr;
    for (auto& e : ws->evx) e = nullptr;
    // small persistent device buffers: status word, arrival counters (kept zero between launches by the kernels that use them)
    auto fail = [&](int code) { fumi_hip_workspace_destroy(ws); return code; };
    if (hipMalloc((void**)&ws->status, 256) != hipSuccess) return fail(FUMI_ENOMEM);
    if (hipHostMalloc((void**)&ws->status_host, 256, hipHostMallocDefault) != hipSuccess) return fail(FUMI_ENOMEM);
    if (hipMalloc((void**)&ws->hcnt, FUMI_HCNT * sizeof(int)) != hipSuccess) return fail(FUMI_ENOMEM);
    if (hipMalloc((void**)&ws->acnt, FUMI_ACNT * sizeof(int)) != hipSuccess) return fail(FUMI_ENOMEM);
    if (hipMemset(ws->status, 0, 256) != hipSuccess || hipMemset(ws->hcnt, 0, FUMI_HCNT * sizeof(int)) != hipSuccess ||
        hipMemset(ws->acnt, 0, FUMI_ACNT * sizeof(int)) != hipSuccess) return fail(FUMI_EHIP);
    {   // high priority: the side stream carries a few small workgroups that should get CU slots as soon as they are ready
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool hi = getenv("FUMI_SIDE_PRIO") && atoi(getenv("FUMI_SIDE_PRIO")) != 0;
        if (hipStreamCreateWithPriority(&ws->side, hipStreamNonBlocking, hi ? greatest : least) != hipSuccess) ws->side = nullptr;   // overlap is optional
    }
    for (auto& e : ws->ev) if (ws->side && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(ws->side); ws->side = nullptr;
    }
    for (auto& e : ws->evx) if (ws->side && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(ws->side); ws->side = nullptr;
    }
    if (bytes_hint) {
        int rc = ws_reserve(ws, bytes_hint);
        if (rc) { fumi_hip_workspace_destroy(ws); return rc; }
    }
    *out = ws;
    return FUMI_OK;
}

void fumi_hip_workspace_destroy(fumi_ws_t* ws) {
    if (!ws) return;
    (void)hipSetDevice(ws->device);
    (void)hipDeviceSynchronize();
    if (ws->base) (void)hipFree(ws->base);
    if (ws->status) (void)hipFree(ws->status);
    if (ws->hcnt) (void)hipFree(ws->hcnt);
    if (ws->acnt) (void)hipFree(ws->acnt);
    if (ws->side_buf) (void)hipFree(ws->side_buf);
    if (ws->w0p) (void)hipFree(ws->w0p);
    delete ws->adam;
    delete ws->glove;
    if (ws->status_host) (void)hipHostFree(ws->status_host);
    for (auto& r : *ws->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : *ws->pool) (void)hipEventDestroy(e);
    for (auto e : ws->ev) if (e) (void)hipEventDestroy(e);
    for (auto e : ws->evx) if (e) (void)hipEventDestroy(e);
    if (ws->side) (void)hipStreamDestroy(ws->side);
    for (int i = 0; i < 3; ++i) {
        if (ws->lanes[i]) (void)hipStreamDestroy(ws->lanes[i]);
        if (ws->lane_ev[i]) (void)hipEventDestroy(ws->lane_ev[i]);
    }
    delete ws->recs; delete ws->pool;
    delete ws;
}

size_t fumi_hip_workspace_bytes(const fumi_ws_t* ws) { return ws ? ws->cap + ws->w0p_cap : 0; }      // slab + operand planes

int fumi_hip_set_profiling(fumi_ws_t* ws, int on) {
    if (!ws) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    HIP_TRY(hipDeviceSynchronize());
    for (auto& r : *ws->recs) { ws->pool->push_back(r.a); ws->pool->push_back(r.b); }
    ws->recs->clear();
    ws->profiling = on;              // bit p = record HIP events around phase p (FUMI_PH_*); -1 = every phase
    memset(ws->prof_seen, 0, sizeof(ws->prof_seen));
    return FUMI_OK;
}

int fumi_hip_set_profiling_every(fumi_ws_t* ws, int every) {
    if (!ws || every < 1) return FUMI_EINVAL;
    ws->prof_every = every;
    memset(ws->prof_seen, 0, sizeof(ws->prof_seen));
    return FUMI_OK;
}

int fumi_hip_get_profile(fumi_ws_t* ws, int phase, double* total_ms, int* count) {
    if (!ws || !total_ms || !count || phase < 0 || phase >= FUMI_PH_COUNT) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    HIP_TRY(hipDeviceSynchronize());
    double tot = 0.0; int n = 0;
    for (auto& r : *ws->recs) {
        if (r.phase != phase) continue;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        tot += ms; ++n;
    }
    *total_ms = tot; *count = n;
    return FUMI_OK;
}

const char* fumi_hip_phase_name(int phase) {
    static const char* names[FUMI_PH_COUNT] = {"class_text_select", "hyper_fwd", "enc_gemm_s", "enc_gemm_q", "xpanel_fwd",
        "adapt", "query", "reverse", "reduce", "xpanel_bwd", "hyper_bwd", "am3_head", "conv_gemm", "conv_first", "conv_ew", "rn_conv", "rn_wgrad", "rn_ew"};
    return (phase >= 0 && phase < FUMI_PH_COUNT) ? names[phase] : "?";
}

int fumi_hip_read_status(fumi_ws_t* ws, fumi_stream_t stream, int* status_out) {
    if (!ws || !status_out) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(ws->status_host, ws->status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(ws->status, 0, sizeof(int), st));
    HIP_TRY(hipStreamSynchronize(st));
    *status_out = *ws->status_host;
    return FUMI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// FuMI
// ------------------------------------------------------------------------------------------------------------
static int fumi_step_impl(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int n_hidden, const int* hid, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale, float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi, const XRows* rows) {
    if (!ws || !hid || !theta || !phi || !y_s || !y_q || !logits_q || !preds_q || !loss_b || !acc_b) return FUMI_EINVAL;
    if (rows ? (!rows->table || !rows->idx_s || !rows->idx_q || rows->n_rows < 1) : (!x_s || !x_q)) return FUMI_EINVAL;
    if (!cls_text && !text_s) return FUMI_EINVAL;
    if (n_hidden < 1 || n_hidden > FUMI_MAX_HIDDEN || B < 1 || N < 1 || S < 1 || Qn < 1 || D < 1 || Dt < 1 || Ht < 1 || T < 0)
        return FUMI_EINVAL;
    if (need_grad && (!g_theta || !g_phi)) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));

    EpisodeProblem p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.N = N; p.S = S; p.Qn = Qn; p.D = D; p.L = n_hidden; p.T = T; p.alpha = alpha;
    p.need_grad = need_grad ? 1 : 0; p.second_order = 1; p.grad_scale = grad_scale;
    if (dropout_p < 0.f || dropout_p >= 1.f) return FUMI_EINVAL;
    p.dropout_p = dropout_p; p.seed = seed;
    for (int i = 0; i < n_hidden; ++i) {
        if (hid[i] < 1 || !theta[2 * i] || !theta[2 * i + 1]) return FUMI_EINVAL;
        p.h[i] = hid[i]; p.W[i] = theta[2 * i]; p.b[i] = theta[2 * i + 1];
        if (need_grad) { if (!g_theta[2 * i] || !g_theta[2 * i + 1]) return FUMI_EINVAL; p.gW[i] = g_theta[2 * i]; p.gb[i] = g_theta[2 * i + 1]; }
    }
    for (int i = 0; i < 4; ++i) if (!phi[i] || (need_grad && !g_phi[i])) return FUMI_EINVAL;
    const int H = hid[n_hidden - 1], R = B * N, H1 = H + 1;
    p.x_s = x_s; p.y_s = y_s; p.x_q = x_q; p.y_q = y_q;
    p.logits_q = logits_q; p.preds_q = preds_q; p.preds_f = preds_q_f32; p.loss_b = loss_b; p.acc_b = acc_b; p.stats = stats;
    if (rows) {
        p.rows = *rows;
        int rcx;
        if ((rcx = launch_index_range_check(st, rows->idx_s, (long)B * S, rows->n_rows, ws->status))) return rcx;
        if ((rcx = launch_index_range_check(st, rows->idx_q, (long)B * Qn, rows->n_rows, ws->status))) return rcx;
    }

    size_t bytes = episode_workspace_bytes(p);
    bytes += ws_align((size_t)R * Dt * 4) + 2 * ws_align((size_t)R * Ht * 4) + 3 * ws_align((size_t)R * H1 * 4);
    const bool hyper_lds = hyper_lds_fits(R, Dt, Ht, H1) != 0;         // LDS-resident hypernetwork kernels (hyper.hip)
    const size_t hpart_n = hyper_lds ? hyper_bwd_workspace_floats(R, Ht, H1) : 0;
    const size_t hfp_n = hyper_lds ? hyper_fwd_workspace_floats(R, Ht, H1) : 0;     // partial layer-1 products of the split forward
    const size_t hbf_n = need_grad ? hyper_bwd_fused_workspace_floats(R, Dt, Ht, H1) : 0;   // row-block slabs of the fused backward
    bytes += ws_align(hpart_n * 4) + ws_align(hfp_n * 4) + ws_align(hbf_n * 4);
    int rc = ws_reserve(ws, bytes);
    if (rc) return rc;
    float* c = ws_f(ws, (size_t)R * Dt);
    float* u = ws_f(ws, (size_t)R * Ht);
    float* ub = ws_f(ws, (size_t)R * Ht);
    float* h = ws_f(ws, (size_t)R * H1);
    float* hbar = ws_f(ws, (size_t)R * H1);
    float* hpb = ws_f(ws, (size_t)R * H1);
    float* hpart = hyper_lds ? ws_f(ws, hpart_n) : nullptr;
    float* hfp = hyper_lds ? ws_f(ws, hfp_n) : nullptr;
    float* hbf = hbf_n ? ws_f(ws, hbf_n) : nullptr;

    // A deferred embedding bag (fumi_hip_glove_bag_select_deferred) produces cls_text: it rides in the first launch of the forward
    // X-panel pass when that pass pre-splits its column operands AND the hypernetwork forward (the reader of cls_text) runs as that
    // pass's rider, i.e. behind it; in every other configuration it is launched here, before anything can read its output.
    GlovePending* glove = (ws->glove && ws->glove->on) ? ws->glove : nullptr;
    static const int glove_ride = getenv("FUMI_GLOVE_RIDE") ? atoi(getenv("FUMI_GLOVE_RIDE")) : 1;
    if (glove) {
        static const int overlap0 = getenv("FUMI_OVERLAP") ? atoi(getenv("FUMI_OVERLAP")) : 0;
        HyperFwdArgs probe_rider;
        const bool rides = glove_ride && cls_text && glove->a.out == cls_text && !(ws->side && (overlap0 & 1)) && hyper_lds &&
            hyper_fwd_split_args(R, Dt, Ht, H1, tanh_head, cls_text, phi[0], phi[1], phi[2], phi[3], u, h, hfp, ws->hcnt, &probe_rider) &&
            xpanel_fwd_presplits(B, S, Qn, D, hid[0], x_s, x_q, theta[0], true, rows, xpanel_planes(ws, B, S, D, hid[0]));
        if (!rides) { if ((rc = glove_flush(ws, st))) return rc; glove = nullptr; }
    }
    p.glove = glove;
    // class text rows (fumi.py:207-210), then the hypernetwork (fumi.py:70-86,104-113)
    const float* ctext = cls_text;
    if (!ctext) {
        ProfScope ps(ws, st, FUMI_PH_SELECT);
        if ((rc = launch_class_text_select(st, B, N, S, Dt, text_s, y_s, c, ws->status))) return rc;
        ctext = c;
    }
    // The text path (hypernetwork) and the image path (the two X-panel passes) only meet at the inner loop, so the
    // hypernetwork can run on the workspace's side stream beside xpanel_fwd (forward) / xpanel_bwd (backward).
    // FUMI_OVERLAP bit 0: hypernetwork forward beside xpanel_fwd, bit 1: hypernetwork backward beside xpanel_bwd.
    // Off by default.  Measured at the bench shapes (ms per step): 0 -> 0.3605, 1 -> 0.366, 2 -> 0.361, 3 -> 0.3533: a cross-
    // stream fork + join costs ~15 us of event latency on this platform, about what hiding the small kernels saves, and the
    // hypernetwork's workgroups stretch the matrix pass they run beside (80 -> 87 us).  Kept as a knob for larger text towers.
    static const int overlap = getenv("FUMI_OVERLAP") ? atoi(getenv("FUMI_OVERLAP")) : 0;
    const bool fork_bwd = ws->side && (overlap & 2) && need_grad;
    const bool fork = ws->side && (overlap & 1);       // forward fork
    hipStream_t sh = fork ? ws->side : st;             // stream of the hypernetwork forward
    GemmArgs g;
    auto hyper_forward = [&]() -> int {
        if (fork) HIP_TRY(hipStreamWaitEvent(sh, ws->ev[0], 0));     // class text rows are ready (recorded before xpanel_fwd)
        int r2;
        {
            ProfScope ps(ws, sh, FUMI_PH_HYPER_FWD);
            r2 = hyper_lds ? launch_hyper_fwd(sh, R, Dt, Ht, H1, tanh_head, ctext, phi[0], phi[1], phi[2], phi[3], u, h, hfp, ws->hcnt) : FUMI_ENOTSUP;
            if (r2 == FUMI_ENOTSUP) {                                // shapes outside the LDS-resident kernels: plain GEMMs
                g = gemm_args(R, Ht, Dt, ctext, Dt, phi[0], Dt, u, Ht);
                g.bias = phi[1]; g.act = 1;
                if ((r2 = launch_gemm(sh, g, 0, 0))) return r2;
                g = gemm_args(R, H1, Ht, u, Ht, phi[2], Ht, h, H1);
                g.bias = phi[3]; g.act = tanh_head ? 2 : 0;
                if ((r2 = launch_gemm(sh, g, 0, 0))) return r2;
            } else if (r2) return r2;
        }
        if (fork) HIP_TRY(hipEventRecord(ws->ev[1], sh));
        return FUMI_OK;
    };
    using HF = decltype(hyper_forward);
    HyperFwdArgs rider;
    if (fork) {
        // enqueued from inside run_episodes, right after xpanel_fwd: the matrix pass is not held up by this host work
        p.inputs_ready = ws->ev[0];
        p.after_xpanel_fwd = [](void* c) -> int { return (*(HF*)c)(); };
        p.hook_ctx = &hyper_forward;
        p.head_ready = ws->ev[1];
    } else {
        // default: the hypernetwork forward rides at the front of the forward X-panel launch (hyper_fwd.h) -- one dependent
        // launch less, no events; shapes the rider does not cover get their own launch right after that pass
        if (hyper_lds && hyper_fwd_split_args(R, Dt, Ht, H1, tanh_head, ctext, phi[0], phi[1], phi[2], phi[3], u, h, hfp, ws->hcnt, &rider)) {
            p.fwd_rider = &rider;
            p.fwd_rider_fallback = [](void* c) -> int { return (*(HF*)c)(); };
            p.hook_ctx = &hyper_forward;
        } else if ((rc = hyper_forward())) return rc;
    }
    if (fork_bwd) p.after_reverse = ws->ev[2];

    p.head = h; p.head_bar = hbar;
    ReduceSegs fin; fin.n = 0; fin.scale = grad_scale;          // every final sum of the step (episodes, gW0 slabs, hypernet
    if (need_grad) p.defer_reduce = &fin;                       // row-block slabs) goes into ONE launch at the very end
    // hypernetwork backward as one grid of independent workgroups (hyper_bwd.h) riding at the front of the backward X-panel
    // launch: its slabs join the step's final reduction
    HyperBwdArgs brider;
    struct BwdCtx { hipStream_t st; const HyperBwdArgs* a; } bctx{st, &brider};
    bool fused_bwd = false;
    // an armed text gradient (fumi_hip_want_text_grad) needs ubar [R,Ht] in memory: every form of the backward leaves it in ub
    float* text_grad = need_grad ? ws->text_grad : nullptr;
    if (need_grad) ws->text_grad = nullptr;
    if (need_grad && !fork_bwd && hbf &&
        hyper_bwd_fused_args(R, Dt, Ht, H1, tanh_head, 1.f, ctext, u, h, hbar, phi[2], hbf, g_phi[0], g_phi[1], g_phi[2], g_phi[3],
                             &fin, &brider, text_grad ? ub : nullptr)) {
        fused_bwd = true;
        p.bwd_rider = &brider;
        p.bwd_rider_fallback = [](void* c) -> int { BwdCtx* x = (BwdCtx*)c; return launch_hyper_bwd_fused(x->st, *x->a); };
        p.hook_ctx2 = &bctx;
    }
    if ((rc = run_episodes(ws, st, p))) return rc;
    if (!need_grad) return FUMI_OK;

    // hypernetwork backward: rows are (episode, class) pairs, weights are shared
    sh = fork_bwd ? ws->side : st;
    if (fork_bwd) HIP_TRY(hipStreamWaitEvent(sh, ws->ev[2], 0));   // head_bar is complete (recorded after the reverse sweep)
    rc = [&]() -> int {
    if (fused_bwd) return FUMI_OK;
    ProfScope ps(ws, sh, FUMI_PH_HYPER_BWD);
    if (hyper_lds) {
        int rc2 = launch_hyper_bwd(sh, R, Dt, Ht, H1, tanh_head, grad_scale, ctext, u, h, hbar, phi[2], ub, hpart,
                                   g_phi[0], g_phi[1], g_phi[2], g_phi[3], &fin);
        if (rc2 != FUMI_ENOTSUP) return rc2;
    }
    const float* hp = hbar;
    if (tanh_head) { if ((rc = launch_tanh_bwd(sh, (long)R * H1, h, hbar, hpb))) return rc; hp = hpb; }
    g = gemm_args(H1, Ht, R, hp, H1, u, Ht, g_phi[2], Ht);           // gA1 = hp^T u
    g.alpha = grad_scale;
    if ((rc = launch_gemm(sh, g, 1, 1))) return rc;
    if ((rc = launch_colsum(sh, hp, R, H1, H1, grad_scale, g_phi[3]))) return rc;
    g = gemm_args(R, Ht, H1, hp, H1, phi[2], Ht, ub, Ht);            // ubar = (hp A1) * relu'(u), mask in the epilogue
    g.mask = u;
    if ((rc = launch_gemm(sh, g, 0, 1))) return rc;
    g = gemm_args(Ht, Dt, R, ub, Ht, ctext, Dt, g_phi[0], Dt);       // gA0 = ubar^T c
    g.alpha = grad_scale;
    if ((rc = launch_gemm(sh, g, 1, 1))) return rc;
    if ((rc = launch_colsum(sh, ub, R, Ht, Ht, grad_scale, g_phi[1]))) return rc;
    return FUMI_OK;
    }();
    if (rc) return rc;
    if (text_grad) {                                                 // d(scale * sum_b loss_b) / d ctext = scale * ubar A0   [R,Dt]
        g = gemm_args(R, Dt, Ht, ub, Ht, phi[0], Dt, text_grad, Dt);
        g.alpha = grad_scale;
        if ((rc = launch_gemm(sh, g, 0, 1))) return rc;
    }
    if (fork_bwd) {                                                  // join: the caller's stream owns every result again
        HIP_TRY(hipEventRecord(ws->ev[3], sh));
        HIP_TRY(hipStreamWaitEvent(st, ws->ev[3], 0));
    }
    {
        ProfScope pr(ws, st, FUMI_PH_REDUCE);
        if ((rc = launch_reduce_multi_final(ws, st, fin))) return rc;      // (+ a deferred optimizer step and publication)
    }
    return FUMI_OK;
}

int fumi_hip_want_text_grad(fumi_ws_t* ws, float* g_cls_text) {
    if (!ws) return FUMI_EINVAL;
    ws->text_grad = g_cls_text;
    return FUMI_OK;
}

int fumi_hip_fumi_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int n_hidden, const int* hid, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale, float dropout_p, uint64_t seed,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi) {
    return fumi_step_impl(ws, stream, B, N, S, Qn, D, n_hidden, hid, Dt, Ht, T, alpha, tanh_head, need_grad, grad_scale, dropout_p,
                          seed, x_s, y_s, x_q, y_q, cls_text, text_s, theta, phi, logits_q, preds_q, preds_q_f32, loss_b, acc_b,
                          stats, g_theta, g_phi, nullptr);
}

int fumi_hip_fumi_step_indexed(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int n_hidden, const int* hid, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale, float dropout_p, uint64_t seed,
        const float* table, int64_t n_rows, const int64_t* idx_s, const int64_t* y_s, const int64_t* idx_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi) {
    XRows rows{table, idx_s, idx_q, (long)n_rows};
    return fumi_step_impl(ws, stream, B, N, S, Qn, D, n_hidden, hid, Dt, Ht, T, alpha, tanh_head, need_grad, grad_scale, dropout_p,
                          seed, nullptr, y_s, nullptr, y_q, cls_text, text_s, theta, phi, logits_q, preds_q, preds_q_f32, loss_b,
                          acc_b, stats, g_theta, g_phi, &rows);
}

// ------------------------------------------------------------------------------------------------------------
// MAML
// ------------------------------------------------------------------------------------------------------------
int fumi_hip_maml_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int D, int n_hidden, const int* hid,
        int T, float alpha, int first_order, int need_grad, float grad_scale,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* const* params,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_params) {
    if (!ws || !params || !x_s || !y_s || !x_q || !y_q || !logits_q || !preds_q || !loss_b || !acc_b) return FUMI_EINVAL;
    if (n_hidden < 0 || n_hidden > FUMI_MAX_HIDDEN || (n_hidden > 0 && !hid) || B < 1 || N < 1 || S < 1 || Qn < 1 || D < 1 || T < 0)
        return FUMI_EINVAL;
    if (need_grad && !g_params) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    if (n_hidden == 0) {
        // hidden_dims=None: the network is lin_final alone (maml.py:24-31).  Low-rank form with the head as "layer 0":
        // one xpanel_fwd (h0 = N), one small per-episode kernel, one xpanel_bwd (linhead.hip).
        const float* Wf = params[0]; const float* bf = params[1];
        if (!Wf || !bf || (need_grad && (!g_params[0] || !g_params[1]))) return FUMI_EINVAL;
        const size_t R = (size_t)S + Qn;
        int kc = 0;
        const int ns = need_grad ? xpanel_bwd_nsplit(B, S, Qn, D, N, &kc) : 0;
        size_t bytes = ws_align(B * R * N * 4) * 2 + ws_align(B * R * S * 4) + ws_align((size_t)B * N * 4) + ws_align((size_t)ns * N * D * 4);
        int rc = ws_reserve(ws, bytes);
        if (rc) return rc;
        float* A = ws_f(ws, B * R * N); float* G = ws_f(ws, B * R * S); float* Abar = ws_f(ws, B * R * N);
        float* bbar = ws_f(ws, (size_t)B * N);
        if ((rc = launch_xpanel_fwd(st, B, S, Qn, D, N, x_s, x_q, Wf, A, G))) return rc;
        if ((rc = launch_linhead(st, B, N, S, Qn, T, alpha, need_grad ? 1 : 0, first_order ? 0 : 1, A, G, bf, y_s, y_q, logits_q,
                                 preds_q, preds_q_f32, loss_b, acc_b, Abar, bbar, ws->status))) return rc;
        ReduceSegs sg; sg.n = 0; sg.scale = grad_scale;
        if (stats) { sg.add(loss_b, B, 1, 1, stats); sg.add(acc_b, B, 1, 1, stats + 1); }
        if (need_grad) {
            float* slabs = ws_f(ws, (size_t)ns * N * D);
            if ((rc = launch_xpanel_bwd(st, B, S, Qn, D, N, x_s, x_q, Abar, slabs, kc, ns))) return rc;
            sg.add(slabs, ns, (long)N * D, (long)N * D, g_params[0]);
            sg.add(bbar, B, N, N, g_params[1]);
        }
        return launch_reduce_multi(st, sg);
    }

    EpisodeProblem p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.N = N; p.S = S; p.Qn = Qn; p.D = D; p.L = n_hidden; p.T = T; p.alpha = alpha;
    p.need_grad = need_grad ? 1 : 0; p.second_order = first_order ? 0 : 1; p.grad_scale = grad_scale;
    for (int i = 0; i < n_hidden; ++i) {
        if (hid[i] < 1 || !params[2 * i] || !params[2 * i + 1]) return FUMI_EINVAL;
        p.h[i] = hid[i]; p.W[i] = params[2 * i]; p.b[i] = params[2 * i + 1];
        if (need_grad) { if (!g_params[2 * i] || !g_params[2 * i + 1]) return FUMI_EINVAL; p.gW[i] = g_params[2 * i]; p.gb[i] = g_params[2 * i + 1]; }
    }
    const float* Wf = params[2 * n_hidden]; const float* bf = params[2 * n_hidden + 1];
    if (!Wf || !bf || (need_grad && (!g_params[2 * n_hidden] || !g_params[2 * n_hidden + 1]))) return FUMI_EINVAL;
    const int H = hid[n_hidden - 1], H1 = H + 1;
    p.x_s = x_s; p.y_s = y_s; p.x_q = x_q; p.y_q = y_q;
    p.logits_q = logits_q; p.preds_q = preds_q; p.preds_f = preds_q_f32; p.loss_b = loss_b; p.acc_b = acc_b; p.stats = stats;

    size_t bytes = episode_workspace_bytes(p) + 2 * ws_align((size_t)B * N * H1 * 4);
    int rc = ws_reserve(ws, bytes);
    if (rc) return rc;
    float* head = ws_f(ws, (size_t)B * N * H1);
    float* hbar = ws_f(ws, (size_t)B * N * H1);
    if ((rc = launch_broadcast_head(st, B, N, H, Wf, bf, head))) return rc;
    p.head = head; p.head_bar = hbar;
    if ((rc = run_episodes(ws, st, p))) return rc;
    if (!need_grad) return FUMI_OK;
    return launch_split_head_grad(st, B, N, H, hbar, grad_scale, g_params[2 * n_hidden], g_params[2 * n_hidden + 1]);
}

// ------------------------------------------------------------------------------------------------------------
// finer-grained ops
// ------------------------------------------------------------------------------------------------------------
int fumi_hip_class_text_select(fumi_ws_t* ws, fumi_stream_t stream, int B, int N, int S, int Dt,
        const float* text_s, const int64_t* y_s, float* out) {
    if (!ws || !text_s || !y_s || !out || B < 1 || N < 1 || S < 1 || Dt < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    return launch_class_text_select((hipStream_t)stream, B, N, S, Dt, text_s, y_s, out, ws->status);
}

int fumi_hip_xpanel_fwd(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int D, int h0,
        const float* x_s, const float* x_q, const float* W0, float* A0, float* G) {
    if (!ws || !x_s || !x_q || !W0 || !A0 || !G || B < 1 || S < 1 || Qn < 1 || D < 1 || h0 < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    ProfScope ps(ws, (hipStream_t)stream, FUMI_PH_XPANEL_FWD);
    return launch_xpanel_fwd((hipStream_t)stream, B, S, Qn, D, h0, x_s, x_q, W0, A0, G, nullptr, nullptr, nullptr, nullptr, nullptr,
                             xpanel_planes(ws, B, S, D, h0));
}

int fumi_hip_xpanel_bwd(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int D, int h0,
        const float* x_s, const float* x_q, const float* Abar, float scale, float* gW0) {
    if (!ws || !x_s || !x_q || !Abar || !gW0 || B < 1 || S < 1 || Qn < 1 || D < 1 || h0 < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    hipStream_t st = (hipStream_t)stream;
    int kc;
    const int ns = xpanel_bwd_nsplit(B, S, Qn, D, h0, &kc);
    const long slab = (long)h0 * D;
    int rc = ws_reserve(ws, ws_align((size_t)ns * slab * 4));
    if (rc) return rc;
    float* slabs = ws_f(ws, (size_t)ns * slab);
    ProfScope ps(ws, st, FUMI_PH_XPANEL_BWD);
    if ((rc = launch_xpanel_bwd(st, B, S, Qn, D, h0, x_s, x_q, Abar, slabs, kc, ns))) return rc;
    return launch_reduce_slabs(st, slabs, ns, slab, slab, scale, gW0);
}

int fumi_hip_linear_fwd(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, int K,
        const float* x, const float* W, const float* b, int act, float* y) {
    if (!ws || !x || !W || !y || M < 1 || N < 1 || K < 1 || act < 0 || act > 2) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    GemmArgs g = gemm_args(M, N, K, x, K, W, K, y, N);
    g.bias = b; g.act = act;
    return launch_gemm((hipStream_t)stream, g, 0, 0);
}

int fumi_hip_linear_bwd_data(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, int K,
        const float* dy, const float* W, float* dx) {
    if (!ws || !dy || !W || !dx || M < 1 || N < 1 || K < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    GemmArgs g = gemm_args(M, K, N, dy, N, W, K, dx, K);             // dx[M,K] = dy[M,N] W[N,K]
    return launch_gemm((hipStream_t)stream, g, 0, 1);
}

int fumi_hip_linear_bwd_weight(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, int K,
        const float* dy, const float* x, float* dW, float* db) {
    if (!ws || !dy || !x || !dW || M < 1 || N < 1 || K < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    GemmArgs g = gemm_args(N, K, M, dy, N, x, K, dW, K);             // dW[N,K] = dy^T x
    int rc = launch_gemm((hipStream_t)stream, g, 1, 1);
    if (rc || !db) return rc;
    return launch_colsum((hipStream_t)stream, dy, M, N, N, 1.f, db);
}

}  // extern "C"
