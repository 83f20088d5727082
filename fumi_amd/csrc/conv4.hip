// Meta-step with the Conv4 image encoder at the im_net seam (fumi/models/fumi.py:89-100 is the seam; the inner loop, the
// query loss and the second-order outer gradient are fumi.py:146-192 / maml.py:156-191).  Every launch covers all B episodes
// of the meta-batch with per-episode fast weights.  Algorithm = oracle/conv4_manual.py (forward-over-reverse second order):
//
//   for t < T:   forward(theta_t) -> tape_t;  backward -> g_t;  theta_{t+1} = theta_t - alpha g_t          (support set)
//   query:       forward(theta_T) -> logits, loss;  backward -> bar_T = d loss / d theta_T
//   for t = T-1 .. 0:  bar_t = bar_{t+1} - alpha H_t bar_{t+1}   (one tangent forward + one tangent backward over tape_t)
//   g_theta = grad_scale * sum_b bar_0[b];   head_bar = bar_0 of the head (handed to the hypernetwork / lin_final).
//
// HBM is the tape: 288 GB hold every pre-activation, pooled activation and their gradients of every inner step (6.5 MB per
// 84 x 84 image and step), nothing is recomputed except the arg-max of the pooling windows.
#include "conv4.h"
#include <string.h>
#include <functional>

namespace {

static int g_cv_lanes = 0;      // fumi_hip_conv4_set_option(1, n): lanes of the meta-steps and encoder calls (0: the defaults)
static int g_c1_fused = 1;     // block 1 without its full-resolution maps (conv_first.hip); 0 = the plain passes (probe tests)

struct Net {
    int B, nblk, Cin, N, F;
    int fused1;                                      // block 1 recomputed band by band: u[0] / du[0] are never materialised
    CvGeom g[CV_MAXBLK];
    int Ho[CV_MAXBLK], Wo[CV_MAXBLK];
    long PSZ, FSZ;                                   // floats per episode: parameter slab, fragment slab
    long offW[CV_MAXBLK], offG[CV_MAXBLK], offB[CV_MAXBLK];     // within a parameter slab
    long fF[CV_MAXBLK], fB[CV_MAXBLK];               // within a fragment slab (forward / backward-data order)
};

static int net_init(Net& n, int B, int nblk, int Cin, int N, int H, int W) {
    if (nblk < 1 || nblk > CV_MAXBLK || Cin < 1 || Cin > 3 || B < 1 || N < 1) return FUMI_EINVAL;
    n.B = B; n.nblk = nblk; n.Cin = Cin; n.N = N;
    long po = 0, fo = 0;
    for (int l = 0; l < nblk; ++l) {
        if (H < 2 || W < 2) return FUMI_EINVAL;
        n.g[l] = cv_geom(H, W);
        if ((size_t)(CV_TILE + 2 * n.g[l].halo) * 256 > 160 * 1024) return FUMI_ENOTSUP;
        n.Ho[l] = H / 2; n.Wo[l] = W / 2;
        n.offW[l] = po; po += l == 0 ? 2048 : 36864;
        n.offG[l] = po; po += 64;
        n.offB[l] = po; po += 64;
        if (l == 0) { n.fF[0] = fo; n.fB[0] = -1; fo += (cv_frag1_floats(Cin) + 63) / 64 * 64; }
        else { n.fF[l] = fo; fo += CV_WFRAG; n.fB[l] = fo; fo += CV_WFRAG; }
        H /= 2; W /= 2;
    }
    n.PSZ = po; n.FSZ = fo;
    n.F = 64 * n.Ho[nblk - 1] * n.Wo[nblk - 1];
    n.fused1 = g_c1_fused && nblk >= 2 && n.g[0].H >= 2 && n.g[0].W >= 2;
    return FUMI_OK;
}

// buffers of one pass over M images per episode (a support step's tape, or the query pass)
struct PassBufs {
    int M;
    float* u[CV_MAXBLK]; float* x[CV_MAXBLK]; float* du[CV_MAXBLK]; float* dx[CV_MAXBLK]; float* coef[CV_MAXBLK];
    float* z; float* p; float* dz;
};
// tangent scratch (support-sized)
struct TanBufs { float* ud[CV_MAXBLK]; float* xd[CV_MAXBLK]; float* dud[CV_MAXBLK]; float* dxd[CV_MAXBLK]; float* dzd; };

static size_t act_floats(const Net& n, int M, int l) { return (size_t)n.B * M * n.g[l].Pp * 64; }
static size_t out_floats(const Net& n, int M, int l) {          // pooled output of block l
    return l + 1 < n.nblk ? (size_t)n.B * M * n.g[l + 1].Pp * 64 : (size_t)n.B * M * n.F;
}
static size_t pass_bytes(const Net& n, int M, bool bwd) {
    size_t b = 0;
    for (int l = 0; l < n.nblk; ++l) {
        if (!(l == 0 && n.fused1)) b += ws_align(act_floats(n, M, l) * 4) * (bwd ? 2 : 1);
        b += ws_align(out_floats(n, M, l) * 4) * (bwd ? 2 : 1);
        b += ws_align((size_t)n.B * CF_N * 64 * 4);
    }
    return b + 3 * ws_align((size_t)n.B * M * n.N * 4);
}
static void pass_carve(fumi_ws* ws, const Net& n, int M, bool bwd, PassBufs& pb) {
    pb.M = M;
    for (int l = 0; l < n.nblk; ++l) {
        const bool maps = !(l == 0 && n.fused1);
        pb.u[l] = maps ? ws_f(ws, act_floats(n, M, l)) : nullptr;
        pb.x[l] = ws_f(ws, out_floats(n, M, l));
        pb.du[l] = bwd && maps ? ws_f(ws, act_floats(n, M, l)) : nullptr;
        pb.dx[l] = bwd ? ws_f(ws, out_floats(n, M, l)) : nullptr;
        pb.coef[l] = ws_f(ws, (size_t)n.B * CF_N * 64);
    }
    pb.z = ws_f(ws, (size_t)n.B * M * n.N); pb.p = ws_f(ws, (size_t)n.B * M * n.N); pb.dz = ws_f(ws, (size_t)n.B * M * n.N);
}
static size_t tan_bytes(const Net& n, int M) {
    size_t b = 0;
    for (int l = 0; l < n.nblk; ++l)
        b += (l == 0 && n.fused1 ? 0 : 2 * ws_align(act_floats(n, M, l) * 4)) + 2 * ws_align(out_floats(n, M, l) * 4);
    return b + ws_align((size_t)n.B * M * n.N * 4);
}
static void tan_carve(fumi_ws* ws, const Net& n, int M, TanBufs& tb) {
    for (int l = 0; l < n.nblk; ++l) {
        const bool maps = !(l == 0 && n.fused1);
        tb.ud[l] = maps ? ws_f(ws, act_floats(n, M, l)) : nullptr; tb.xd[l] = ws_f(ws, out_floats(n, M, l));
        tb.dud[l] = maps ? ws_f(ws, act_floats(n, M, l)) : nullptr; tb.dxd[l] = ws_f(ws, out_floats(n, M, l));
    }
    tb.dzd = ws_f(ws, (size_t)n.B * M * n.N);
}

struct Scratch { float* cpart; float* wpart; float* rpart; double* dsum; float* rowl; size_t cpart_n, wpart_n, rpart_n, dsum_n, rowl_n; };

static EwGeom ew_geom(const Net& n, int M, int l) {
    EwGeom e; e.B = n.B; e.M = M; e.g = n.g[l]; e.Ho = n.Ho[l]; e.Wo = n.Wo[l];
    e.last = l + 1 == n.nblk;
    e.gn = e.last ? n.g[l] : n.g[l + 1];
    return e;
}

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
// a launch inside a profiling phase (HIP events on the stream only when bench.py switched the phase on)
#define TRYP(phase, expr) do { ProfScope _ps(c.ws, c.st, phase); int _rc = (expr); if (_rc) return _rc; } while (0)

// the fragment-order copies of one parameter slot (block 1's canonical weights are gathered into a dense temporary first)
static int frags_of_slot(hipStream_t st, const Net& n, const float* params, float* frags, float* tmp1 /*[B][2048+frag1]*/) {
    // block 1: gather the strided canonical weights into a dense temp, build, scatter back (tiny)
    const int nf1 = cv_frag1_floats(n.Cin);
    float* dense = tmp1; float* fdense = tmp1 + (size_t)n.B * 2048;
    HIP_TRY(hipMemcpy2DAsync(dense, 2048 * 4, params + n.offW[0], n.PSZ * 4, 2048 * 4, n.B, hipMemcpyDeviceToDevice, st));
    TRY(launch_wfrag1(st, n.B, n.Cin, dense, fdense));
    HIP_TRY(hipMemcpy2DAsync(frags + n.fF[0], n.FSZ * 4, fdense, (size_t)nf1 * 4, (size_t)nf1 * 4, n.B, hipMemcpyDeviceToDevice, st));
    for (int l = 1; l < n.nblk; ++l)
        TRY(launch_wfrag64(st, n.B, params + n.offW[l], n.PSZ, frags + n.fF[l], frags + n.fB[l], n.FSZ));
    return FUMI_OK;
}

// meta-parameters (torch layouts) -> one canonical copy per episode
static int slot0_from_theta(hipStream_t st, const Net& n, const float* const* theta, float* P0, float* toi_tmp) {
    TRY(launch_w1_to_canon(st, n.B, n.Cin, theta[0], 0, P0 + n.offW[0], n.PSZ));
    for (int l = 0; l < n.nblk; ++l) {
        if (l) {
            TRY(launch_oihw_to_toi(st, 1, theta[3 * l], toi_tmp + (size_t)l * 36864));
            TRY(launch_broadcast(st, n.B, 36864, toi_tmp + (size_t)l * 36864, P0 + n.offW[l], n.PSZ));
        }
        TRY(launch_broadcast(st, n.B, 64, theta[3 * l + 1], P0 + n.offG[l], n.PSZ));
        TRY(launch_broadcast(st, n.B, 64, theta[3 * l + 2], P0 + n.offB[l], n.PSZ));
    }
    return FUMI_OK;
}

struct StepCtx {
    fumi_ws* ws; hipStream_t st; Net n; Scratch sc;
    const float* img_s; const float* img_q; const int64_t* y_s; const int64_t* y_q;
    int S, Qn;
};

static C1Args c1_args(const Net& n, int M, const float* img, const float* frags, const float* coef) {
    C1Args a; memset(&a, 0, sizeof(a));
    a.B = n.B; a.M = M; a.Cin = n.Cin; a.g = n.g[0]; a.gn = n.g[1]; a.Ho = n.Ho[0]; a.Wo = n.Wo[0];
    a.img = img; a.frag = frags + n.fF[0]; a.frag_stride = n.FSZ; a.coef = coef;
    return a;
}

static int forward_pass(StepCtx& c, int M, const float* img, const float* params, const float* frags, PassBufs& pb,
                        const float* head, const int64_t* y, float scale, float* logits, int64_t* preds, float* preds_f,
                        float* loss_b, float* acc_b) {
    const Net& n = c.n;
    for (int l = 0; l < n.nblk; ++l) {
        const long npix = (long)M * n.g[l].Pp;
        const int tiles = cv_tiles(npix);
        int nt_stats = tiles;
        if (l == 0 && n.fused1) {
            int chunk;
            nt_stats = c1_chunks(n.B, M, n.g[0], &chunk);
            C1Args fa = c1_args(n, M, img, frags, nullptr); fa.part = c.sc.cpart;
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 3, 0));
        } else if (l == 0) {
            Conv1Args a; memset(&a, 0, sizeof(a));
            a.B = n.B; a.M = M; a.Cin = n.Cin; a.g = n.g[0]; a.img = img; a.frag = frags + n.fF[0]; a.frag_stride = n.FSZ;
            a.out = pb.u[0]; a.stats = c.sc.cpart; a.dot = nullptr;
            TRYP(FUMI_PH_CONV_FIRST, launch_conv1(c.st, a));
        } else {
            Conv64Args a; a.B = n.B; a.nsrc = 1; a.npix = npix; a.g = n.g[l];
            a.in[0] = pb.x[l - 1]; a.frag[0] = frags + n.fF[l]; a.frag_stride[0] = n.FSZ; a.in[1] = nullptr; a.frag[1] = nullptr; a.frag_stride[1] = 0;
            a.out = pb.u[l]; a.stats = c.sc.cpart; a.dot = nullptr;
            TRYP(FUMI_PH_CONV_GEMM, launch_conv64(c.st, a));
        }
        CoefArgs ca; memset(&ca, 0, sizeof(ca));
        ca.B = n.B; ca.mode = CFM_FWD; ca.nt = nt_stats; ca.K = 2; ca.n = (float)((double)M * n.g[l].H * n.g[l].W);
        ca.part = c.sc.cpart; ca.coef = pb.coef[l]; ca.g = params + n.offG[l]; ca.beta = params + n.offB[l]; ca.pstride = n.PSZ;
        TRYP(FUMI_PH_CONV_EW, launch_coef(c.st, ca, c.sc.dsum));
        if (l == 0 && n.fused1) {
            C1Args fa = c1_args(n, M, img, frags, pb.coef[0]); fa.x = pb.x[0];
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 0, 0));
            continue;
        }
        PoolFwdArgs pa; pa.e = ew_geom(n, M, l); pa.u = pb.u[l]; pa.ud = nullptr; pa.coef = pb.coef[l]; pa.x = pb.x[l]; pa.xd = nullptr;
        TRYP(FUMI_PH_CONV_EW, launch_pool_fwd(c.st, pa, 0));
    }
    if (!head) return FUMI_OK;                                        // features only (fumi_hip_conv4_features)
    HeadArgs h; memset(&h, 0, sizeof(h));
    h.B = n.B; h.M = M; h.N = n.N; h.F = n.F; h.scale = scale; h.f = pb.x[n.nblk - 1]; h.head = head; h.y = y;
    h.z = logits ? logits : pb.z; h.p = pb.p; h.dz = pb.dz; h.preds = preds; h.preds_f = preds_f; h.loss_b = loss_b; h.acc_b = acc_b;
    h.status = c.ws->status; h.row_loss = c.sc.rowl; h.row_hit = c.sc.rowl + (size_t)n.B * M;
    TRYP(FUMI_PH_CONV_EW, launch_head_logits(c.st, h));
    return FUMI_OK;
}

// gradient of the pass's loss w.r.t. (parameter slab, head): G [B][PSZ], dh [B][N][F+1]
static int backward_pass(StepCtx& c, int M, const float* img, const float* frags, PassBufs& pb, const float* head,
                         float* G, float* dh) {
    const Net& n = c.n;
    if (head) {                             // (head == NULL: the caller put the feature adjoints into pb.dx[last] itself)
        HeadGradArgs hg; memset(&hg, 0, sizeof(hg));
        hg.B = n.B; hg.M = M; hg.N = n.N; hg.F = n.F; hg.nsrc = 1; hg.dz[0] = pb.dz; hg.f[0] = pb.x[n.nblk - 1]; hg.head[0] = head;
        hg.dh = dh; hg.df = pb.dx[n.nblk - 1];
        TRYP(FUMI_PH_CONV_EW, launch_head_grad(c.st, hg));
    }
    for (int l = n.nblk - 1; l >= 0; --l) {
        const EwGeom e = ew_geom(n, M, l);
        const long npix = (long)M * n.g[l].Pp;
        if (l == 0 && n.fused1) {
            int chunk;
            const int nt = c1_chunks(n.B, M, n.g[0], &chunk);
            C1Args fa = c1_args(n, M, img, frags, pb.coef[0]); fa.dxo = pb.dx[0]; fa.part = c.sc.rpart; fa.wpart = c.sc.wpart;
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 1, 0));
            CoefArgs ca; memset(&ca, 0, sizeof(ca));
            ca.B = n.B; ca.mode = CFM_BWD; ca.nt = nt; ca.K = 2; ca.n = (float)((double)M * n.g[0].H * n.g[0].W);
            ca.part = c.sc.rpart; ca.coef = pb.coef[0]; ca.dg = G + n.offG[0]; ca.dbeta = G + n.offB[0]; ca.gstride = n.PSZ;
            TRYP(FUMI_PH_CONV_EW, launch_coef(c.st, ca, c.sc.dsum));
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 2, 0));
            TRY(launch_reduce_batched(c.st, n.B, nt, 2048, c.sc.wpart, 1.f, G + n.offW[0], n.PSZ));
            continue;
        }
        BwdRedArgs ra; ra.e = e; ra.u = pb.u[l]; ra.ud = nullptr; ra.dxo = pb.dx[l]; ra.dxod = nullptr; ra.coef = pb.coef[l];
        ra.part = c.sc.rpart; ra.nt = ew_bwd_red_nt(e);
        TRYP(FUMI_PH_CONV_EW, launch_bwd_reduce(c.st, ra, 0));
        CoefArgs ca; memset(&ca, 0, sizeof(ca));
        ca.B = n.B; ca.mode = CFM_BWD; ca.nt = ra.nt; ca.K = 2; ca.n = (float)((double)M * n.g[l].H * n.g[l].W);
        ca.part = c.sc.rpart; ca.coef = pb.coef[l]; ca.dg = G + n.offG[l]; ca.dbeta = G + n.offB[l]; ca.gstride = n.PSZ;
        TRYP(FUMI_PH_CONV_EW, launch_coef(c.st, ca, c.sc.dsum));
        BwdApplyArgs aa; aa.e = e; aa.u = pb.u[l]; aa.ud = nullptr; aa.dxo = pb.dx[l]; aa.dxod = nullptr; aa.coef = pb.coef[l]; aa.du = pb.du[l];
        TRYP(FUMI_PH_CONV_EW, launch_bwd_apply(c.st, aa, 0));
        const int ns = cv_wgrad_nsplit(n.B, npix);
        if (l == 0) {
            Wgrad1Args wa; wa.B = n.B; wa.M = M; wa.Cin = n.Cin; wa.nsplit = ns; wa.g = n.g[0]; wa.img = img; wa.dy = pb.du[0]; wa.part = c.sc.wpart;
            TRYP(FUMI_PH_CONV_FIRST, launch_wgrad1(c.st, wa));
            TRY(launch_reduce_batched(c.st, n.B, ns, 2048, c.sc.wpart, 1.f, G + n.offW[0], n.PSZ));
        } else {
            Wgrad64Args wa; wa.B = n.B; wa.nsrc = 1; wa.nsplit = ns; wa.npix = npix; wa.g = n.g[l];
            wa.x[0] = pb.x[l - 1]; wa.dy[0] = pb.du[l]; wa.x[1] = nullptr; wa.dy[1] = nullptr; wa.part = c.sc.wpart;
            TRYP(FUMI_PH_CONV_GEMM, launch_wgrad64(c.st, wa));
            TRY(launch_reduce_batched(c.st, n.B, ns, 36864, c.sc.wpart, 1.f, G + n.offW[l], n.PSZ));
            Conv64Args a; a.B = n.B; a.nsrc = 1; a.npix = npix; a.g = n.g[l];
            a.in[0] = pb.du[l]; a.frag[0] = frags + n.fB[l]; a.frag_stride[0] = n.FSZ; a.in[1] = nullptr; a.frag[1] = nullptr; a.frag_stride[1] = 0;
            a.out = pb.dx[l - 1]; a.stats = nullptr; a.dot = nullptr;
            TRYP(FUMI_PH_CONV_GEMM, launch_conv64(c.st, a));
        }
    }
    return FUMI_OK;
}

// HV [B][PSZ], HVh [B][N][F+1] = Hessian of the support loss at the tape's parameters times (V, Vh)
static int hvp_pass(StepCtx& c, int M, const float* img, const float* frags, PassBufs& pb, TanBufs& tb, const float* head,
                    const float* V, const float* Vfrags, const float* Vh, float scale, float* HV, float* HVh) {
    const Net& n = c.n;
    // ---- tangent forward
    for (int l = 0; l < n.nblk; ++l) {
        const long npix = (long)M * n.g[l].Pp;
        const int tiles = cv_tiles(npix);
        int nt_stats = tiles;
        if (l == 0 && n.fused1) {
            int chunk;
            nt_stats = c1_chunks(n.B, M, n.g[0], &chunk);
            C1Args fa = c1_args(n, M, img, frags, nullptr); fa.fragd = Vfrags + n.fF[0]; fa.fragd_stride = n.FSZ; fa.part = c.sc.cpart;
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 3, 1));
        } else if (l == 0) {
            Conv1Args a; memset(&a, 0, sizeof(a));
            a.B = n.B; a.M = M; a.Cin = n.Cin; a.g = n.g[0]; a.img = img; a.frag = Vfrags + n.fF[0]; a.frag_stride = n.FSZ;
            a.out = tb.ud[0]; a.stats = c.sc.cpart; a.dot = pb.u[0];
            TRYP(FUMI_PH_CONV_FIRST, launch_conv1(c.st, a));
        } else {
            Conv64Args a; a.B = n.B; a.nsrc = 2; a.npix = npix; a.g = n.g[l];
            a.in[0] = pb.x[l - 1]; a.frag[0] = Vfrags + n.fF[l]; a.frag_stride[0] = n.FSZ;
            a.in[1] = tb.xd[l - 1]; a.frag[1] = frags + n.fF[l]; a.frag_stride[1] = n.FSZ;
            a.out = tb.ud[l]; a.stats = c.sc.cpart; a.dot = pb.u[l];
            TRYP(FUMI_PH_CONV_GEMM, launch_conv64(c.st, a));
        }
        CoefArgs ca; memset(&ca, 0, sizeof(ca));
        ca.B = n.B; ca.mode = CFM_TFWD; ca.nt = nt_stats; ca.K = 2; ca.n = (float)((double)M * n.g[l].H * n.g[l].W);
        ca.part = c.sc.cpart; ca.coef = pb.coef[l]; ca.gd = V + n.offG[l]; ca.betad = V + n.offB[l]; ca.dstride = n.PSZ;
        TRYP(FUMI_PH_CONV_EW, launch_coef(c.st, ca, c.sc.dsum));
        if (l == 0 && n.fused1) {
            C1Args fa = c1_args(n, M, img, frags, pb.coef[0]); fa.fragd = Vfrags + n.fF[0]; fa.fragd_stride = n.FSZ; fa.xd = tb.xd[0];
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 0, 1));
            continue;
        }
        PoolFwdArgs pa; pa.e = ew_geom(n, M, l); pa.u = pb.u[l]; pa.ud = tb.ud[l]; pa.coef = pb.coef[l]; pa.x = nullptr; pa.xd = tb.xd[l];
        TRYP(FUMI_PH_CONV_EW, launch_pool_fwd(c.st, pa, 1));
    }
    HeadArgs h; memset(&h, 0, sizeof(h));
    h.B = n.B; h.M = M; h.N = n.N; h.F = n.F; h.scale = scale; h.f = pb.x[n.nblk - 1]; h.head = head; h.y = nullptr;
    h.fd = tb.xd[n.nblk - 1]; h.headd = Vh; h.p = pb.p; h.dz = tb.dzd;
    TRYP(FUMI_PH_CONV_EW, launch_head_logits(c.st, h));
    // ---- tangent backward
    HeadGradArgs hg; memset(&hg, 0, sizeof(hg));
    hg.B = n.B; hg.M = M; hg.N = n.N; hg.F = n.F; hg.nsrc = 2;
    hg.dz[0] = tb.dzd; hg.f[0] = pb.x[n.nblk - 1]; hg.head[0] = head;
    hg.dz[1] = pb.dz; hg.f[1] = tb.xd[n.nblk - 1]; hg.head[1] = Vh;
    hg.dh = HVh; hg.df = tb.dxd[n.nblk - 1];
    TRYP(FUMI_PH_CONV_EW, launch_head_grad(c.st, hg));
    for (int l = n.nblk - 1; l >= 0; --l) {
        const EwGeom e = ew_geom(n, M, l);
        const long npix = (long)M * n.g[l].Pp;
        if (l == 0 && n.fused1) {
            int chunk;
            const int nt = c1_chunks(n.B, M, n.g[0], &chunk);
            C1Args fa = c1_args(n, M, img, frags, pb.coef[0]); fa.fragd = Vfrags + n.fF[0]; fa.fragd_stride = n.FSZ;
            fa.dxo = pb.dx[0]; fa.dxod = tb.dxd[0]; fa.part = c.sc.rpart; fa.wpart = c.sc.wpart;
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 1, 1));
            CoefArgs ca; memset(&ca, 0, sizeof(ca));
            ca.B = n.B; ca.mode = CFM_TBWD; ca.nt = nt; ca.K = 3; ca.n = (float)((double)M * n.g[0].H * n.g[0].W);
            ca.part = c.sc.rpart; ca.coef = pb.coef[0]; ca.dg = HV + n.offG[0]; ca.dbeta = HV + n.offB[0]; ca.gstride = n.PSZ;
            TRYP(FUMI_PH_CONV_EW, launch_coef(c.st, ca, c.sc.dsum));
            TRYP(FUMI_PH_CONV_FIRST, launch_c1(c.st, fa, 2, 1));
            TRY(launch_reduce_batched(c.st, n.B, nt, 2048, c.sc.wpart, 1.f, HV + n.offW[0], n.PSZ));
            continue;
        }
        BwdRedArgs ra; ra.e = e; ra.u = pb.u[l]; ra.ud = tb.ud[l]; ra.dxo = pb.dx[l]; ra.dxod = tb.dxd[l]; ra.coef = pb.coef[l];
        ra.part = c.sc.rpart; ra.nt = ew_bwd_red_nt(e);
        TRYP(FUMI_PH_CONV_EW, launch_bwd_reduce(c.st, ra, 1));
        CoefArgs ca; memset(&ca, 0, sizeof(ca));
        ca.B = n.B; ca.mode = CFM_TBWD; ca.nt = ra.nt; ca.K = 3; ca.n = (float)((double)M * n.g[l].H * n.g[l].W);
        ca.part = c.sc.rpart; ca.coef = pb.coef[l]; ca.dg = HV + n.offG[l]; ca.dbeta = HV + n.offB[l]; ca.gstride = n.PSZ;
        TRYP(FUMI_PH_CONV_EW, launch_coef(c.st, ca, c.sc.dsum));
        BwdApplyArgs aa; aa.e = e; aa.u = pb.u[l]; aa.ud = tb.ud[l]; aa.dxo = pb.dx[l]; aa.dxod = tb.dxd[l]; aa.coef = pb.coef[l]; aa.du = tb.dud[l];
        TRYP(FUMI_PH_CONV_EW, launch_bwd_apply(c.st, aa, 1));
        const int ns = cv_wgrad_nsplit(n.B, npix);
        if (l == 0) {
            Wgrad1Args wa; wa.B = n.B; wa.M = M; wa.Cin = n.Cin; wa.nsplit = ns; wa.g = n.g[0]; wa.img = img; wa.dy = tb.dud[0]; wa.part = c.sc.wpart;
            TRYP(FUMI_PH_CONV_FIRST, launch_wgrad1(c.st, wa));
            TRY(launch_reduce_batched(c.st, n.B, ns, 2048, c.sc.wpart, 1.f, HV + n.offW[0], n.PSZ));
        } else {
            Wgrad64Args wa; wa.B = n.B; wa.nsrc = 2; wa.nsplit = ns; wa.npix = npix; wa.g = n.g[l];
            wa.x[0] = pb.x[l - 1]; wa.dy[0] = tb.dud[l]; wa.x[1] = tb.xd[l - 1]; wa.dy[1] = pb.du[l]; wa.part = c.sc.wpart;
            TRYP(FUMI_PH_CONV_GEMM, launch_wgrad64(c.st, wa));
            TRY(launch_reduce_batched(c.st, n.B, ns, 36864, c.sc.wpart, 1.f, HV + n.offW[l], n.PSZ));
            Conv64Args a; a.B = n.B; a.nsrc = 2; a.npix = npix; a.g = n.g[l];
            a.in[0] = tb.dud[l]; a.frag[0] = frags + n.fB[l]; a.frag_stride[0] = n.FSZ;
            a.in[1] = pb.du[l]; a.frag[1] = Vfrags + n.fB[l]; a.frag_stride[1] = n.FSZ;
            a.out = tb.dxd[l - 1]; a.stats = nullptr; a.dot = nullptr;
            TRYP(FUMI_PH_CONV_GEMM, launch_conv64(c.st, a));
        }
    }
    return FUMI_OK;
}

constexpr int CV_MAXTAPE = 32;                   // taped inner steps of a second-order step
// last step's buffer table, for fumi_hip_conv4_probe (tests compare every intermediate with oracle/conv4_manual.py)
struct ProbeTab {
    bool valid; fumi_ws* ws; char* base;            // the workspace (and its slab) the pointers below were carved from
    Net n; int T, S, Qn; int ntape;
    PassBufs tape[CV_MAXTAPE]; PassBufs query; TanBufs tan;
    float* params; float* heads; float* G; float* dh; float* bar; float* barh; float* HV; float* HVh;
};
static ProbeTab g_probe;

}  // namespace

struct Conv4Problem {
    int B, N, S, Qn, Cin, H, W, nblk, T;
    float alpha, grad_scale;
    int need_grad, second_order;
    const float* x_s; const int64_t* y_s; const float* x_q; const int64_t* y_q;
    const float* theta[3 * CV_MAXBLK];      // W [64][Cin|64][3][3], BN weight [64], BN bias [64] per block
    const float* head;                      // [B][N][F+1]
    float* logits_q; int64_t* preds_q; float* preds_f; float* loss_b; float* acc_b; float* stats;
    float* g_theta[3 * CV_MAXBLK];
    float* head_bar;                        // [B][N][F+1] d loss_b / d head_b (unscaled)
};

int conv4_feature_dim(int nblk, int H, int W) {
    for (int l = 0; l < nblk; ++l) { H /= 2; W /= 2; }
    return 64 * H * W;
}

static size_t conv4_scratch_sizes(const Net& n, int S, int Qn, Scratch& sc) {
    size_t cp = 0, wp = 0, rp = 0;
    const int Ms[2] = {S, Qn};
    for (int mi = 0; mi < 2; ++mi)
        for (int l = 0; l < n.nblk; ++l) {
            const long npix = (long)Ms[mi] * n.g[l].Pp;
            const size_t c1 = (size_t)n.B * cv_tiles(npix) * 128;
            if (c1 > cp) cp = c1;
            const size_t w1 = (size_t)n.B * cv_wgrad_nsplit(n.B, npix) * (l == 0 ? 2048 : 36864);
            if (w1 > wp) wp = w1;
            const size_t r1 = (size_t)n.B * ew_bwd_red_nt(ew_geom(n, Ms[mi], l)) * 3 * 64;
            if (r1 > rp) rp = r1;
            if (l == 0 && n.fused1) {
                int chunk;
                const size_t nt = (size_t)c1_chunks(n.B, Ms[mi], n.g[0], &chunk);
                if (n.B * nt * 192 > rp) rp = n.B * nt * 192;
                if (n.B * nt * 2048 > wp) wp = n.B * nt * 2048;
            }
        }
    sc.cpart_n = cp; sc.wpart_n = wp; sc.rpart_n = rp;
    sc.dsum_n = coef_scratch_doubles(n.B, (int)(cp / ((size_t)n.B * 128)) + 1) + coef_scratch_doubles(n.B, (int)(rp / ((size_t)n.B * 192)) + 1);
    sc.rowl_n = 2 * (size_t)n.B * (S > Qn ? S : Qn);
    return ws_align(cp * 4) + ws_align(wp * 4) + ws_align(rp * 4) + ws_align(sc.dsum_n * 8) + ws_align(sc.rowl_n * 4);
}

// `prepare(extra)` runs once the slab is reserved: the caller carves its own buffers (head, head_bar, the head's producer's
// activations) from the `extra_bytes` region, launches the producer of the head and returns the two pointers.
struct Conv4Hooks {
    size_t extra_bytes;
    std::function<int(char* extra, const float** head, float** head_bar)> prepare;
};

int run_conv4_episodes(fumi_ws* ws, hipStream_t st, Conv4Problem p, const Conv4Hooks& hooks) {
    const size_t extra_bytes = ws_align(hooks.extra_bytes);
    // LANES (as rn12.hip): the parts of the meta-batch are independent until their meta-gradients are added, so parts 1.. run on
    // streams of their own (ws_lane_stream), each in its own part of the workspace, beside part 0 on the caller's stream -- the
    // MFMA-bound GEMMs of one fill the HBM-bound element-wise passes and partial last rounds of the others (proxy: one process of
    // 32 episodes 573.3, two of 16 617.8, four of 8 645 episodes/s; in one process 1 / 2 / 3 / 4 lanes: 571.4 / 616.5 / 625.5 /
    // 594.5, whatever GPU_MAX_HW_QUEUES says).  Up to three lanes of at least 4 episodes by default; FUMI_CV_LANES=n (<= 4) caps
    // them; phase timing or a small batch: one lane, and only then the probe table (fumi_hip_conv4_probe) is filled.
    constexpr int MAXLANES = 4;
    static const int lanes_env0 = getenv("FUMI_CV_LANES") ? atoi(getenv("FUMI_CV_LANES")) : 3;
    const int lanes_env = g_cv_lanes > 0 ? g_cv_lanes : lanes_env0;
    int lanes = 1;
    if (lanes_env >= 2 && !ws->profiling && ws->side) {
        lanes = p.B / 4;
        lanes = lanes < 1 ? 1 : (lanes > MAXLANES ? MAXLANES : lanes);
        lanes = lanes > lanes_env ? lanes_env : lanes;
    }
    for (int i = 1; i < lanes; ++i) if (!ws_lane_stream(ws, i)) { lanes = 1; break; }
    StepCtx cx[MAXLANES];
    for (int i = 0; i < MAXLANES; ++i) { cx[i].ws = ws; cx[i].st = i ? ws->lanes[i - 1] : st; }
    const int Bc = (p.B + lanes - 1) / lanes;                         // episodes per lane
    int rc = net_init(cx[0].n, Bc, p.nblk, p.Cin, p.N, p.H, p.W);
    if (rc) return rc;
    if (p.T < 0 || p.S < 1 || p.Qn < 1) return FUMI_EINVAL;
    const bool grad = p.need_grad != 0, second = grad && p.second_order && p.T > 0;
    // taped inner steps: the tape of every step stays resident (2 GiB per step at 32 episodes x 25 support images), so the cap is
    // the table's size, not memory; the reference's defaults are 5 (train) and 100 (test: no tape), fumi/utils/utils.py:171-179
    if (second && p.T > CV_MAXTAPE) return FUMI_ENOTSUP;             // (the host wrappers say so in words before they call)
    const int ntape = second ? p.T : 1;
    const int nslot = second ? p.T + 1 : 2;
    const size_t F1 = (size_t)cx[0].n.N * (cx[0].n.F + 1);
    // ---- workspace: the caller's region, then one region per lane (sized for Bc episodes)
    auto lane_bytes = [&](const Net& n, Scratch& sc) {
        const size_t hsz = (size_t)n.B * F1;
        size_t bytes = conv4_scratch_sizes(n, p.S, p.Qn, sc);
        bytes += (size_t)ntape * pass_bytes(n, p.S, true) + pass_bytes(n, p.Qn, grad);
        if (second) bytes += tan_bytes(n, p.S);
        bytes += (size_t)nslot * (ws_align((size_t)n.B * n.PSZ * 4) + ws_align((size_t)n.B * n.FSZ * 4) + ws_align(hsz * 4));
        bytes += 4 * ws_align((size_t)n.B * n.PSZ * 4) + 4 * ws_align(hsz * 4) + ws_align((size_t)n.B * n.FSZ * 4);   // G, bar, HV, tmp | dh, barh, HVh | Vfrags
        bytes += ws_align((size_t)n.B * (2048 + 4096) * 4) + ws_align((size_t)(n.nblk) * 36864 * 4) + ws_align((size_t)n.PSZ * 4);
        return ws_align(bytes + 4096);
    };
    const size_t region = lane_bytes(cx[0].n, cx[0].sc);
    if ((rc = ws_reserve(ws, extra_bytes + lanes * region))) return rc;
    char* extra = ws->base + ws->off;
    ws->off += extra_bytes;                                           // (the caller's buffers: head, hypernetwork activations)
    const size_t lane0_off = ws->off;
    TRY(hooks.prepare(extra, &p.head, &p.head_bar));
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(ws->ev[2], st));                       // the lanes start behind the head's producer on the caller's stream
        for (int i = 1; i < lanes; ++i) HIP_TRY(hipStreamWaitEvent(ws->lanes[i - 1], ws->ev[2], 0));
    }
    ProbeTab& pt = g_probe;
    pt.valid = false;
    float* gsum_lane[MAXLANES] = {nullptr, nullptr, nullptr, nullptr};
    fumi_ws* const ws_real = ws;
    // episodes [b0, b0 + bc) on lane `lane`
    auto chunk_body = [&](int lane, int b0, int bc) -> int {
        StepCtx& c = cx[lane];
        const hipStream_t st = c.st;                                  // (shadows: everything of this half goes to its lane's stream)
        int r = net_init(c.n, bc, p.nblk, p.Cin, p.N, p.H, p.W);
        if (r) return r;
        const Net& n = c.n;
        const size_t hsz = (size_t)n.B * F1;
        (void)lane_bytes(n, c.sc);                                    // (the scratch sizes of this half)
        fumi_ws view = *ws_real;                                      // (a private bump pointer)
        fumi_ws* ws = &view;
        ws->off = lane0_off + (size_t)lane * region;
        const float* x_s = p.x_s + (size_t)b0 * p.S * p.Cin * p.H * p.W; const float* x_q = p.x_q + (size_t)b0 * p.Qn * p.Cin * p.H * p.W;
        const int64_t* y_s = p.y_s + (size_t)b0 * p.S; const int64_t* y_q = p.y_q + (size_t)b0 * p.Qn;
        c.S = p.S; c.Qn = p.Qn; c.img_s = x_s; c.img_q = x_q; c.y_s = y_s; c.y_q = y_q;
        c.sc.cpart = ws_f(ws, c.sc.cpart_n); c.sc.wpart = ws_f(ws, c.sc.wpart_n); c.sc.rpart = ws_f(ws, c.sc.rpart_n);
        c.sc.dsum = (double*)ws_f(ws, c.sc.dsum_n * 2); c.sc.rowl = ws_f(ws, c.sc.rowl_n);
        PassBufs tape_l[CV_MAXTAPE]; PassBufs query_l; TanBufs tan_l;
        PassBufs* tape = lanes == 1 ? pt.tape : tape_l; PassBufs& query = lanes == 1 ? pt.query : query_l; TanBufs& tan = lanes == 1 ? pt.tan : tan_l;
        for (int t = 0; t < ntape; ++t) pass_carve(ws, n, p.S, true, tape[t]);
        pass_carve(ws, n, p.Qn, grad, query);
        if (second) tan_carve(ws, n, p.S, tan);
        float* params = ws_f(ws, (size_t)nslot * n.B * n.PSZ);
        float* frags = ws_f(ws, (size_t)nslot * n.B * n.FSZ);
        float* heads = ws_f(ws, (size_t)nslot * hsz);
        float* G = ws_f(ws, (size_t)n.B * n.PSZ); float* bar = ws_f(ws, (size_t)n.B * n.PSZ); float* HV = ws_f(ws, (size_t)n.B * n.PSZ);
        float* dh = ws_f(ws, hsz); float* barh = ws_f(ws, hsz); float* HVh = ws_f(ws, hsz);
        float* Vfrags = ws_f(ws, (size_t)n.B * n.FSZ);
        float* tmp1 = ws_f(ws, (size_t)n.B * (2048 + 4096));
        float* toi_tmp = ws_f(ws, (size_t)n.nblk * 36864);
        float* gsum = ws_f(ws, (size_t)n.PSZ);
        gsum_lane[lane] = gsum;
        if (ws->off > lane0_off + (size_t)(lane + 1) * region || ws->off > ws->cap) return FUMI_ENOMEM;
        auto P = [&](int s_) { return params + (size_t)s_ * n.B * n.PSZ; };
        auto Fr = [&](int s_) { return frags + (size_t)s_ * n.B * n.FSZ; };
        auto Hd = [&](int s_) { return heads + (size_t)s_ * hsz; };

        // ---- slot 0: the meta-parameters, one copy per episode (canonical layouts), and the caller's head
        TRY(slot0_from_theta(st, n, p.theta, P(0), toi_tmp));
        TRY(frags_of_slot(st, n, P(0), Fr(0), tmp1));
        HIP_TRY(hipMemcpyAsync(Hd(0), p.head + (size_t)b0 * F1, hsz * 4, hipMemcpyDeviceToDevice, st));

        // ---- inner loop on the support set
        int cur = 0;
        for (int t = 0; t < p.T; ++t) {
            PassBufs& pb = tape[second ? t : 0];
            const int nxt = second ? t + 1 : cur ^ 1;
            TRY(forward_pass(c, p.S, x_s, P(cur), Fr(cur), pb, Hd(cur), y_s, 1.f / p.S, nullptr, nullptr, nullptr, nullptr, nullptr));
            TRY(backward_pass(c, p.S, x_s, Fr(cur), pb, Hd(cur), G, dh));
            TRY(launch_axpy(st, (long)n.B * n.PSZ, P(cur), -p.alpha, G, P(nxt)));
            TRY(launch_axpy(st, (long)hsz, Hd(cur), -p.alpha, dh, Hd(nxt)));
            TRY(frags_of_slot(st, n, P(nxt), Fr(nxt), tmp1));
            cur = nxt;
        }
        // ---- query pass with the adapted parameters
        TRY(forward_pass(c, p.Qn, x_q, P(cur), Fr(cur), query, Hd(cur), y_q, 1.f / p.Qn, p.logits_q + (size_t)b0 * p.Qn * n.N,
                         p.preds_q + (size_t)b0 * p.Qn, p.preds_f ? p.preds_f + (size_t)b0 * p.Qn : nullptr, p.loss_b + b0, p.acc_b + b0));
        if (lanes == 1) {
            pt.n = n; pt.T = p.T; pt.S = p.S; pt.Qn = p.Qn; pt.ntape = ntape; pt.params = params; pt.heads = heads; pt.G = G; pt.dh = dh;
            pt.bar = bar; pt.barh = barh; pt.HV = HV; pt.HVh = HVh; pt.ws = ws_real; pt.base = ws_real->base; pt.valid = true;
        }
        if (!grad) return FUMI_OK;
        TRY(backward_pass(c, p.Qn, x_q, Fr(cur), query, Hd(cur), bar, barh));
        // ---- second-order reverse sweep
        if (second) {
            for (int t = p.T - 1; t >= 0; --t) {
                TRY(frags_of_slot(st, n, bar, Vfrags, tmp1));
                TRY(hvp_pass(c, p.S, x_s, Fr(t), tape[t], tan, Hd(t), bar, Vfrags, barh, 1.f / p.S, HV, HVh));
                TRY(launch_axpy(st, (long)n.B * n.PSZ, bar, -p.alpha, HV, bar));
                TRY(launch_axpy(st, (long)hsz, barh, -p.alpha, HVh, barh));
            }
        }
        // ---- this half's meta-gradient: scaled sum over its episodes
        TRY(launch_reduce_batched(st, 1, n.B, n.PSZ, bar, p.grad_scale, gsum, 0));
        HIP_TRY(hipMemcpyAsync(p.head_bar + (size_t)b0 * F1, barh, hsz * 4, hipMemcpyDeviceToDevice, st));
        return FUMI_OK;
    };
    for (int i = 0; i < lanes; ++i) {
        const int b0 = i * Bc, bc = p.B - b0 < Bc ? p.B - b0 : Bc;
        if (bc > 0) TRY(chunk_body(i, b0, bc));
    }
    for (int i = 1; i < lanes; ++i) {                                 // the caller's stream waits for the other lanes
        HIP_TRY(hipEventRecord(ws->lane_ev[i - 1], ws->lanes[i - 1]));
        HIP_TRY(hipStreamWaitEvent(st, ws->lane_ev[i - 1], 0));
    }
    const Net& n = cx[0].n;
    if (p.stats) {
        ReduceSegs sg; sg.n = 0; sg.scale = p.grad_scale;
        sg.add(p.loss_b, p.B, 1, 1, p.stats); sg.add(p.acc_b, p.B, 1, 1, p.stats + 1);
        TRY(launch_reduce_multi(st, sg));
    }
    if (!grad) return FUMI_OK;
    // ---- meta-gradients: the lanes' sums added, back to the parameters' own layouts
    float* gsum = gsum_lane[0];
    for (int i = 1; i < lanes; ++i) if (gsum_lane[i]) TRY(launch_axpy(st, n.PSZ, gsum, 1.f, gsum_lane[i], gsum));
    TRY(launch_w1_from_canon(st, n.Cin, gsum + n.offW[0], p.g_theta[0], 1.f));
    for (int l = 0; l < n.nblk; ++l) {
        if (l) TRY(launch_toi_to_oihw(st, 1, gsum + n.offW[l], p.g_theta[3 * l], 1.f));
        HIP_TRY(hipMemcpyAsync(p.g_theta[3 * l + 1], gsum + n.offG[l], 64 * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(p.g_theta[3 * l + 2], gsum + n.offB[l], 64 * 4, hipMemcpyDeviceToDevice, st));
    }
    return FUMI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
static int fill_problem(Conv4Problem& p, int B, int N, int S, int Qn, int Cin, int H, int W, int nblk, int T, float alpha,
                        int need_grad, int second_order, float grad_scale, const float* x_s, const int64_t* y_s,
                        const float* x_q, const int64_t* y_q, const float* const* theta, float* logits_q, int64_t* preds_q,
                        float* preds_f, float* loss_b, float* acc_b, float* stats, float* const* g_theta) {
    memset(&p, 0, sizeof(p));
    if (!x_s || !y_s || !x_q || !y_q || !theta || !logits_q || !preds_q || !loss_b || !acc_b) return FUMI_EINVAL;
    if (nblk < 1 || nblk > CV_MAXBLK || (need_grad && !g_theta)) return FUMI_EINVAL;
    p.B = B; p.N = N; p.S = S; p.Qn = Qn; p.Cin = Cin; p.H = H; p.W = W; p.nblk = nblk; p.T = T; p.alpha = alpha;
    p.grad_scale = grad_scale; p.need_grad = need_grad ? 1 : 0; p.second_order = second_order ? 1 : 0;
    p.x_s = x_s; p.y_s = y_s; p.x_q = x_q; p.y_q = y_q;
    for (int i = 0; i < 3 * nblk; ++i) {
        if (!theta[i] || (need_grad && !g_theta[i])) return FUMI_EINVAL;
        p.theta[i] = theta[i]; p.g_theta[i] = need_grad ? g_theta[i] : nullptr;
    }
    p.logits_q = logits_q; p.preds_q = preds_q; p.preds_f = preds_f; p.loss_b = loss_b; p.acc_b = acc_b; p.stats = stats;
    return FUMI_OK;
}

extern "C" {

int fumi_hip_conv4_set_option(int key, int value) {
    if (key == 0) { g_c1_fused = value ? 1 : 0; return FUMI_OK; }
    if (key == 1) { if (value < 0 || value > 4) return FUMI_EINVAL; g_cv_lanes = value; return FUMI_OK; }
    return FUMI_EINVAL;
}

int fumi_hip_conv4_feature_dim(int nblk, int H, int W) { return (nblk < 1 || nblk > CV_MAXBLK) ? FUMI_EINVAL : conv4_feature_dim(nblk, H, W); }

int fumi_hip_maml_conv4_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk,
        int T, float alpha, int first_order, int need_grad, float grad_scale,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* const* params,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_params) {
    if (!ws || !params) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    Conv4Problem p;
    int rc = fill_problem(p, B, N, S, Qn, Cin, H, W, nblk, T, alpha, need_grad, !first_order, grad_scale, x_s, y_s, x_q, y_q, params,
                          logits_q, preds_q, preds_q_f32, loss_b, acc_b, stats, g_params);
    if (rc) return rc;
    const float* Wf = params[3 * nblk]; const float* bf = params[3 * nblk + 1];
    if (!Wf || !bf || (need_grad && (!g_params[3 * nblk] || !g_params[3 * nblk + 1]))) return FUMI_EINVAL;
    const int F = conv4_feature_dim(nblk, H, W);
    if (F < 64) return FUMI_EINVAL;
    const size_t hsz = (size_t)B * N * (F + 1);
    float* hbar = nullptr;
    Conv4Hooks hk;
    hk.extra_bytes = 2 * ws_align(hsz * 4);
    hk.prepare = [&](char* extra, const float** head, float** head_bar) -> int {
        float* h = (float*)extra;
        hbar = (float*)(extra + ws_align(hsz * 4));
        *head = h; *head_bar = hbar;
        return launch_broadcast_head(st, B, N, F, Wf, bf, h);            // every episode starts from lin_final (maml.py:24-31)
    };
    if ((rc = run_conv4_episodes(ws, st, p, hk))) { for (int i = 0; i < 3; ++i) if (ws->lanes[i]) (void)hipStreamSynchronize(ws->lanes[i]); return rc; }
    if (!need_grad) return FUMI_OK;
    return launch_split_head_grad(st, B, N, F, hbar, grad_scale, g_params[3 * nblk], g_params[3 * nblk + 1]);
}

int fumi_hip_fumi_conv4_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi) {
    if (!ws || !theta || !phi || (!cls_text && !text_s) || Dt < 1 || Ht < 1) return FUMI_EINVAL;
    if (need_grad && !g_phi) return FUMI_EINVAL;
    for (int i = 0; i < 4; ++i) if (!phi[i] || (need_grad && !g_phi[i])) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    Conv4Problem p;
    int rc = fill_problem(p, B, N, S, Qn, Cin, H, W, nblk, T, alpha, need_grad, 1, grad_scale, x_s, y_s, x_q, y_q, theta,
                          logits_q, preds_q, preds_q_f32, loss_b, acc_b, stats, g_theta);       // fumi.py:176: second order always
    if (rc) return rc;
    const int F = conv4_feature_dim(nblk, H, W);
    if (F < 64) return FUMI_EINVAL;
    const int R = B * N, H1 = F + 1;
    // hypernetwork rows are (episode, class) pairs: Linear(Dt, Ht) . ReLU . Linear(Ht, F+1) [. Tanh]  (fumi.py:70-86,104-113)
    float *c = nullptr, *u = nullptr, *ub = nullptr, *h = nullptr, *hbar = nullptr, *hpb = nullptr;
    const float* ctext = cls_text;
    Conv4Hooks hk;
    hk.extra_bytes = ws_align((size_t)R * Dt * 4) + 2 * ws_align((size_t)R * Ht * 4) + 3 * ws_align((size_t)R * H1 * 4);
    hk.prepare = [&](char* extra, const float** head, float** head_bar) -> int {
        auto take = [&](size_t nfl) { float* q = (float*)extra; extra += ws_align(nfl * 4); return q; };
        c = take((size_t)R * Dt); u = take((size_t)R * Ht); ub = take((size_t)R * Ht);
        h = take((size_t)R * H1); hbar = take((size_t)R * H1); hpb = take((size_t)R * H1);
        int r2;
        if (!ctext) {                                                    // first support row of each class (fumi.py:207-210)
            if ((r2 = launch_class_text_select(st, B, N, S, Dt, text_s, y_s, c, ws->status))) return r2;
            ctext = c;
        }
        GemmArgs g = gemm_args(R, Ht, Dt, ctext, Dt, phi[0], Dt, u, Ht);
        g.bias = phi[1]; g.act = 1;
        if ((r2 = launch_gemm(st, g, 0, 0))) return r2;
        g = gemm_args(R, H1, Ht, u, Ht, phi[2], Ht, h, H1);
        g.bias = phi[3]; g.act = tanh_head ? 2 : 0;
        if ((r2 = launch_gemm(st, g, 0, 0))) return r2;
        *head = h; *head_bar = hbar;
        return FUMI_OK;
    };
    if ((rc = run_conv4_episodes(ws, st, p, hk))) { for (int i = 0; i < 3; ++i) if (ws->lanes[i]) (void)hipStreamSynchronize(ws->lanes[i]); return rc; }
    if (!need_grad) return FUMI_OK;
    const float* hp = hbar;
    if (tanh_head) { if ((rc = launch_tanh_bwd(st, (long)R * H1, h, hbar, hpb))) return rc; hp = hpb; }
    GemmArgs g = gemm_args(H1, Ht, R, hp, H1, u, Ht, g_phi[2], Ht);    // gA1 = hp^T u
    g.alpha = grad_scale;
    if ((rc = launch_gemm(st, g, 1, 1))) return rc;
    if ((rc = launch_colsum(st, hp, R, H1, H1, grad_scale, g_phi[3]))) return rc;
    g = gemm_args(R, Ht, H1, hp, H1, phi[2], Ht, ub, Ht);              // ubar = (hp A1) * relu'(u)
    g.mask = u;
    if ((rc = launch_gemm(st, g, 0, 1))) return rc;
    g = gemm_args(Ht, Dt, R, ub, Ht, ctext, Dt, g_phi[0], Dt);         // gA0 = ubar^T c
    g.alpha = grad_scale;
    if ((rc = launch_gemm(st, g, 1, 1))) return rc;
    if (float* tg = ws->text_grad) {                                   // armed by fumi_hip_want_text_grad: scale * ubar A0  [R,Dt]
        ws->text_grad = nullptr;
        g = gemm_args(R, Dt, Ht, ub, Ht, phi[0], Dt, tg, Dt);
        g.alpha = grad_scale;
        if ((rc = launch_gemm(st, g, 0, 1))) return rc;
    }
    return launch_colsum(st, ub, R, Ht, Ht, grad_scale, g_phi[1]);
}

// Copies one intermediate of the LAST conv4 step of this process out of the workspace (tests compare every tensor of the
// sweep with oracle/conv4_manual.py).  pass: 0..T-1 = support step t, T = query pass, T+1 = tangent scratch of the last HVP
// (inner step 0).  kind: 0 u, 1 x (pooled output / features), 2 du, 3 dx, 4 coef [B][16][64], 5 p, 6 dz;  tangent pass: 0 u',
// 1 x', 2 du', 3 dx', 6 dz'.  pass = -1: kind 0 parameter slabs [slot=block][B][PSZ], 1 heads [slot][B][N][F+1], 2 G, 3 dh,
// 4 bar, 5 bar_h, 6 HV, 7 HV_h  (block = slot for kinds 0 / 1).  Returns the number of floats in *n_out (copies min(n, max)).
int fumi_hip_conv4_probe(fumi_ws_t* ws, fumi_stream_t stream, int pass, int kind, int block, float* out, size_t max_floats,
                         size_t* n_out) {
    if (!ws || !out || !n_out || !g_probe.valid) return FUMI_EINVAL;
    const ProbeTab& pt = g_probe;
    if (pt.ws != ws || pt.base != ws->base) return FUMI_EINVAL;         // another workspace's step, or the slab has moved since
    const Net& n = pt.n;
    const float* src = nullptr; size_t cnt = 0;
    const size_t hsz = (size_t)n.B * n.N * (n.F + 1);
    if (pass == -1) {
        switch (kind) {
            case 0: src = pt.params + (size_t)block * n.B * n.PSZ; cnt = (size_t)n.B * n.PSZ; break;
            case 1: src = pt.heads + (size_t)block * hsz; cnt = hsz; break;
            case 2: src = pt.G; cnt = (size_t)n.B * n.PSZ; break;
            case 3: src = pt.dh; cnt = hsz; break;
            case 4: src = pt.bar; cnt = (size_t)n.B * n.PSZ; break;
            case 5: src = pt.barh; cnt = hsz; break;
            case 6: src = pt.HV; cnt = (size_t)n.B * n.PSZ; break;
            case 7: src = pt.HVh; cnt = hsz; break;
            default: return FUMI_EINVAL;
        }
    } else if (pass == pt.T + 1) {
        if (block < 0 || block >= n.nblk) return FUMI_EINVAL;
        const int M = pt.S;
        switch (kind) {
            case 0: src = pt.tan.ud[block]; cnt = act_floats(n, M, block); break;
            case 1: src = pt.tan.xd[block]; cnt = out_floats(n, M, block); break;
            case 2: src = pt.tan.dud[block]; cnt = act_floats(n, M, block); break;
            case 3: src = pt.tan.dxd[block]; cnt = out_floats(n, M, block); break;
            case 6: src = pt.tan.dzd; cnt = (size_t)n.B * M * n.N; break;
            default: return FUMI_EINVAL;
        }
    } else {
        if (pass < 0 || pass > pt.T || block < 0 || block >= n.nblk) return FUMI_EINVAL;
        const PassBufs& pb = pass == pt.T ? pt.query : pt.tape[pass < pt.ntape ? pass : 0];
        const int M = pb.M;
        switch (kind) {
            case 0: src = pb.u[block]; cnt = act_floats(n, M, block); break;
            case 1: src = pb.x[block]; cnt = out_floats(n, M, block); break;
            case 2: src = pb.du[block]; cnt = act_floats(n, M, block); break;
            case 3: src = pb.dx[block]; cnt = out_floats(n, M, block); break;
            case 4: src = pb.coef[block]; cnt = (size_t)n.B * CF_N * 64; break;
            case 5: src = pb.p; cnt = (size_t)n.B * M * n.N; break;
            case 6: src = pb.dz; cnt = (size_t)n.B * M * n.N; break;
            default: return FUMI_EINVAL;
        }
    }
    if (!src) return FUMI_ENOTSUP;                                      // (block-1 maps do not exist on the fused path)
    *n_out = cnt;
    const size_t k = cnt < max_floats ? cnt : max_floats;
    HIP_TRY(hipMemcpyAsync(out, src, k * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FUMI_OK;
}

// Forward only: feats [G*M, F] = Conv4(x [G, M, Cin, H, W]) with the batch statistics of every group of M images taken
// separately (a support set or a query set is one group, as in the meta-step).  Serves Conv4.forward / FUMI.im_forward.
int fumi_hip_conv4_features(fumi_ws_t* ws, fumi_stream_t stream, int G, int M, int Cin, int H, int W, int nblk,
        const float* x, const float* const* theta, float* feats) {
    if (!ws || !x || !theta || !feats || G < 1 || M < 1) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    StepCtx c; c.ws = ws; c.st = st;
    int rc = net_init(c.n, G, nblk, Cin, 1, H, W);
    if (rc) return rc;
    const Net& n = c.n;
    for (int i = 0; i < 3 * nblk; ++i) if (!theta[i]) return FUMI_EINVAL;
    size_t bytes = conv4_scratch_sizes(n, M, M, c.sc) + pass_bytes(n, M, false);
    bytes += ws_align((size_t)n.B * n.PSZ * 4) + ws_align((size_t)n.B * n.FSZ * 4) + ws_align((size_t)n.B * (2048 + 4096) * 4) +
             ws_align((size_t)n.nblk * 36864 * 4);
    if ((rc = ws_reserve(ws, bytes))) return rc;
    g_probe.valid = false;
    c.sc.cpart = ws_f(ws, c.sc.cpart_n); c.sc.wpart = ws_f(ws, c.sc.wpart_n); c.sc.rpart = ws_f(ws, c.sc.rpart_n);
    c.sc.dsum = (double*)ws_f(ws, c.sc.dsum_n * 2); c.sc.rowl = ws_f(ws, c.sc.rowl_n);
    PassBufs pb; pass_carve(ws, n, M, false, pb);
    float* params = ws_f(ws, (size_t)n.B * n.PSZ); float* frags = ws_f(ws, (size_t)n.B * n.FSZ);
    float* tmp1 = ws_f(ws, (size_t)n.B * (2048 + 4096)); float* toi_tmp = ws_f(ws, (size_t)n.nblk * 36864);
    TRY(slot0_from_theta(st, n, theta, params, toi_tmp));
    TRY(frags_of_slot(st, n, params, frags, tmp1));
    TRY(forward_pass(c, M, x, params, frags, pb, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr));
    HIP_TRY(hipMemcpyAsync(feats, pb.x[n.nblk - 1], (size_t)G * M * n.F * 4, hipMemcpyDeviceToDevice, st));
    return FUMI_OK;
}

// ---- Conv4 as the image encoder in front of a step that owns its own workspace (AM3): forward with the tape kept, backward from
// the feature adjoints.  Every episode's support set and query set is one batch-statistics group, as in the meta-steps above.
// The two calls share the tape through `ws`: encode(keep_tape = 1) lays the activations out in the workspace, encode_bwd carves
// the identical layout again (same arguments) and walks it backwards -- so `ws` must not serve any other call in between (give the
// encoder a workspace of its own: fumi_hip_workspace_create).  A token records what was laid out; a mismatch is FUMI_EINVAL.
namespace {
struct EncodeToken { bool valid; fumi_ws* ws; char* base; int B, S, Qn, Cin, H, W, nblk, lanes; };
EncodeToken g_enc = {false, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 1};
struct EncodeBufs { PassBufs ps, pq; float *params, *frags, *tmp1, *toi_tmp, *Gs, *Gq, *gsum; };

size_t encode_bytes(const Net& n, int S, int Qn, Scratch& sc) {
    size_t b = conv4_scratch_sizes(n, S, Qn, sc) + pass_bytes(n, S, true) + pass_bytes(n, Qn, true);
    b += 3 * ws_align((size_t)n.B * n.PSZ * 4) + ws_align((size_t)n.B * n.FSZ * 4) + ws_align((size_t)n.B * (2048 + 4096) * 4) +
         ws_align((size_t)n.nblk * 36864 * 4) + ws_align((size_t)n.PSZ * 4);
    return b;
}
void encode_carve(fumi_ws* ws, StepCtx& c, int S, int Qn, EncodeBufs& e) {
    const Net& n = c.n;
    c.sc.cpart = ws_f(ws, c.sc.cpart_n); c.sc.wpart = ws_f(ws, c.sc.wpart_n); c.sc.rpart = ws_f(ws, c.sc.rpart_n);
    c.sc.dsum = (double*)ws_f(ws, c.sc.dsum_n * 2); c.sc.rowl = ws_f(ws, c.sc.rowl_n);
    pass_carve(ws, n, S, true, e.ps); pass_carve(ws, n, Qn, true, e.pq);
    e.params = ws_f(ws, (size_t)n.B * n.PSZ);
    e.Gs = ws_f(ws, 2 * (size_t)n.B * n.PSZ); e.Gq = e.Gs + (size_t)n.B * n.PSZ;     // adjacent: summed as 2 B slabs
    e.frags = ws_f(ws, (size_t)n.B * n.FSZ);
    e.tmp1 = ws_f(ws, (size_t)n.B * (2048 + 4096)); e.toi_tmp = ws_f(ws, (size_t)n.nblk * 36864); e.gsum = ws_f(ws, (size_t)n.PSZ);
}
}  // namespace

// lanes of the encoder calls (as run_conv4_episodes: parts of the episodes on streams of their own, each with its own carve).
// AM3 + Conv4 at 32 episodes with 1 / 2 / 3 lanes: 741.8 / 790.2 / 751.5 episodes/s -- two by default here.
static int encode_lanes(fumi_ws* ws, int B) {
    static const int lanes_env0 = getenv("FUMI_CV_LANES") ? atoi(getenv("FUMI_CV_LANES")) : 2;
    const int lanes_env = g_cv_lanes > 0 ? g_cv_lanes : lanes_env0;
    int lanes = 1;
    if (lanes_env >= 2 && !ws->profiling && ws->side) {
        lanes = B / 4;
        lanes = lanes < 1 ? 1 : (lanes > 4 ? 4 : lanes);
        lanes = lanes > lanes_env ? lanes_env : lanes;
    }
    for (int i = 1; i < lanes; ++i) if (!ws_lane_stream(ws, i)) return 1;
    return lanes;
}
static void encode_abandon(fumi_ws* ws) { for (int i = 0; i < 3; ++i) if (ws->lanes[i]) (void)hipStreamSynchronize(ws->lanes[i]); }

// shared frame of the two encoder calls: `lanes` parts of the episodes, part i carved at i * region and run on its lane's stream by
// body(ctx, bufs, b0, bc); fork after everything on the caller's stream, join before the caller's stream goes on
typedef std::function<int(StepCtx&, EncodeBufs&, int, int, int)> EncodeBody;
static int encode_on_lanes(fumi_ws* ws, hipStream_t st, int lanes, int B, int S, int Qn, int Cin, int H, int W, int nblk,
                           const EncodeBody& body) {
    const int Bc = (B + lanes - 1) / lanes;
    StepCtx c0; c0.ws = ws; c0.st = st;
    int rc = net_init(c0.n, Bc, nblk, Cin, 1, H, W);
    if (rc) return rc;
    const size_t region = ws_align(encode_bytes(c0.n, S, Qn, c0.sc) + 4096);
    if ((rc = ws_reserve(ws, lanes * region))) return rc;
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(ws->ev[2], st));
        for (int i = 1; i < lanes; ++i) HIP_TRY(hipStreamWaitEvent(ws->lanes[i - 1], ws->ev[2], 0));
    }
    for (int i = 0; i < lanes; ++i) {
        const int b0 = i * Bc, bc = B - b0 < Bc ? B - b0 : Bc;
        if (bc <= 0) continue;
        StepCtx c; c.ws = ws; c.st = i ? ws->lanes[i - 1] : st;
        if ((rc = net_init(c.n, bc, nblk, Cin, 1, H, W))) return rc;
        (void)encode_bytes(c.n, S, Qn, c.sc);
        fumi_ws view = *ws;
        view.off = (size_t)i * region;
        EncodeBufs e; encode_carve(&view, c, S, Qn, e);
        if (view.off > (size_t)(i + 1) * region || view.off > ws->cap) return FUMI_ENOMEM;
        if ((rc = body(c, e, i, b0, bc))) return rc;
    }
    for (int i = 1; i < lanes; ++i) {
        HIP_TRY(hipEventRecord(ws->lane_ev[i - 1], ws->lanes[i - 1]));
        HIP_TRY(hipStreamWaitEvent(st, ws->lane_ev[i - 1], 0));
    }
    return FUMI_OK;
}

int fumi_hip_conv4_encode(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int Cin, int H, int W, int nblk,
        const float* x_s, const float* x_q, const float* const* theta, float* feats_s, float* feats_q, int keep_tape) {
    if (!ws || !x_s || !x_q || !theta || !feats_s || !feats_q || B < 1 || S < 1 || Qn < 1) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    for (int i = 0; i < 3 * nblk; ++i) if (!theta[i]) return FUMI_EINVAL;
    g_enc.valid = false; g_probe.valid = false;
    const int lanes = encode_lanes(ws, B);
    const size_t img = (size_t)Cin * H * W;
    int rc = encode_on_lanes(ws, st, lanes, B, S, Qn, Cin, H, W, nblk, [&](StepCtx& c, EncodeBufs& e, int, int b0, int bc) -> int {
        const Net& n = c.n;
        const hipStream_t st = c.st;
        const float* xs = x_s + (size_t)b0 * S * img; const float* xq = x_q + (size_t)b0 * Qn * img;
        TRY(slot0_from_theta(st, n, theta, e.params, e.toi_tmp));
        TRY(frags_of_slot(st, n, e.params, e.frags, e.tmp1));
        TRY(forward_pass(c, S, xs, e.params, e.frags, e.ps, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr));
        TRY(forward_pass(c, Qn, xq, e.params, e.frags, e.pq, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr));
        HIP_TRY(hipMemcpyAsync(feats_s + (size_t)b0 * S * n.F, e.ps.x[n.nblk - 1], (size_t)bc * S * n.F * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(feats_q + (size_t)b0 * Qn * n.F, e.pq.x[n.nblk - 1], (size_t)bc * Qn * n.F * 4, hipMemcpyDeviceToDevice, st));
        return FUMI_OK;
    });
    if (rc) { encode_abandon(ws); return rc; }
    if (keep_tape) g_enc = EncodeToken{true, ws, ws->base, B, S, Qn, Cin, H, W, nblk, lanes};
    return FUMI_OK;
}

// g_theta (3 nblk pointers, torch layouts) = scale * d/dtheta of sum over images <dfeats, feats>, summed over the episodes
int fumi_hip_conv4_encode_bwd(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int Qn, int Cin, int H, int W, int nblk,
        const float* x_s, const float* x_q, const float* dfeats_s, const float* dfeats_q, float scale, float* const* g_theta) {
    if (!ws || !x_s || !x_q || !dfeats_s || !dfeats_q || !g_theta) return FUMI_EINVAL;
    const EncodeToken t = g_enc;
    if (!t.valid || t.ws != ws || t.base != ws->base || t.B != B || t.S != S || t.Qn != Qn || t.Cin != Cin || t.H != H || t.W != W ||
        t.nblk != nblk) return FUMI_EINVAL;                       // no tape of this shape in this workspace
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    for (int i = 0; i < 3 * nblk; ++i) if (!g_theta[i]) return FUMI_EINVAL;
    g_enc.valid = false;                                           // the backward pass overwrites parts of the tape
    const int lanes = t.lanes;                                     // the layout the forward call left
    for (int i = 1; i < lanes; ++i) if (!ws_lane_stream(ws, i)) return FUMI_EHIP;
    const size_t img = (size_t)Cin * H * W;
    float* gsum_lane[4] = {nullptr, nullptr, nullptr, nullptr};
    Net n0;
    int rc = encode_on_lanes(ws, st, lanes, B, S, Qn, Cin, H, W, nblk, [&](StepCtx& c, EncodeBufs& e, int lane, int b0, int bc) -> int {
        const Net& n = c.n;
        const hipStream_t st = c.st;
        if (ws->base != t.base) return FUMI_EINVAL;
        const float* xs = x_s + (size_t)b0 * S * img; const float* xq = x_q + (size_t)b0 * Qn * img;
        HIP_TRY(hipMemcpyAsync(e.ps.dx[n.nblk - 1], dfeats_s + (size_t)b0 * S * n.F, (size_t)bc * S * n.F * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(e.pq.dx[n.nblk - 1], dfeats_q + (size_t)b0 * Qn * n.F, (size_t)bc * Qn * n.F * 4, hipMemcpyDeviceToDevice, st));
        TRY(backward_pass(c, S, xs, e.frags, e.ps, nullptr, e.Gs, nullptr));
        TRY(backward_pass(c, Qn, xq, e.frags, e.pq, nullptr, e.Gq, nullptr));
        // Gs and Gq are adjacent [bc][PSZ] slabs: one sum over 2 bc parameter slabs
        TRY(launch_reduce_batched(st, 1, 2 * n.B, n.PSZ, e.Gs, scale, e.gsum, 0));
        gsum_lane[lane] = e.gsum;
        if (lane == 0) n0 = n;
        return FUMI_OK;
    });
    if (rc) { encode_abandon(ws); return rc; }
    const Net& n = n0;
    float* gsum = gsum_lane[0];
    for (int i = 1; i < lanes; ++i) if (gsum_lane[i]) TRY(launch_axpy(st, n.PSZ, gsum, 1.f, gsum_lane[i], gsum));
    TRY(launch_w1_from_canon(st, n.Cin, gsum + n.offW[0], g_theta[0], 1.f));
    for (int l = 0; l < n.nblk; ++l) {
        if (l) TRY(launch_toi_to_oihw(st, 1, gsum + n.offW[l], g_theta[3 * l], 1.f));
        HIP_TRY(hipMemcpyAsync(g_theta[3 * l + 1], gsum + n.offG[l], 64 * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(g_theta[3 * l + 2], gsum + n.offB[l], 64 * 4, hipMemcpyDeviceToDevice, st));
    }
    return FUMI_OK;
}

// ---- finer-grained ops --------------------------------------------------------------------------------------------------
static int conv3x3_common(fumi_ws_t* ws, hipStream_t st, int M, int H, int W, const float* x, const float* Wt, float* y, bool data_grad) {
    if (!ws || !x || !Wt || !y || M < 1 || H < 1 || W < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    const CvGeom g = cv_geom(H, W);
    const size_t act = (size_t)M * g.Pp * 64;
    int rc = ws_reserve(ws, 2 * ws_align(act * 4) + 3 * ws_align((size_t)CV_WFRAG * 4));
    if (rc) return rc;
    float* xp = ws_f(ws, act); float* yp = ws_f(ws, act);
    float* toi = ws_f(ws, CV_WFRAG); float* ff = ws_f(ws, CV_WFRAG); float* fb = ws_f(ws, CV_WFRAG);
    TRY(launch_pad_cl(st, M, g, x, xp));
    TRY(launch_oihw_to_toi(st, 1, Wt, toi));
    TRY(launch_wfrag64(st, 1, toi, 0, ff, fb, 0));
    Conv64Args a; a.B = 1; a.nsrc = 1; a.npix = (long)M * g.Pp; a.g = g;
    a.in[0] = xp; a.frag[0] = data_grad ? fb : ff; a.frag_stride[0] = 0; a.in[1] = nullptr; a.frag[1] = nullptr; a.frag_stride[1] = 0;
    a.out = yp; a.stats = nullptr; a.dot = nullptr;
    TRY(launch_conv64(st, a));
    return launch_unpad_cl(st, M, g, yp, y);
}

int fumi_hip_conv3x3_fwd(fumi_ws_t* ws, fumi_stream_t stream, int M, int H, int W, const float* x, const float* Wt, float* y) {
    return conv3x3_common(ws, (hipStream_t)stream, M, H, W, x, Wt, y, false);
}
int fumi_hip_conv3x3_bwd_data(fumi_ws_t* ws, fumi_stream_t stream, int M, int H, int W, const float* dy, const float* Wt, float* dx) {
    return conv3x3_common(ws, (hipStream_t)stream, M, H, W, dy, Wt, dx, true);
}
int fumi_hip_conv3x3_bwd_weight(fumi_ws_t* ws, fumi_stream_t stream, int M, int H, int W, const float* x, const float* dy, float* dW) {
    if (!ws || !x || !dy || !dW || M < 1 || H < 1 || W < 1) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    const CvGeom g = cv_geom(H, W);
    const size_t act = (size_t)M * g.Pp * 64;
    const long npix = (long)M * g.Pp;
    const int ns = cv_wgrad_nsplit(1, npix);
    int rc = ws_reserve(ws, 2 * ws_align(act * 4) + ws_align((size_t)(ns + 1) * CV_WFRAG * 4));
    if (rc) return rc;
    float* xp = ws_f(ws, act); float* dp = ws_f(ws, act); float* part = ws_f(ws, (size_t)(ns + 1) * CV_WFRAG);
    float* toi = part + (size_t)ns * CV_WFRAG;
    TRY(launch_pad_cl(st, M, g, x, xp));
    TRY(launch_pad_cl(st, M, g, dy, dp));
    Wgrad64Args wa; wa.B = 1; wa.nsrc = 1; wa.nsplit = ns; wa.npix = npix; wa.g = g;
    wa.x[0] = xp; wa.dy[0] = dp; wa.x[1] = nullptr; wa.dy[1] = nullptr; wa.part = part;
    TRY(launch_wgrad64(st, wa));
    TRY(launch_reduce_batched(st, 1, ns, CV_WFRAG, part, 1.f, toi, 0));
    return launch_toi_to_oihw(st, 1, toi, dW, 1.f);
}

int fumi_hip_sgd_axpy(fumi_ws_t* ws, fumi_stream_t stream, long n, const float* p, float step_size, const float* g, float* out) {
    if (!ws || !p || !g || !out || n < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    return launch_axpy((hipStream_t)stream, n, p, -step_size, g, out);
}

int fumi_hip_ce_fwd_bwd(fumi_ws_t* ws, fumi_stream_t stream, int M, int N, const float* z, const int64_t* y, float* loss, float* dz,
        int64_t* preds) {
    if (!ws || !z || !y || !loss || M < 1 || N < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    return launch_ce((hipStream_t)stream, M, N, z, y, loss, dz, preds, ws->status);
}

int fumi_hip_proto_reduce(fumi_ws_t* ws, fumi_stream_t stream, int B, int S, int N, int P, const float* x, const int64_t* y, float* out) {
    if (!ws || !x || !y || !out || B < 1 || S < 1 || N < 1 || P < 1) return FUMI_EINVAL;
    HIP_TRY(hipSetDevice(ws->device));
    return launch_proto((hipStream_t)stream, B, S, N, P, x, y, out, ws->status);
}

}  // extern "C"
