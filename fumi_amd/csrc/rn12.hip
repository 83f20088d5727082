// Meta-step with the bf16 ResNet-12 image encoder at the im_net seam (fumi/models/fumi.py:89-100 is the seam; the inner loop, the
// query loss and the second-order outer gradient are fumi.py:146-192 / maml.py:156-191; BASELINE.json configs[4]).  Algorithm =
// oracle/resnet12_manual.py (forward-over-reverse second order, Pearlmutter), as conv4.hip does it for Conv4:
//
//   for t < T:   forward(theta_t) -> tape_t;  backward -> g_t;  theta_{t+1} = theta_t - alpha g_t          (support set)
//   query:       forward(theta_T) -> logits, loss;  backward -> bar_T
//   for t = T-1 .. 0:  bar_t = bar_{t+1} - alpha H_t bar_{t+1}   (one tangent forward + one tangent backward over tape_t)
//
// Memory plan.  HBM is the tape: every conv output, activation and gradient map of every inner step stays resident in bf16
// (12 maps per conv resolution and image: 4 u, 2 a, 4 du, 2 da, plus the pooled output and its gradient) -- 27 MB per 84 x 84 image
// and step at channels (64, 160, 320, 640), i.e. 13.6 GB for the 100 support images x 5 steps of a 20-way 5-shot episode, plus the
// query pass and one support-sized set of tangent maps.  288 GB hold a handful of such episodes, not a meta-batch of 64: the
// meta-batch is processed in CHUNKS of episodes (episodes are independent given the meta-parameters; the meta-gradient is the sum
// over chunks), the chunk size derived from the workspace budget.  Master weights and all parameter-space vectors are fp32.
#include "rn12.h"
#include "conv4.h"                     // head kernels, axpy / broadcast / batched reduction (conv_ew.hip)
#include <string.h>
#include <stdlib.h>
#include <functional>

namespace {

struct RnLayer { int Cin, Cin_real, Cout, ntaps; long offW, offG, offB, fF, fB; };
struct RnNet {
    int B, nblk, Cimg, N, F;
    int C[RN_MAXBLK];
    RnGeom g[RN_MAXBLK + 1];                       // g[l]: conv resolution of block l; g[nblk]: the last pooled output
    RnLayer L[RN_MAXBLK][RN_NCONV];
    long PSZ, FSZ;                                 // per episode: floats of the parameter slab, bf16 elements of the fragment slab
};

int net_init(RnNet& n, int B, int nblk, int Cimg, int N, int H, int W, const int* channels) {
    if (nblk < 1 || nblk > RN_MAXBLK || Cimg < 1 || Cimg > 8 || B < 1 || N < 1 || !channels) return FUMI_EINVAL;
    n.B = B; n.nblk = nblk; n.Cimg = Cimg; n.N = N;
    long po = 0, fo = 0;
    int cin = 16, cin_real = Cimg;
    for (int l = 0; l < nblk; ++l) {
        const int c = channels[l];
        if (c < 32 || (c & 31) || c > 2048 || H < 2 || W < 2) return FUMI_EINVAL;
        n.C[l] = c; n.g[l] = rn_geom(H, W);
        if (rn_conv_lds_bytes(n.g[l], c) > 160 * 1024) return FUMI_ENOTSUP;
        for (int k = 0; k < RN_NCONV; ++k) {
            RnLayer& y = n.L[l][k];
            y.Cin = (k == 0 || k == 3) ? cin : c; y.Cin_real = (k == 0 || k == 3) ? cin_real : c;
            y.Cout = c; y.ntaps = k == 3 ? 1 : 9;
            y.offW = po; po += (long)c * y.Cin_real * y.ntaps;
            y.offG = po; po += c; y.offB = po; po += c;
            const long fe = (long)y.ntaps * y.Cin * c;
            y.fF = fo; fo += fe;
            const bool need_bwd = !(l == 0 && (k == 0 || k == 3));       // images are constants: no input gradient for block 1's c1 / cs
            y.fB = need_bwd ? fo : -1; if (need_bwd) fo += fe;
        }
        cin = c; cin_real = c; H /= 2; W /= 2;
    }
    n.g[nblk] = rn_geom(H, W);
    n.PSZ = (po + 63) / 64 * 64; n.FSZ = (fo + 127) / 128 * 128;
    n.F = channels[nblk - 1];
    return FUMI_OK;
}

struct RnPass {                                    // buffers of one pass over M images per episode
    int M;
    rbf16* u[RN_MAXBLK][4]; rbf16* a[RN_MAXBLK][2]; rbf16* out[RN_MAXBLK];
    rbf16* du[RN_MAXBLK][4]; rbf16* da[RN_MAXBLK][2]; rbf16* dout[RN_MAXBLK];
    float* coef[RN_MAXBLK][4];
    float* f; float* df; float* z; float* p; float* dz;
};
struct RnTan {                                     // tangent maps (support-sized, one inner step at a time)
    rbf16* ud[RN_MAXBLK][4]; rbf16* ad[RN_MAXBLK][2]; rbf16* outd[RN_MAXBLK];
    rbf16* dud[RN_MAXBLK][4]; rbf16* dad[RN_MAXBLK][2]; rbf16* doutd[RN_MAXBLK];
    float* fd; float* dfd; float* dzd;
};

size_t map_el(const RnNet& n, int M, int l) { return (size_t)n.B * M * n.g[l].Pp * n.C[l]; }
size_t out_el(const RnNet& n, int M, int l) { return (size_t)n.B * M * n.g[l + 1].Pp * n.C[l]; }
rbf16* ws_h(fumi_ws* ws, size_t n) { return (rbf16*)ws_f(ws, (n + 1) / 2); }

size_t pass_bytes(const RnNet& n, int M, bool bwd) {
    size_t b = 0;
    for (int l = 0; l < n.nblk; ++l) {
        b += (bwd ? 12 : 6) * ws_align(map_el(n, M, l) * 2) + (bwd ? 2 : 1) * ws_align(out_el(n, M, l) * 2);
        b += 4 * ws_align((size_t)n.B * RCF_N * n.C[l] * 4);
    }
    return b + 2 * ws_align((size_t)n.B * M * n.F * 4) + 3 * ws_align((size_t)n.B * M * n.N * 4);
}
void pass_carve(fumi_ws* ws, const RnNet& n, int M, bool bwd, RnPass& pb) {
    pb.M = M;
    for (int l = 0; l < n.nblk; ++l) {
        for (int k = 0; k < 4; ++k) pb.u[l][k] = ws_h(ws, map_el(n, M, l));
        for (int k = 0; k < 2; ++k) pb.a[l][k] = ws_h(ws, map_el(n, M, l));
        pb.out[l] = ws_h(ws, out_el(n, M, l));
        for (int k = 0; k < 4; ++k) pb.du[l][k] = bwd ? ws_h(ws, map_el(n, M, l)) : nullptr;
        for (int k = 0; k < 2; ++k) pb.da[l][k] = bwd ? ws_h(ws, map_el(n, M, l)) : nullptr;
        pb.dout[l] = bwd ? ws_h(ws, out_el(n, M, l)) : nullptr;
        for (int k = 0; k < 4; ++k) pb.coef[l][k] = ws_f(ws, (size_t)n.B * RCF_N * n.C[l]);
    }
    pb.f = ws_f(ws, (size_t)n.B * M * n.F); pb.df = ws_f(ws, (size_t)n.B * M * n.F);
    pb.z = ws_f(ws, (size_t)n.B * M * n.N); pb.p = ws_f(ws, (size_t)n.B * M * n.N); pb.dz = ws_f(ws, (size_t)n.B * M * n.N);
}
size_t tan_bytes(const RnNet& n, int M) {
    size_t b = 0;
    for (int l = 0; l < n.nblk; ++l) b += 12 * ws_align(map_el(n, M, l) * 2) + 2 * ws_align(out_el(n, M, l) * 2);
    return b + 2 * ws_align((size_t)n.B * M * n.F * 4) + ws_align((size_t)n.B * M * n.N * 4);
}
void tan_carve(fumi_ws* ws, const RnNet& n, int M, RnTan& tb) {
    for (int l = 0; l < n.nblk; ++l) {
        for (int k = 0; k < 4; ++k) { tb.ud[l][k] = ws_h(ws, map_el(n, M, l)); tb.dud[l][k] = ws_h(ws, map_el(n, M, l)); }
        for (int k = 0; k < 2; ++k) { tb.ad[l][k] = ws_h(ws, map_el(n, M, l)); tb.dad[l][k] = ws_h(ws, map_el(n, M, l)); }
        tb.outd[l] = ws_h(ws, out_el(n, M, l)); tb.doutd[l] = ws_h(ws, out_el(n, M, l));
    }
    tb.fd = ws_f(ws, (size_t)n.B * M * n.F); tb.dfd = ws_f(ws, (size_t)n.B * M * n.F); tb.dzd = ws_f(ws, (size_t)n.B * M * n.N);
}

struct RnScratch { float* cpart; float* rpart; float* wpart; float* rowl; float* c2; size_t cpart_n, rpart_n, wpart_n, rowl_n, c2_n; };

size_t scratch_sizes(const RnNet& n, int S, int Qn, RnScratch& sc) {
    size_t cp = 0, rp = 0, wp = 0;
    const int Ms[2] = {S, Qn};
    for (int mi = 0; mi < 2; ++mi)
        for (int l = 0; l < n.nblk; ++l) {
            const long npix = (long)Ms[mi] * n.g[l].Pp;
            const size_t c1 = (size_t)n.B * rn_conv_tiles(npix, n.g[l]) * 2 * n.C[l];
            cp = c1 > cp ? c1 : cp;
            RnMap m; m.B = n.B; m.M = Ms[mi]; m.C = n.C[l]; m.g = n.g[l];
            const size_t r1 = (size_t)n.B * rn_red_nt(m) * 5 * n.C[l];
            rp = r1 > rp ? r1 : rp;
            for (int k = 0; k < RN_NCONV; ++k) {
                const RnLayer& y = n.L[l][k];
                const int Ci32 = (y.Cin + 31) / 32 * 32;
                const size_t w1 = (size_t)n.B * rn_wgrad_nsplit(n.B, npix, y.Cin, y.Cout) * y.ntaps * y.Cout * Ci32;
                wp = w1 > wp ? w1 : wp;
            }
        }
    sc.cpart_n = cp; sc.rpart_n = rp; sc.wpart_n = wp; sc.rowl_n = 2 * (size_t)n.B * (S > Qn ? S : Qn);
    sc.c2_n = (cp > rp ? cp : rp) / 32 + 4096;                       // first-stage sums of the coefficient kernel (groups of 64 slabs)
    return ws_align(cp * 4) + ws_align(rp * 4) + ws_align(wp * 4) + ws_align(sc.rowl_n * 4) + ws_align(sc.c2_n * 4);
}

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define TRYP(phase, expr) do { ProfScope _ps(c.ws, c.st, phase); int _rc = (expr); if (_rc) return _rc; } while (0)

// `side` (the workspace's low-priority second stream, FUMI_RN_SIDE=0: off): the weight gradients are leaves of the backward chain
// (nothing in the pass reads them), MFMA-bound, and a quarter of the step, while a quarter of the chain itself is HBM-bound
// element-wise passes -- so they are forked off behind the map they read (ev[0]) and fill the matrix pipe under those passes;
// the pass joins (ev[1]) before it returns.  Their partial-sum scratch is touched by that stream only.  Measured: 568.8 -> 551.7 ms
// per 8-episode step (3 %: the convolutions and the weight gradients each hold 2 x 240 of a SIMD's 512 registers per lane, so the
// element-wise waves cannot sit beside them -- the gain is the tails of the launches filling up, not true co-residency).
struct RnCtx { fumi_ws* ws; hipStream_t st; RnNet n; RnScratch sc; hipStream_t side = nullptr; bool forked = false;
               hipEvent_t ev_fork = nullptr, ev_join = nullptr; };

static hipStream_t rn_side_stream(fumi_ws* ws) {
    static const int on = getenv("FUMI_RN_SIDE") ? atoi(getenv("FUMI_RN_SIDE")) : 1;
    if (ws->profiling) return nullptr;              // phase timing (bench.py's roofline step): one stream, every phase on its own
    return on ? ws->side : nullptr;
}
// a failed call may leave weight-gradient launches on the second stream: nothing else may touch the workspace before they are done
static void rn_abandon(fumi_ws* ws) {
    if (ws && ws->side) (void)hipStreamSynchronize(ws->side);
    for (int i = 0; ws && i < 3; ++i) if (ws->lanes[i]) (void)hipStreamSynchronize(ws->lanes[i]);
}
// main waits for everything forked so far
static int rn_join(RnCtx& c) {
    if (!c.forked) return FUMI_OK;
    HIP_TRY(hipEventRecord(c.ev_join, c.side));
    HIP_TRY(hipStreamWaitEvent(c.st, c.ev_join, 0));
    c.forked = false;
    return FUMI_OK;
}

// bf16 fragment copies of one parameter slot (every layer, forward and backward-data order)
int frags_of_slot(hipStream_t st, const RnNet& n, const float* params, rbf16* frags) {
    for (int l = 0; l < n.nblk; ++l)
        for (int k = 0; k < RN_NCONV; ++k) {
            const RnLayer& y = n.L[l][k];
            TRY(launch_rn_wprep(st, n.B, y.Cout, y.Cin, y.Cin_real, y.ntaps, params + y.offW, n.PSZ, frags + y.fF,
                                y.fB >= 0 ? frags + y.fB : nullptr, n.FSZ));
        }
    return FUMI_OK;
}

RnMap map_of(const RnNet& n, int M, int l) { RnMap m; m.B = n.B; m.M = M; m.C = n.C[l]; m.g = n.g[l]; return m; }

RnSrc src_of(const rbf16* in, long in_stride, const rbf16* frag, long fstride, int Cin, int ntaps) {
    RnSrc s; s.in = in; s.in_stride = in_stride; s.frag = frag; s.frag_stride = fstride; s.Cin = Cin; s.ntaps = ntaps; return s;
}

// one conv (1..4 sources) + its statistics + the coefficient table of its BN
int conv_bn(RnCtx& c, int M, int l, int nsrc, const RnSrc* src, rbf16* out, const rbf16* dot, int mode, float* coef,
            const float* g, const float* beta, long pstride) {
    const RnNet& n = c.n;
    RnConvArgs a; memset(&a, 0, sizeof(a));
    a.B = n.B; a.nsrc = nsrc; a.Cout = n.C[l]; a.npix = (long)M * n.g[l].Pp; a.g = n.g[l];
    for (int s = 0; s < 4; ++s) a.src[s] = src[s < nsrc ? s : 0];
    a.out = out; a.out_stride = (long)M * n.g[l].Pp * n.C[l];
    a.stats = c.sc.cpart; a.dot = dot; a.dot_stride = a.out_stride;
    int nt = 0;
    TRYP(FUMI_PH_RN_CONV, launch_rn_conv(c.st, a, &nt));
    if ((size_t)n.B * nt * 2 * n.C[l] > c.sc.cpart_n) return FUMI_ENOMEM;        // (cannot happen: scratch_sizes bounds the tiles)
    RnCoefArgs ca; memset(&ca, 0, sizeof(ca));
    ca.B = n.B; ca.C = n.C[l]; ca.mode = mode; ca.nt = nt; ca.K = 2; ca.k0 = 0; ca.k1 = 1; ca.k2 = 0;
    ca.n = (float)((double)M * n.g[l].H * n.g[l].W);
    ca.part = c.sc.cpart; ca.coef = coef;
    if (mode == RCM_FWD) { ca.g = g; ca.beta = beta; ca.pstride = pstride; }
    else { ca.gd = g; ca.betad = beta; ca.dstride = pstride; }
    TRYP(FUMI_PH_RN_EW, launch_rn_coef(c.st, ca, c.sc.c2));
    return FUMI_OK;
}

// input-gradient convolution (no statistics)
int conv_plain(RnCtx& c, long npix, const RnGeom& g, int Cout, int nsrc, const RnSrc* src, rbf16* out) {
    RnConvArgs a; memset(&a, 0, sizeof(a));
    a.B = c.n.B; a.nsrc = nsrc; a.Cout = Cout; a.npix = npix; a.g = g;
    for (int s = 0; s < 4; ++s) a.src[s] = src[s < nsrc ? s : 0];
    a.out = out; a.out_stride = npix * Cout;
    TRYP(FUMI_PH_RN_CONV, launch_rn_conv(c.st, a));
    return FUMI_OK;
}

int wgrad(RnCtx& c, int M, int l, const RnLayer& y, int npair, const rbf16* x0, const rbf16* dy0, const rbf16* x1, const rbf16* dy1,
          float* G) {
    const RnNet& n = c.n;
    RnWgradArgs a; memset(&a, 0, sizeof(a));
    a.B = n.B; a.npair = npair; a.Cin = y.Cin; a.Cout = y.Cout; a.ntaps = y.ntaps; a.npix = (long)M * n.g[l].Pp; a.g = n.g[l];
    a.nsplit = rn_wgrad_nsplit(n.B, a.npix, y.Cin, y.Cout);
    a.x[0] = x0; a.dy[0] = dy0; a.x[1] = x1; a.dy[1] = dy1;
    a.x_stride = a.npix * y.Cin; a.dy_stride = a.npix * y.Cout;
    a.part = c.sc.wpart;
    hipStream_t st = c.st;
    if (c.side) {                                   // everything this product reads has been launched on the main stream by now
        HIP_TRY(hipEventRecord(c.ev_fork, c.st));
        HIP_TRY(hipStreamWaitEvent(c.side, c.ev_fork, 0));
        st = c.side; c.forked = true;
    }
    { ProfScope _ps(c.ws, st, FUMI_PH_RN_WGRAD); TRY(launch_rn_wgrad(st, a)); }
    { ProfScope _ps(c.ws, st, FUMI_PH_RN_WGRAD);
      TRY(launch_rn_wgrad_reduce(st, n.B, a.nsplit, y.ntaps, y.Cout, y.Cin, y.Cin_real, c.sc.wpart, G + y.offW, n.PSZ)); }
    return FUMI_OK;
}

RnJoin join_of(const RnNet& n, int M, int l, const RnPass& pb, const RnTan* tb) {
    RnJoin j; memset(&j, 0, sizeof(j));
    j.m = map_of(n, M, l); j.gn = n.g[l + 1]; j.Ho = n.g[l].H / 2; j.Wo = n.g[l].W / 2;
    j.u3 = pb.u[l][2]; j.us = pb.u[l][3]; j.coef3 = pb.coef[l][2]; j.coefs = pb.coef[l][3];
    if (tb) { j.u3d = tb->ud[l][2]; j.usd = tb->ud[l][3]; }
    return j;
}

int forward_pass(RnCtx& c, int M, const rbf16* img16, const float* params, const rbf16* frags, RnPass& pb, const float* head,
                 const int64_t* y, float scale, float* logits, int64_t* preds, float* preds_f, float* loss_b, float* acc_b) {
    const RnNet& n = c.n;
    for (int l = 0; l < n.nblk; ++l) {
        const RnLayer* L = n.L[l];
        const rbf16* xin = l ? pb.out[l - 1] : img16;
        const long xs = (long)M * n.g[l].Pp * L[0].Cin, ms = (long)M * n.g[l].Pp * n.C[l];
        const RnMap m = map_of(n, M, l);
        RnSrc s[4];
        s[0] = src_of(xin, xs, frags + L[0].fF, n.FSZ, L[0].Cin, 9);
        TRY(conv_bn(c, M, l, 1, s, pb.u[l][0], nullptr, RCM_FWD, pb.coef[l][0], params + L[0].offG, params + L[0].offB, n.PSZ));
        TRYP(FUMI_PH_RN_EW, launch_rn_act(c.st, m, pb.u[l][0], nullptr, pb.coef[l][0], pb.a[l][0]));
        s[0] = src_of(pb.a[l][0], ms, frags + L[1].fF, n.FSZ, L[1].Cin, 9);
        TRY(conv_bn(c, M, l, 1, s, pb.u[l][1], nullptr, RCM_FWD, pb.coef[l][1], params + L[1].offG, params + L[1].offB, n.PSZ));
        TRYP(FUMI_PH_RN_EW, launch_rn_act(c.st, m, pb.u[l][1], nullptr, pb.coef[l][1], pb.a[l][1]));
        s[0] = src_of(pb.a[l][1], ms, frags + L[2].fF, n.FSZ, L[2].Cin, 9);
        TRY(conv_bn(c, M, l, 1, s, pb.u[l][2], nullptr, RCM_FWD, pb.coef[l][2], params + L[2].offG, params + L[2].offB, n.PSZ));
        s[0] = src_of(xin, xs, frags + L[3].fF, n.FSZ, L[3].Cin, 1);
        TRY(conv_bn(c, M, l, 1, s, pb.u[l][3], nullptr, RCM_FWD, pb.coef[l][3], params + L[3].offG, params + L[3].offB, n.PSZ));
        TRYP(FUMI_PH_RN_EW, launch_rn_join_fwd(c.st, join_of(n, M, l, pb, nullptr), pb.out[l], 0));
    }
    TRYP(FUMI_PH_RN_EW, launch_rn_avgpool(c.st, n.B * M, n.F, n.g[n.nblk], pb.out[n.nblk - 1], pb.f));
    if (!head) return FUMI_OK;
    HeadArgs h; memset(&h, 0, sizeof(h));
    h.B = n.B; h.M = M; h.N = n.N; h.F = n.F; h.scale = scale; h.f = pb.f; h.head = head; h.y = y;
    h.z = logits ? logits : pb.z; h.p = pb.p; h.dz = pb.dz; h.preds = preds; h.preds_f = preds_f; h.loss_b = loss_b; h.acc_b = acc_b;
    h.status = c.ws->status; h.row_loss = c.sc.rowl; h.row_hit = c.sc.rowl + (size_t)n.B * M;
    TRYP(FUMI_PH_RN_EW, launch_head_logits(c.st, h));
    return FUMI_OK;
}

// BWD coefficients of one BN from reduced partial sums (slices k0 / k1 of K) + its weight / bias gradients
int coef_bwd(RnCtx& c, int M, int l, int nt, int K, int k0, int k1, int k2, int mode, float* coef, const float* gd, float* dg, float* db) {
    const RnNet& n = c.n;
    RnCoefArgs ca; memset(&ca, 0, sizeof(ca));
    ca.B = n.B; ca.C = n.C[l]; ca.mode = mode; ca.nt = nt; ca.K = K; ca.k0 = k0; ca.k1 = k1; ca.k2 = k2;
    ca.n = (float)((double)M * n.g[l].H * n.g[l].W);
    ca.part = c.sc.rpart; ca.coef = coef; ca.gd = gd; ca.dstride = n.PSZ; ca.dg = dg; ca.dbeta = db; ca.gstride = n.PSZ;
    TRYP(FUMI_PH_RN_EW, launch_rn_coef(c.st, ca, c.sc.c2));
    return FUMI_OK;
}

// G [B][PSZ], dh [B][N][F+1]: gradient of the pass's loss (head == NULL: the feature adjoints are already in pb.df)
int backward_pass(RnCtx& c, int M, const rbf16* img16, const rbf16* frags, RnPass& pb, const float* head, float* G, float* dh) {
    const RnNet& n = c.n;
    if (head) {
        HeadGradArgs hg; memset(&hg, 0, sizeof(hg));
        hg.B = n.B; hg.M = M; hg.N = n.N; hg.F = n.F; hg.nsrc = 1; hg.dz[0] = pb.dz; hg.f[0] = pb.f; hg.head[0] = head;
        hg.dh = dh; hg.df = pb.df;
        TRYP(FUMI_PH_RN_EW, launch_head_grad(c.st, hg));
    }
    TRYP(FUMI_PH_RN_EW, launch_rn_avgpool_bwd(c.st, n.B * M, n.F, n.g[n.nblk], pb.df, pb.dout[n.nblk - 1]));
    for (int l = n.nblk - 1; l >= 0; --l) {
        const RnLayer* L = n.L[l];
        const rbf16* xin = l ? pb.out[l - 1] : img16;
        const long npix = (long)M * n.g[l].Pp, ms = npix * n.C[l];
        const RnMap m = map_of(n, M, l);
        const int nt = rn_red_nt(m);
        const RnJoin j = join_of(n, M, l, pb, nullptr);
        TRYP(FUMI_PH_RN_EW, launch_rn_join_reduce(c.st, j, pb.dout[l], nullptr, c.sc.rpart, 0));
        TRY(coef_bwd(c, M, l, nt, 3, 0, 1, 0, RCM_BWD, pb.coef[l][2], nullptr, G + L[2].offG, G + L[2].offB));
        TRY(coef_bwd(c, M, l, nt, 3, 0, 2, 0, RCM_BWD, pb.coef[l][3], nullptr, G + L[3].offG, G + L[3].offB));
        TRYP(FUMI_PH_RN_EW, launch_rn_join_apply(c.st, j, pb.dout[l], nullptr, pb.du[l][2], pb.du[l][3], 0));
        RnSrc s[4];
        for (int k = 2; k >= 1; --k) {                    // c3 then c2: weight gradient, input gradient, the BN below
            TRY(wgrad(c, M, l, L[k], 1, pb.a[l][k - 1], pb.du[l][k], nullptr, nullptr, G));
            s[0] = src_of(pb.du[l][k], ms, frags + L[k].fB, n.FSZ, n.C[l], 9);
            TRY(conv_plain(c, npix, n.g[l], n.C[l], 1, s, pb.da[l][k - 1]));
            TRYP(FUMI_PH_RN_EW, launch_rn_bwd_reduce(c.st, m, pb.u[l][k - 1], nullptr, pb.da[l][k - 1], nullptr, pb.coef[l][k - 1], c.sc.rpart, 0));
            TRY(coef_bwd(c, M, l, nt, 2, 0, 1, 0, RCM_BWD, pb.coef[l][k - 1], nullptr, G + L[k - 1].offG, G + L[k - 1].offB));
            TRYP(FUMI_PH_RN_EW, launch_rn_bwd_apply(c.st, m, pb.u[l][k - 1], nullptr, pb.da[l][k - 1], nullptr, pb.coef[l][k - 1], pb.du[l][k - 1], 0));
        }
        TRY(wgrad(c, M, l, L[0], 1, xin, pb.du[l][0], nullptr, nullptr, G));
        TRY(wgrad(c, M, l, L[3], 1, xin, pb.du[l][3], nullptr, nullptr, G));
        if (l) {
            s[0] = src_of(pb.du[l][0], ms, frags + L[0].fB, n.FSZ, n.C[l], 9);
            s[1] = src_of(pb.du[l][3], ms, frags + L[3].fB, n.FSZ, n.C[l], 1);
            TRY(conv_plain(c, npix, n.g[l], n.C[l - 1], 2, s, pb.dout[l - 1]));
        }
    }
    return rn_join(c);
}

// HV [B][PSZ], HVh [B][N][F+1] = Hessian of the support loss at the tape's parameters times (V, Vh)
int hvp_pass(RnCtx& c, int M, const rbf16* img16, const rbf16* frags, RnPass& pb, RnTan& tb, const float* head, const float* V,
             const rbf16* Vfrags, const float* Vh, float scale, float* HV, float* HVh) {
    const RnNet& n = c.n;
    // ---- tangent forward
    for (int l = 0; l < n.nblk; ++l) {
        const RnLayer* L = n.L[l];
        const rbf16* xin = l ? pb.out[l - 1] : img16;
        const rbf16* xind = l ? tb.outd[l - 1] : nullptr;
        const long xs = (long)M * n.g[l].Pp * L[0].Cin, ms = (long)M * n.g[l].Pp * n.C[l];
        const RnMap m = map_of(n, M, l);
        RnSrc s[4];
        s[0] = src_of(xin, xs, Vfrags + L[0].fF, n.FSZ, L[0].Cin, 9);
        if (xind) s[1] = src_of(xind, xs, frags + L[0].fF, n.FSZ, L[0].Cin, 9);
        TRY(conv_bn(c, M, l, xind ? 2 : 1, s, tb.ud[l][0], pb.u[l][0], RCM_TFWD, pb.coef[l][0], V + L[0].offG, V + L[0].offB, n.PSZ));
        TRYP(FUMI_PH_RN_EW, launch_rn_act(c.st, m, pb.u[l][0], tb.ud[l][0], pb.coef[l][0], tb.ad[l][0]));
        for (int k = 1; k <= 2; ++k) {
            s[0] = src_of(pb.a[l][k - 1], ms, Vfrags + L[k].fF, n.FSZ, n.C[l], 9);
            s[1] = src_of(tb.ad[l][k - 1], ms, frags + L[k].fF, n.FSZ, n.C[l], 9);
            TRY(conv_bn(c, M, l, 2, s, tb.ud[l][k], pb.u[l][k], RCM_TFWD, pb.coef[l][k], V + L[k].offG, V + L[k].offB, n.PSZ));
            if (k == 1) TRYP(FUMI_PH_RN_EW, launch_rn_act(c.st, m, pb.u[l][1], tb.ud[l][1], pb.coef[l][1], tb.ad[l][1]));
        }
        s[0] = src_of(xin, xs, Vfrags + L[3].fF, n.FSZ, L[3].Cin, 1);
        if (xind) s[1] = src_of(xind, xs, frags + L[3].fF, n.FSZ, L[3].Cin, 1);
        TRY(conv_bn(c, M, l, xind ? 2 : 1, s, tb.ud[l][3], pb.u[l][3], RCM_TFWD, pb.coef[l][3], V + L[3].offG, V + L[3].offB, n.PSZ));
        TRYP(FUMI_PH_RN_EW, launch_rn_join_fwd(c.st, join_of(n, M, l, pb, &tb), tb.outd[l], 1));
    }
    TRYP(FUMI_PH_RN_EW, launch_rn_avgpool(c.st, n.B * M, n.F, n.g[n.nblk], tb.outd[n.nblk - 1], tb.fd));
    HeadArgs h; memset(&h, 0, sizeof(h));
    h.B = n.B; h.M = M; h.N = n.N; h.F = n.F; h.scale = scale; h.f = pb.f; h.head = head; h.y = nullptr;
    h.fd = tb.fd; h.headd = Vh; h.p = pb.p; h.dz = tb.dzd;
    TRYP(FUMI_PH_RN_EW, launch_head_logits(c.st, h));
    // ---- tangent backward
    HeadGradArgs hg; memset(&hg, 0, sizeof(hg));
    hg.B = n.B; hg.M = M; hg.N = n.N; hg.F = n.F; hg.nsrc = 2;
    hg.dz[0] = tb.dzd; hg.f[0] = pb.f; hg.head[0] = head;
    hg.dz[1] = pb.dz; hg.f[1] = tb.fd; hg.head[1] = Vh;
    hg.dh = HVh; hg.df = tb.dfd;
    TRYP(FUMI_PH_RN_EW, launch_head_grad(c.st, hg));
    TRYP(FUMI_PH_RN_EW, launch_rn_avgpool_bwd(c.st, n.B * M, n.F, n.g[n.nblk], tb.dfd, tb.doutd[n.nblk - 1]));
    for (int l = n.nblk - 1; l >= 0; --l) {
        const RnLayer* L = n.L[l];
        const rbf16* xin = l ? pb.out[l - 1] : img16;
        const rbf16* xind = l ? tb.outd[l - 1] : nullptr;
        const long npix = (long)M * n.g[l].Pp, ms = npix * n.C[l];
        const RnMap m = map_of(n, M, l);
        const int nt = rn_red_nt(m);
        const RnJoin j = join_of(n, M, l, pb, &tb);
        TRYP(FUMI_PH_RN_EW, launch_rn_join_reduce(c.st, j, pb.dout[l], tb.doutd[l], c.sc.rpart, 1));
        TRY(coef_bwd(c, M, l, nt, 5, 0, 1, 2, RCM_TBWD, pb.coef[l][2], V + L[2].offG, HV + L[2].offG, HV + L[2].offB));
        TRY(coef_bwd(c, M, l, nt, 5, 0, 3, 4, RCM_TBWD, pb.coef[l][3], V + L[3].offG, HV + L[3].offG, HV + L[3].offB));
        TRYP(FUMI_PH_RN_EW, launch_rn_join_apply(c.st, j, pb.dout[l], tb.doutd[l], tb.dud[l][2], tb.dud[l][3], 1));
        RnSrc s[4];
        for (int k = 2; k >= 1; --k) {
            TRY(wgrad(c, M, l, L[k], 2, pb.a[l][k - 1], tb.dud[l][k], tb.ad[l][k - 1], pb.du[l][k], HV));
            s[0] = src_of(tb.dud[l][k], ms, frags + L[k].fB, n.FSZ, n.C[l], 9);
            s[1] = src_of(pb.du[l][k], ms, Vfrags + L[k].fB, n.FSZ, n.C[l], 9);
            TRY(conv_plain(c, npix, n.g[l], n.C[l], 2, s, tb.dad[l][k - 1]));
            TRYP(FUMI_PH_RN_EW, launch_rn_bwd_reduce(c.st, m, pb.u[l][k - 1], tb.ud[l][k - 1], pb.da[l][k - 1], tb.dad[l][k - 1],
                                                     pb.coef[l][k - 1], c.sc.rpart, 1));
            TRY(coef_bwd(c, M, l, nt, 3, 0, 1, 2, RCM_TBWD, pb.coef[l][k - 1], V + L[k - 1].offG, HV + L[k - 1].offG, HV + L[k - 1].offB));
            TRYP(FUMI_PH_RN_EW, launch_rn_bwd_apply(c.st, m, pb.u[l][k - 1], tb.ud[l][k - 1], pb.da[l][k - 1], tb.dad[l][k - 1],
                                                    pb.coef[l][k - 1], tb.dud[l][k - 1], 1));
        }
        TRY(wgrad(c, M, l, L[0], xind ? 2 : 1, xin, tb.dud[l][0], xind, pb.du[l][0], HV));
        TRY(wgrad(c, M, l, L[3], xind ? 2 : 1, xin, tb.dud[l][3], xind, pb.du[l][3], HV));
        if (l) {
            s[0] = src_of(tb.dud[l][0], ms, frags + L[0].fB, n.FSZ, n.C[l], 9);
            s[1] = src_of(pb.du[l][0], ms, Vfrags + L[0].fB, n.FSZ, n.C[l], 9);
            s[2] = src_of(tb.dud[l][3], ms, frags + L[3].fB, n.FSZ, n.C[l], 1);
            s[3] = src_of(pb.du[l][3], ms, Vfrags + L[3].fB, n.FSZ, n.C[l], 1);
            TRY(conv_plain(c, npix, n.g[l], n.C[l - 1], 4, s, tb.doutd[l - 1]));
        }
    }
    return rn_join(c);
}

size_t g_rn_budget = 0;                               // workspace budget in bytes for the episode chunking (0: default)

// ---- test hooks (fumi_hip_resnet12_set_option / fumi_hip_rn12_probe) -------------------------------------------------------------------
// probe mode: one lane, and when the meta-batch is a single chunk the buffer table of the step is kept so that tests can read EVERY
// stored intermediate (tests/test_resnet12_probe.py feeds each stage of oracle/resnet12_manual.py the engine's own upstream maps and
// compares that stage's output alone: bf16 decorrelation cannot accumulate).  In this mode the gradient of every inner step and the
// direction of the last Hessian-vector product are kept too (they are overwritten otherwise).  hvp_stop = k: the reverse sweep ends
// after inner step k (its tangent maps, HV and V stay in place for the probe); 0 = the whole sweep.
int g_rn_probe = 0, g_rn_hvp_stop = 0;
constexpr int RN_MAXTAPE = 16;
struct RnProbeTab {
    bool valid; fumi_ws* ws; char* base;
    RnNet n; int T, S, Qn, ntape, nslot, second;
    RnPass tape[RN_MAXTAPE]; RnPass query; RnTan tan;
    float* params; float* heads; float* G; float* dh; float* bar; float* barh; float* HV; float* HVh;
    float* Gsave; float* dhsave; float* Vsave; float* Vhsave; rbf16* img_s; rbf16* img_q;
};
RnProbeTab g_rn_tab;

}  // namespace

struct Rn12Problem {
    int B, N, S, Qn, Cimg, H, W, nblk, T; int channels[RN_MAXBLK];
    float alpha, grad_scale;
    int need_grad, second_order, chunk;
    const float* x_s; const int64_t* y_s; const float* x_q; const int64_t* y_q;
    const float* theta[12 * RN_MAXBLK];
    const float* head; float* head_bar;              // [B][N][F+1]
    float* logits_q; int64_t* preds_q; float* preds_f; float* loss_b; float* acc_b; float* stats;
    float* g_theta[12 * RN_MAXBLK];
};

// bytes of the workspace a chunk of `Bc` episodes needs
static size_t chunk_bytes(RnNet& n, int Bc, const Rn12Problem& p, RnScratch& sc) {
    n.B = Bc;
    const bool grad = p.need_grad != 0, second = grad && p.second_order && p.T > 0;
    const int ntape = second ? p.T : 1, nslot = second ? p.T + 1 : 2;
    const size_t hsz = (size_t)Bc * n.N * (n.F + 1);
    size_t b = scratch_sizes(n, p.S, p.Qn, sc);
    b += ws_align((size_t)Bc * p.S * n.g[0].Pp * 16 * 2) + ws_align((size_t)Bc * p.Qn * n.g[0].Pp * 16 * 2);      // prepared images
    b += (size_t)ntape * pass_bytes(n, p.S, true) + pass_bytes(n, p.Qn, grad);
    if (second) b += tan_bytes(n, p.S);
    b += (size_t)nslot * (ws_align((size_t)Bc * n.PSZ * 4) + ws_align((size_t)Bc * n.FSZ * 2) + ws_align(hsz * 4));
    b += 3 * ws_align((size_t)Bc * n.PSZ * 4) + 3 * ws_align(hsz * 4) + ws_align((size_t)Bc * n.FSZ * 2);        // G, bar, HV | dh, barh, HVh | Vfrags
    b += 2 * ws_align((size_t)n.PSZ * 4);                                                                           // gsum, gacc
    if (g_rn_probe) b += (size_t)(ntape + 1) * (ws_align((size_t)Bc * n.PSZ * 4) + ws_align(hsz * 4));             // Gsave, dhsave | Vsave, Vhsave
    return b + (1u << 16);
}

int run_rn12_episodes(fumi_ws* ws, hipStream_t st, const Rn12Problem& p) {
    // Two LANES: the meta-batch's chunks of episodes are independent until their meta-gradients are added, so odd chunks run on a
    // second stream (ws->lane, own half of the workspace) beside the even ones on the caller's.  Each lane's kernels alternate between
    // MFMA-bound (convolutions, weight gradients: 3/4 of the time) and HBM-bound (element-wise: 1/4) and leave partial last rounds
    // of workgroups; two lanes out of phase fill both (measured with two processes of 4 episodes each against one of 8: 15.9 vs
    // 14.1 episodes/s; in one process at 24 episodes 2 / 3 / 4 lanes: 14.46 / 14.81 / 14.24 -- two by default, FUMI_RN_LANES=n <= 4).
    // FUMI_RN_LANES=1 or phase timing: one lane (then the weight gradients fork onto ws->side instead).
    static const int lanes_env = getenv("FUMI_RN_LANES") ? atoi(getenv("FUMI_RN_LANES")) : 2;
    constexpr int MAXLANES = 4;
    RnCtx cx[MAXLANES];
    for (int i = 0; i < MAXLANES; ++i) { cx[i].ws = ws; cx[i].st = st; }
    RnCtx& c = cx[0];
    int rc = net_init(c.n, p.B, p.nblk, p.Cimg, p.N, p.H, p.W, p.channels);
    if (rc) return rc;
    RnNet& n = c.n;
    if (p.T < 0 || p.S < 1 || p.Qn < 1) return FUMI_EINVAL;
    const bool grad = p.need_grad != 0, second = grad && p.second_order && p.T > 0;
    if (second && p.T > RN_MAXTAPE) return FUMI_ENOTSUP;
    const int ntape = second ? p.T : 1, nslot = second ? p.T + 1 : 2;
    int lanes = (lanes_env >= 2 && !ws->profiling && !g_rn_probe && p.B >= 2 && ws->side) ? (lanes_env > MAXLANES ? MAXLANES : lanes_env) : 1;
    if (lanes > p.B) lanes = p.B;
    g_rn_tab.valid = false;
    // ---- chunk size: the largest number of episodes whose tapes (one per lane) fit the budget
    size_t budget = g_rn_budget;
    if (!budget) {
        const char* e = getenv("FUMI_RN12_BUDGET_GB");
        budget = (size_t)((e && atof(e) > 0 ? atof(e) : 200.0) * (double)(1ull << 30));
    }
    int Bc = p.chunk > 0 ? (p.chunk < p.B ? p.chunk : p.B) : (p.B + lanes - 1) / lanes;
    if (p.chunk <= 0) {
        // the largest chunk whose lanes fit (chunk_bytes is close to linear in Bc: start from the estimate, then step down)
        const size_t one = chunk_bytes(n, 1, p, c.sc);
        const long est = (long)(budget / ((size_t)lanes * (one ? one : 1))) + 1;
        if (Bc > est) Bc = (int)(est < 1 ? 1 : est);
        while (Bc > 1 && lanes * chunk_bytes(n, Bc, p, c.sc) > budget) --Bc;
    }
    if (Bc >= p.B) lanes = 1;                                             // a single chunk
    const size_t region = ws_align(chunk_bytes(n, Bc, p, c.sc));
    if ((rc = ws_reserve(ws, lanes * region))) return rc;
    // (a high-priority lane stream and GPU_MAX_HW_QUEUES=8 were tried: 1082 / 1068 vs 1063 ms per 16 episodes)
    for (int i = 1; i < lanes; ++i) if (!ws_lane_stream(ws, i)) { lanes = 1; break; }
    // (one lane: its weight gradients fork onto ws->side; with two lanes a stream of weight gradients per lane added nothing --
    // 530.5 vs 524.2 ms per 8-episode step -- and they stay in line)
    if (lanes == 1) { c.side = rn_side_stream(ws); c.ev_fork = ws->ev[0]; c.ev_join = ws->ev[1]; }
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(ws->ev[2], st));                           // the lanes start behind everything already on the caller's stream
        for (int i = 1; i < lanes; ++i) {
            cx[i].n = n; cx[i].st = ws->lanes[i - 1];
            HIP_TRY(hipStreamWaitEvent(ws->lanes[i - 1], ws->ev[2], 0));
        }
    }
    const size_t F1 = (size_t)n.N * (n.F + 1);
    float* gacc_lane[MAXLANES] = {nullptr, nullptr, nullptr, nullptr};
    fumi_ws* const ws_real = ws;
    // one chunk of episodes [b0, b0 + bc) on its lane's stream, in its lane's half of the workspace
    auto chunk_body = [&](int lane, int b0, int bc, bool first_of_lane, float** gacc_out) -> int {
        RnCtx& c = cx[lane];
        RnNet& n = c.n;
        const hipStream_t st = c.st;                                      // (shadows: everything of this chunk goes to its lane's stream)
        fumi_ws view = *ws_real;                                          // (a private bump pointer per chunk)
        fumi_ws* ws = &view;
        (void)chunk_bytes(n, bc, p, c.sc);                               // (sets n.B = bc and the scratch sizes of this chunk)
        ws->off = (size_t)lane * region;                                 // (ws: this thread's carving view of the workspace)
        float* gsum = ws_f(ws, (size_t)n.PSZ); float* gacc = ws_f(ws, (size_t)n.PSZ);   // first carve: same address in every chunk of a lane
        *gacc_out = gacc;
        c.sc.cpart = ws_f(ws, c.sc.cpart_n); c.sc.rpart = ws_f(ws, c.sc.rpart_n); c.sc.wpart = ws_f(ws, c.sc.wpart_n);
        c.sc.rowl = ws_f(ws, c.sc.rowl_n); c.sc.c2 = ws_f(ws, c.sc.c2_n);
        rbf16* img_s = ws_h(ws, (size_t)bc * p.S * n.g[0].Pp * 16); rbf16* img_q = ws_h(ws, (size_t)bc * p.Qn * n.g[0].Pp * 16);
        std::vector<RnPass> tape(ntape);
        for (int t = 0; t < ntape; ++t) pass_carve(ws, n, p.S, true, tape[t]);
        RnPass query; pass_carve(ws, n, p.Qn, grad, query);
        RnTan tan; memset(&tan, 0, sizeof(tan));
        if (second) tan_carve(ws, n, p.S, tan);
        const size_t hsz = (size_t)bc * F1;
        float* params = ws_f(ws, (size_t)nslot * bc * n.PSZ);
        rbf16* frags = ws_h(ws, (size_t)nslot * bc * n.FSZ);
        float* heads = ws_f(ws, (size_t)nslot * hsz);
        float* G = ws_f(ws, (size_t)bc * n.PSZ); float* bar = ws_f(ws, (size_t)bc * n.PSZ); float* HV = ws_f(ws, (size_t)bc * n.PSZ);
        float* dh = ws_f(ws, hsz); float* barh = ws_f(ws, hsz); float* HVh = ws_f(ws, hsz);
        rbf16* Vfrags = ws_h(ws, (size_t)bc * n.FSZ);
        float* Gsave = nullptr; float* dhsave = nullptr; float* Vsave = nullptr; float* Vhsave = nullptr;
        if (g_rn_probe) {
            Gsave = ws_f(ws, (size_t)ntape * bc * n.PSZ); dhsave = ws_f(ws, (size_t)ntape * hsz);
            Vsave = ws_f(ws, (size_t)bc * n.PSZ); Vhsave = ws_f(ws, hsz);
        }
        if (ws->off > (size_t)(lane + 1) * region || ws->off > ws->cap) return FUMI_ENOMEM;
        if (g_rn_probe && bc == p.B) {                                   // the whole meta-batch in one chunk: keep the buffer table
            RnProbeTab& pt = g_rn_tab;
            pt.ws = ws_real; pt.base = ws_real->base; pt.n = n; pt.T = p.T; pt.S = p.S; pt.Qn = p.Qn; pt.ntape = ntape; pt.nslot = nslot;
            pt.second = second ? 1 : 0;
            for (int t = 0; t < ntape; ++t) pt.tape[t] = tape[t];
            pt.query = query; pt.tan = tan; pt.params = params; pt.heads = heads; pt.G = G; pt.dh = dh; pt.bar = bar; pt.barh = barh;
            pt.HV = HV; pt.HVh = HVh; pt.Gsave = Gsave; pt.dhsave = dhsave; pt.Vsave = Vsave; pt.Vhsave = Vhsave;
            pt.img_s = img_s; pt.img_q = img_q;
            pt.valid = true;
        }
        auto P = [&](int s) { return params + (size_t)s * bc * n.PSZ; };
        auto Fr = [&](int s) { return frags + (size_t)s * bc * n.FSZ; };
        auto Hd = [&](int s) { return heads + (size_t)s * hsz; };
        const float* x_s = p.x_s + (size_t)b0 * p.S * p.Cimg * p.H * p.W; const float* x_q = p.x_q + (size_t)b0 * p.Qn * p.Cimg * p.H * p.W;
        const int64_t* y_s = p.y_s + (size_t)b0 * p.S; const int64_t* y_q = p.y_q + (size_t)b0 * p.Qn;
        // ---- images -> bf16 padded channels-last (once per chunk), slot 0 = the meta-parameters broadcast to every episode
        TRY(launch_rn_img_prep(st, (long)bc * p.S, p.Cimg, n.g[0], x_s, img_s));
        TRY(launch_rn_img_prep(st, (long)bc * p.Qn, p.Cimg, n.g[0], x_q, img_q));
        HIP_TRY(hipMemsetAsync(G, 0, (size_t)bc * n.PSZ * 4, st));          // (the slab's padding words stay 0 in every vector derived from G)
        HIP_TRY(hipMemsetAsync(P(0), 0, (size_t)bc * n.PSZ * 4, st));
        HIP_TRY(hipMemsetAsync(HV, 0, (size_t)bc * n.PSZ * 4, st));
        for (int l = 0; l < n.nblk; ++l)
            for (int k = 0; k < RN_NCONV; ++k) {
                const RnLayer& y = n.L[l][k];
                const float* const* th = p.theta + 12 * l + 3 * k;
                TRY(launch_broadcast(st, bc, (long)y.Cout * y.Cin_real * y.ntaps, th[0], P(0) + y.offW, n.PSZ));
                TRY(launch_broadcast(st, bc, y.Cout, th[1], P(0) + y.offG, n.PSZ));
                TRY(launch_broadcast(st, bc, y.Cout, th[2], P(0) + y.offB, n.PSZ));
            }
        TRY(frags_of_slot(st, n, P(0), Fr(0)));
        HIP_TRY(hipMemcpyAsync(Hd(0), p.head + (size_t)b0 * F1, hsz * 4, hipMemcpyDeviceToDevice, st));
        // ---- inner loop on the support set
        int cur = 0;
        for (int t = 0; t < p.T; ++t) {
            RnPass& pb = tape[second ? t : 0];
            const int nxt = second ? t + 1 : cur ^ 1;
            TRY(forward_pass(c, p.S, img_s, P(cur), Fr(cur), pb, Hd(cur), y_s, 1.f / p.S, nullptr, nullptr, nullptr, nullptr, nullptr));
            TRY(backward_pass(c, p.S, img_s, Fr(cur), pb, Hd(cur), G, dh));
            if (Gsave) {
                const int ts = second ? t : 0;
                HIP_TRY(hipMemcpyAsync(Gsave + (size_t)ts * bc * n.PSZ, G, (size_t)bc * n.PSZ * 4, hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemcpyAsync(dhsave + (size_t)ts * hsz, dh, hsz * 4, hipMemcpyDeviceToDevice, st));
            }
            TRYP(FUMI_PH_RN_EW, launch_axpy(st, (long)bc * n.PSZ, P(cur), -p.alpha, G, P(nxt)));
            TRYP(FUMI_PH_RN_EW, launch_axpy(st, (long)hsz, Hd(cur), -p.alpha, dh, Hd(nxt)));
            TRY(frags_of_slot(st, n, P(nxt), Fr(nxt)));
            cur = nxt;
        }
        // ---- query pass with the adapted parameters
        TRY(forward_pass(c, p.Qn, img_q, P(cur), Fr(cur), query, Hd(cur), y_q, 1.f / p.Qn, p.logits_q + (size_t)b0 * p.Qn * n.N,
                         p.preds_q + (size_t)b0 * p.Qn, p.preds_f ? p.preds_f + (size_t)b0 * p.Qn : nullptr, p.loss_b + b0, p.acc_b + b0));
        if (!grad) return FUMI_OK;
        HIP_TRY(hipMemsetAsync(bar, 0, (size_t)bc * n.PSZ * 4, st));
        TRY(backward_pass(c, p.Qn, img_q, Fr(cur), query, Hd(cur), bar, barh));
        if (second) {
            const int t_stop = g_rn_probe && g_rn_hvp_stop > 0 && g_rn_hvp_stop < p.T ? g_rn_hvp_stop : 0;
            for (int t = p.T - 1; t >= t_stop; --t) {
                if (Vsave) {
                    HIP_TRY(hipMemcpyAsync(Vsave, bar, (size_t)bc * n.PSZ * 4, hipMemcpyDeviceToDevice, st));
                    HIP_TRY(hipMemcpyAsync(Vhsave, barh, hsz * 4, hipMemcpyDeviceToDevice, st));
                }
                TRY(frags_of_slot(st, n, bar, Vfrags));
                TRY(hvp_pass(c, p.S, img_s, Fr(t), tape[t], tan, Hd(t), bar, Vfrags, barh, 1.f / p.S, HV, HVh));
                TRYP(FUMI_PH_RN_EW, launch_axpy(st, (long)bc * n.PSZ, bar, -p.alpha, HV, bar));
                TRYP(FUMI_PH_RN_EW, launch_axpy(st, (long)hsz, barh, -p.alpha, HVh, barh));
            }
        }
        // ---- meta-gradient of the chunk: scaled sum over its episodes, added to the running sum
        TRY(launch_reduce_batched(st, 1, bc, n.PSZ, bar, p.grad_scale, gsum, 0));
        if (first_of_lane) HIP_TRY(hipMemcpyAsync(gacc, gsum, (size_t)n.PSZ * 4, hipMemcpyDeviceToDevice, st));
        else TRY(launch_axpy(st, n.PSZ, gacc, 1.f, gsum, gacc));
        HIP_TRY(hipMemcpyAsync(p.head_bar + (size_t)b0 * F1, barh, hsz * 4, hipMemcpyDeviceToDevice, st));
        return FUMI_OK;
    };
    // chunk k belongs to lane k mod lanes; the chunks are enqueued in order, alternating lanes (a second host thread enqueuing lane 1 on
    // its own changed nothing: 1071 vs 1073 ms per 16 episodes)
    {
        bool first[MAXLANES] = {true, true, true, true};
        int ck = 0;
        for (int b0 = 0; b0 < p.B; b0 += Bc, ++ck) {
            const int bc = p.B - b0 < Bc ? p.B - b0 : Bc, lane = ck % lanes;
            TRY(chunk_body(lane, b0, bc, first[lane], &gacc_lane[lane]));
            first[lane] = false;
        }
    }
    for (int i = 1; i < lanes; ++i) {                                     // the caller's stream waits for the other lanes
        HIP_TRY(hipEventRecord(ws->lane_ev[i - 1], ws->lanes[i - 1]));
        HIP_TRY(hipStreamWaitEvent(st, ws->lane_ev[i - 1], 0));
    }
    float* gacc = gacc_lane[0];
    for (int i = 1; i < lanes; ++i) if (grad && gacc_lane[i]) TRY(launch_axpy(st, n.PSZ, gacc, 1.f, gacc_lane[i], gacc));
    if (p.stats) {
        ReduceSegs sg; sg.n = 0; sg.scale = p.grad_scale;
        sg.add(p.loss_b, p.B, 1, 1, p.stats); sg.add(p.acc_b, p.B, 1, 1, p.stats + 1);
        TRY(launch_reduce_multi(st, sg));
    }
    if (!grad) return FUMI_OK;
    for (int l = 0; l < n.nblk; ++l)
        for (int k = 0; k < RN_NCONV; ++k) {
            const RnLayer& y = n.L[l][k];
            float* const* g = p.g_theta + 12 * l + 3 * k;
            HIP_TRY(hipMemcpyAsync(g[0], gacc + y.offW, (size_t)y.Cout * y.Cin_real * y.ntaps * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(g[1], gacc + y.offG, (size_t)y.Cout * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(g[2], gacc + y.offB, (size_t)y.Cout * 4, hipMemcpyDeviceToDevice, st));
        }
    return FUMI_OK;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
static int fill_problem(Rn12Problem& p, int B, int N, int S, int Qn, int Cimg, int H, int W, int nblk, const int* channels, int T,
                        float alpha, int need_grad, int second_order, float grad_scale, int chunk, const float* x_s, const int64_t* y_s,
                        const float* x_q, const int64_t* y_q, const float* const* theta, float* logits_q, int64_t* preds_q,
                        float* preds_f, float* loss_b, float* acc_b, float* stats, float* const* g_theta) {
    memset(&p, 0, sizeof(p));
    if (!x_s || !y_s || !x_q || !y_q || !theta || !logits_q || !preds_q || !loss_b || !acc_b || !channels) return FUMI_EINVAL;
    if (nblk < 1 || nblk > RN_MAXBLK || (need_grad && !g_theta)) return FUMI_EINVAL;
    p.B = B; p.N = N; p.S = S; p.Qn = Qn; p.Cimg = Cimg; p.H = H; p.W = W; p.nblk = nblk; p.T = T; p.alpha = alpha;
    for (int l = 0; l < nblk; ++l) p.channels[l] = channels[l];
    p.grad_scale = grad_scale; p.need_grad = need_grad ? 1 : 0; p.second_order = second_order ? 1 : 0; p.chunk = chunk;
    p.x_s = x_s; p.y_s = y_s; p.x_q = x_q; p.y_q = y_q;
    for (int i = 0; i < 12 * nblk; ++i) {
        if (!theta[i] || (need_grad && !g_theta[i])) return FUMI_EINVAL;
        p.theta[i] = theta[i]; p.g_theta[i] = need_grad ? g_theta[i] : nullptr;
    }
    p.logits_q = logits_q; p.preds_q = preds_q; p.preds_f = preds_f; p.loss_b = loss_b; p.acc_b = acc_b; p.stats = stats;
    return FUMI_OK;
}

// small side allocation that outlives the chunk loop's workspace rewinds: the heads and their adjoints of the WHOLE meta-batch and
// the hypernetwork's activations (owned by the workspace, grown on demand)
namespace {
int side_reserve(fumi_ws* ws, size_t floats, float** out) {
    if (ws->side_cap < floats) {
        HIP_TRY(hipDeviceSynchronize());
        if (ws->side_buf) (void)hipFree(ws->side_buf);
        ws->side_buf = nullptr; ws->side_cap = 0;
        HIP_TRY(hipMalloc((void**)&ws->side_buf, floats * 4 + 1024));
        ws->side_cap = floats;
    }
    *out = ws->side_buf;
    return FUMI_OK;
}
size_t al64(size_t n) { return (n + 63) / 64 * 64; }
}  // namespace

extern "C" {

// Test hooks.  key 0: probe mode (one lane; a single-chunk step keeps its buffer table for fumi_hip_rn12_probe and the per-step
// gradients / the last Hessian-vector product's direction).  key 1: the reverse sweep stops after inner step `value` (probe mode).
int fumi_hip_resnet12_set_option(int key, int value) {
    if (key == 0) { g_rn_probe = value ? 1 : 0; if (!value) g_rn_tab.valid = false; return FUMI_OK; }
    if (key == 1) { if (value < 0) return FUMI_EINVAL; g_rn_hvp_stop = value; return FUMI_OK; }
    return FUMI_EINVAL;
}

// Copies one stored intermediate of the LAST single-chunk ResNet-12 step run in probe mode out of the workspace.
//   pass 0 .. T-1: the tape of support step `pass` (first order / T = 0: only the last step's tape exists, pass 0);  pass T: the
//   query pass;  pass T + 1: the tangent maps of the last Hessian-vector product (inner step hvp_stop);  pass -1: parameter space.
//   kind (pass >= 0): 0 u[block][idx 0..3: c1, c2, c3, shortcut], 1 a[block][idx 0..1], 2 out[block], 3 du[block][idx], 4 da[block][idx],
//     5 dout[block] (bf16 padded channels-last maps [B][M (H+2)(W+2)][C]); 6 coef[block][idx] fp32 [B][RCF_N][C]; 7 f, 8 df fp32 [B][M][F];
//     9 z, 10 p, 11 dz fp32 [B][M][N] (9 of the query pass: the caller's logits hold it).  The tangent pass has kinds 0-5, 7, 8, 11.
//   kind (pass -1): 0 parameter slot idx [B][PSZ], 1 head slot idx [B][N][F+1], 2 G / 3 dh of inner step idx, 4 bar, 5 bar_h, 6 HV,
//     7 HV_h, 8 V / 9 V_h (direction of the last Hessian-vector product), 10 / 11 prepared support / query images bf16 [B][M Pp][16].
// *n_bytes = its size, *is_bf16 = element type; at most max_bytes are copied.
int fumi_hip_rn12_probe(fumi_ws_t* ws, fumi_stream_t stream, int pass, int kind, int block, int idx, void* out, size_t max_bytes,
                        size_t* n_bytes, int* is_bf16) {
    if (!ws || !out || !n_bytes || !is_bf16 || !g_rn_tab.valid) return FUMI_EINVAL;
    const RnProbeTab& pt = g_rn_tab;
    if (pt.ws != ws || pt.base != ws->base) return FUMI_EINVAL;           // another workspace's step, or the slab has moved since
    const RnNet& n = pt.n;
    const void* src = nullptr; size_t bytes = 0; int bf = 0;
    const size_t hsz = (size_t)n.B * n.N * (n.F + 1), psz = (size_t)n.B * n.PSZ;
    if (pass == -1) {
        switch (kind) {
            case 0: if (idx < 0 || idx >= pt.nslot) return FUMI_EINVAL; src = pt.params + (size_t)idx * psz; bytes = psz * 4; break;
            case 1: if (idx < 0 || idx >= pt.nslot) return FUMI_EINVAL; src = pt.heads + (size_t)idx * hsz; bytes = hsz * 4; break;
            case 2: if (idx < 0 || idx >= pt.ntape) return FUMI_EINVAL; src = pt.Gsave + (size_t)idx * psz; bytes = psz * 4; break;
            case 3: if (idx < 0 || idx >= pt.ntape) return FUMI_EINVAL; src = pt.dhsave + (size_t)idx * hsz; bytes = hsz * 4; break;
            case 4: src = pt.bar; bytes = psz * 4; break;
            case 5: src = pt.barh; bytes = hsz * 4; break;
            case 6: src = pt.HV; bytes = psz * 4; break;
            case 7: src = pt.HVh; bytes = hsz * 4; break;
            case 8: src = pt.Vsave; bytes = psz * 4; break;
            case 9: src = pt.Vhsave; bytes = hsz * 4; break;
            case 10: src = pt.img_s; bytes = (size_t)n.B * pt.S * n.g[0].Pp * 16 * 2; bf = 1; break;
            case 11: src = pt.img_q; bytes = (size_t)n.B * pt.Qn * n.g[0].Pp * 16 * 2; bf = 1; break;
            default: return FUMI_EINVAL;
        }
    } else if (pass == pt.T + 1) {
        if (!pt.second || block < 0 || block >= n.nblk) return FUMI_EINVAL;
        const int M = pt.S;
        const RnTan& tb = pt.tan;
        switch (kind) {
            case 0: if (idx < 0 || idx > 3) return FUMI_EINVAL; src = tb.ud[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 1: if (idx < 0 || idx > 1) return FUMI_EINVAL; src = tb.ad[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 2: src = tb.outd[block]; bytes = out_el(n, M, block) * 2; bf = 1; break;
            case 3: if (idx < 0 || idx > 3) return FUMI_EINVAL; src = tb.dud[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 4: if (idx < 0 || idx > 1) return FUMI_EINVAL; src = tb.dad[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 5: src = tb.doutd[block]; bytes = out_el(n, M, block) * 2; bf = 1; break;
            case 7: src = tb.fd; bytes = (size_t)n.B * M * n.F * 4; break;
            case 8: src = tb.dfd; bytes = (size_t)n.B * M * n.F * 4; break;
            case 11: src = tb.dzd; bytes = (size_t)n.B * M * n.N * 4; break;
            default: return FUMI_EINVAL;
        }
    } else {
        if (pass < 0 || pass > pt.T || block < 0 || block >= n.nblk) return FUMI_EINVAL;
        if (pass < pt.T && pass >= pt.ntape) return FUMI_EINVAL;
        const RnPass& pb = pass == pt.T ? pt.query : pt.tape[pass];
        const int M = pb.M;
        switch (kind) {
            case 0: if (idx < 0 || idx > 3) return FUMI_EINVAL; src = pb.u[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 1: if (idx < 0 || idx > 1) return FUMI_EINVAL; src = pb.a[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 2: src = pb.out[block]; bytes = out_el(n, M, block) * 2; bf = 1; break;
            case 3: if (idx < 0 || idx > 3) return FUMI_EINVAL; src = pb.du[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 4: if (idx < 0 || idx > 1) return FUMI_EINVAL; src = pb.da[block][idx]; bytes = map_el(n, M, block) * 2; bf = 1; break;
            case 5: src = pb.dout[block]; bytes = out_el(n, M, block) * 2; bf = 1; break;
            case 6: if (idx < 0 || idx > 3) return FUMI_EINVAL; src = pb.coef[block][idx]; bytes = (size_t)n.B * RCF_N * n.C[block] * 4; break;
            case 7: src = pb.f; bytes = (size_t)n.B * M * n.F * 4; break;
            case 8: src = pb.df; bytes = (size_t)n.B * M * n.F * 4; break;
            case 9: src = pb.z; bytes = (size_t)n.B * M * n.N * 4; break;
            case 10: src = pb.p; bytes = (size_t)n.B * M * n.N * 4; break;
            case 11: src = pb.dz; bytes = (size_t)n.B * M * n.N * 4; break;
            default: return FUMI_EINVAL;
        }
    }
    if (!src) return FUMI_ENOTSUP;                                        // (a forward-only pass has no gradient maps)
    *n_bytes = bytes; *is_bf16 = bf;
    const size_t k = bytes < max_bytes ? bytes : max_bytes;
    if (k) HIP_TRY(hipMemcpyAsync(out, src, k, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FUMI_OK;
}

int fumi_hip_resnet12_set_budget(double gigabytes) { g_rn_budget = gigabytes > 0 ? (size_t)(gigabytes * (double)(1ull << 30)) : 0; return FUMI_OK; }

int fumi_hip_maml_resnet12_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk, const int* channels,
        int T, float alpha, int first_order, int need_grad, float grad_scale, int chunk,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* const* params,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_params) {
    if (!ws || !params || !channels) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    Rn12Problem p;
    int rc = fill_problem(p, B, N, S, Qn, Cin, H, W, nblk, channels, T, alpha, need_grad, !first_order, grad_scale, chunk, x_s, y_s, x_q,
                          y_q, params, logits_q, preds_q, preds_q_f32, loss_b, acc_b, stats, g_params);
    if (rc) return rc;
    const float* Wf = params[12 * nblk]; const float* bf = params[12 * nblk + 1];
    if (!Wf || !bf || (need_grad && (!g_params[12 * nblk] || !g_params[12 * nblk + 1]))) return FUMI_EINVAL;
    const int F = channels[nblk - 1];
    const size_t hsz = (size_t)B * N * (F + 1);
    float* side;
    if ((rc = side_reserve(ws, 2 * al64(hsz), &side))) return rc;
    float* h = side; float* hbar = side + al64(hsz);
    if ((rc = launch_broadcast_head(st, B, N, F, Wf, bf, h))) return rc;            // every episode starts from lin_final (maml.py:24-31)
    p.head = h; p.head_bar = hbar;
    if ((rc = run_rn12_episodes(ws, st, p))) { rn_abandon(ws); return rc; }
    if (!need_grad) return FUMI_OK;
    return launch_split_head_grad(st, B, N, F, hbar, grad_scale, g_params[12 * nblk], g_params[12 * nblk + 1]);
}

int fumi_hip_fumi_resnet12_step(fumi_ws_t* ws, fumi_stream_t stream,
        int B, int N, int S, int Qn, int Cin, int H, int W, int nblk, const int* channels, int Dt, int Ht,
        int T, float alpha, int tanh_head, int need_grad, float grad_scale, int chunk,
        const float* x_s, const int64_t* y_s, const float* x_q, const int64_t* y_q,
        const float* cls_text, const float* text_s,
        const float* const* theta, const float* const* phi,
        float* logits_q, int64_t* preds_q, float* preds_q_f32, float* loss_b, float* acc_b, float* stats,
        float* const* g_theta, float* const* g_phi) {
    if (!ws || !theta || !phi || !channels || (!cls_text && !text_s) || Dt < 1 || Ht < 1) return FUMI_EINVAL;
    if (need_grad && !g_phi) return FUMI_EINVAL;
    for (int i = 0; i < 4; ++i) if (!phi[i] || (need_grad && !g_phi[i])) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    Rn12Problem p;
    int rc = fill_problem(p, B, N, S, Qn, Cin, H, W, nblk, channels, T, alpha, need_grad, 1, grad_scale, chunk, x_s, y_s, x_q, y_q, theta,
                          logits_q, preds_q, preds_q_f32, loss_b, acc_b, stats, g_theta);          // fumi.py:176: second order always
    if (rc) return rc;
    const int F = channels[nblk - 1];
    const int R = B * N, H1 = F + 1;
    // hypernetwork rows are (episode, class) pairs: Linear(Dt, Ht) . ReLU . Linear(Ht, F+1) [. Tanh]  (fumi.py:70-86,104-113)
    float* side;
    if ((rc = side_reserve(ws, al64((size_t)R * Dt) + 2 * al64((size_t)R * Ht) + 3 * al64((size_t)R * H1), &side))) return rc;
    float* c = side; float* u = c + al64((size_t)R * Dt); float* ub = u + al64((size_t)R * Ht);
    float* h = ub + al64((size_t)R * Ht); float* hbar = h + al64((size_t)R * H1); float* hpb = hbar + al64((size_t)R * H1);
    const float* ctext = cls_text;
    if (!ctext) {                                                        // first support row of each class (fumi.py:207-210)
        if ((rc = launch_class_text_select(st, B, N, S, Dt, text_s, y_s, c, ws->status))) return rc;
        ctext = c;
    }
    GemmArgs g = gemm_args(R, Ht, Dt, ctext, Dt, phi[0], Dt, u, Ht);
    g.bias = phi[1]; g.act = 1;
    if ((rc = launch_gemm(st, g, 0, 0))) return rc;
    g = gemm_args(R, H1, Ht, u, Ht, phi[2], Ht, h, H1);
    g.bias = phi[3]; g.act = tanh_head ? 2 : 0;
    if ((rc = launch_gemm(st, g, 0, 0))) return rc;
    p.head = h; p.head_bar = hbar;
    if ((rc = run_rn12_episodes(ws, st, p))) { rn_abandon(ws); return rc; }
    if (!need_grad) return FUMI_OK;
    const float* hp = hbar;
    if (tanh_head) { if ((rc = launch_tanh_bwd(st, (long)R * H1, h, hbar, hpb))) return rc; hp = hpb; }
    g = gemm_args(H1, Ht, R, hp, H1, u, Ht, g_phi[2], Ht);               // gA1 = hp^T u
    g.alpha = grad_scale;
    if ((rc = launch_gemm(st, g, 1, 1))) return rc;
    if ((rc = launch_colsum(st, hp, R, H1, H1, grad_scale, g_phi[3]))) return rc;
    g = gemm_args(R, Ht, H1, hp, H1, phi[2], Ht, ub, Ht);                // ubar = (hp A1) * relu'(u)
    g.mask = u;
    if ((rc = launch_gemm(st, g, 0, 1))) return rc;
    g = gemm_args(Ht, Dt, R, ub, Ht, ctext, Dt, g_phi[0], Dt);           // gA0 = ubar^T c
    g.alpha = grad_scale;
    if ((rc = launch_gemm(st, g, 1, 1))) return rc;
    if (float* tg = ws->text_grad) {                                     // armed by fumi_hip_want_text_grad: scale * ubar A0  [R,Dt]
        ws->text_grad = nullptr;
        g = gemm_args(R, Dt, Ht, ub, Ht, phi[0], Dt, tg, Dt);
        g.alpha = grad_scale;
        if ((rc = launch_gemm(st, g, 0, 1))) return rc;
    }
    return launch_colsum(st, ub, R, Ht, Ht, grad_scale, g_phi[1]);
}

// Forward only: feats [G*M, F] = ResNet12(x [G, M, Cin, H, W]) with the batch statistics of every group of M images taken separately.
int fumi_hip_resnet12_features(fumi_ws_t* ws, fumi_stream_t stream, int G, int M, int Cin, int H, int W, int nblk, const int* channels,
        const float* x, const float* const* theta, float* feats) {
    if (!ws || !x || !theta || !feats || !channels || G < 1 || M < 1) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    RnCtx c; c.ws = ws; c.st = st;
    int rc = net_init(c.n, G, nblk, Cin, 1, H, W, channels);
    if (rc) return rc;
    const RnNet& n = c.n;
    for (int i = 0; i < 12 * nblk; ++i) if (!theta[i]) return FUMI_EINVAL;
    size_t bytes = scratch_sizes(n, M, M, c.sc) + pass_bytes(n, M, false) + ws_align((size_t)G * M * n.g[0].Pp * 16 * 2) +
                   ws_align((size_t)G * n.PSZ * 4) + ws_align((size_t)G * n.FSZ * 2);
    if ((rc = ws_reserve(ws, bytes))) return rc;
    g_rn_tab.valid = false;
    c.sc.cpart = ws_f(ws, c.sc.cpart_n); c.sc.rpart = ws_f(ws, c.sc.rpart_n); c.sc.wpart = ws_f(ws, c.sc.wpart_n); c.sc.rowl = ws_f(ws, c.sc.rowl_n); c.sc.c2 = ws_f(ws, c.sc.c2_n);
    RnPass pb; pass_carve(ws, n, M, false, pb);
    rbf16* img = ws_h(ws, (size_t)G * M * n.g[0].Pp * 16);
    float* params = ws_f(ws, (size_t)G * n.PSZ); rbf16* frags = ws_h(ws, (size_t)G * n.FSZ);
    TRY(launch_rn_img_prep(st, (long)G * M, Cin, n.g[0], x, img));
    for (int l = 0; l < n.nblk; ++l)
        for (int k = 0; k < RN_NCONV; ++k) {
            const RnLayer& y = n.L[l][k];
            const float* const* th = theta + 12 * l + 3 * k;
            TRY(launch_broadcast(st, G, (long)y.Cout * y.Cin_real * y.ntaps, th[0], params + y.offW, n.PSZ));
            TRY(launch_broadcast(st, G, y.Cout, th[1], params + y.offG, n.PSZ));
            TRY(launch_broadcast(st, G, y.Cout, th[2], params + y.offB, n.PSZ));
        }
    TRY(frags_of_slot(st, n, params, frags));
    TRY(forward_pass(c, M, img, params, frags, pb, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr));
    HIP_TRY(hipMemcpyAsync(feats, pb.f, (size_t)G * M * n.F * 4, hipMemcpyDeviceToDevice, st));
    return FUMI_OK;
}

// ---- unit ops on raw bf16 maps (parity tests of the matrix kernels) ------------------------------------------------------------------
// y [B][M*(H+2)(W+2)][Cout] bf16 = conv_{ntaps}(x [B][M*(H+2)(W+2)][Cin] bf16 padded channels-last, Wt [B][Cout][Cin][k][k] fp32);
// transpose != 0: the input-gradient product (x has Cout channels, y has Cin).  stats (optional) [B][2][Cy]: sum, sum of squares.
int fumi_hip_rn12_conv(fumi_ws_t* ws, fumi_stream_t stream, int B, int M, int H, int W, int Cin, int Cout, int ntaps, int transpose,
        const void* x, const float* Wt, void* y, float* stats) {
    if (!ws || !x || !Wt || !y || B < 1 || M < 1 || (ntaps != 9 && ntaps != 1)) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    const RnGeom g = rn_geom(H, W);
    const long npix = (long)M * g.Pp;
    const int Cy = transpose ? Cin : Cout, Cx = transpose ? Cout : Cin;
    const long fe = (long)ntaps * Cin * Cout;
    const int tiles = rn_conv_tiles(npix, g);
    int rc = ws_reserve(ws, 2 * ws_align((size_t)B * fe * 2) + ws_align((size_t)B * tiles * 2 * Cy * 4));
    if (rc) return rc;
    rbf16* ff = ws_h(ws, (size_t)B * fe); rbf16* fb = ws_h(ws, (size_t)B * fe); float* part = ws_f(ws, (size_t)B * tiles * 2 * Cy);
    TRY(launch_rn_wprep(st, B, Cout, Cin, Cin, ntaps, Wt, (long)Cout * Cin * ntaps, ff, transpose ? fb : nullptr, fe));
    RnConvArgs a; memset(&a, 0, sizeof(a));
    a.B = B; a.nsrc = 1; a.Cout = Cy; a.npix = npix; a.g = g;
    a.src[0].in = (const rbf16*)x; a.src[0].in_stride = npix * Cx; a.src[0].frag = transpose ? fb : ff; a.src[0].frag_stride = fe;
    a.src[0].Cin = Cx; a.src[0].ntaps = ntaps;
    a.src[1] = a.src[2] = a.src[3] = a.src[0];
    a.out = (rbf16*)y; a.out_stride = npix * Cy; a.stats = stats ? part : nullptr;
    HIP_TRY(hipMemsetAsync(y, 0, (size_t)B * npix * Cy * 2, st));      // the kernel writes interior pixels only (nothing reads the border of a conv output)
    int nt = 0;
    TRY(launch_rn_conv(st, a, &nt));
    if (stats) TRY(launch_reduce_batched(st, B, nt, 2L * Cy, part, 1.f, stats, 2L * Cy));
    return FUMI_OK;
}

// dW [B][Cout][Cin][k][k] fp32 = sum_p dy[p][co] x[p + off][ci]   (x, dy: raw bf16 maps as above)
int fumi_hip_rn12_wgrad(fumi_ws_t* ws, fumi_stream_t stream, int B, int M, int H, int W, int Cin, int Cout, int ntaps,
        const void* x, const void* dy, float* dW) {
    if (!ws || !x || !dy || !dW || B < 1 || M < 1 || (ntaps != 9 && ntaps != 1)) return FUMI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(ws->device));
    const RnGeom g = rn_geom(H, W);
    RnWgradArgs a; memset(&a, 0, sizeof(a));
    a.B = B; a.npair = 1; a.Cin = Cin; a.Cout = Cout; a.ntaps = ntaps; a.npix = (long)M * g.Pp; a.g = g;
    a.nsplit = rn_wgrad_nsplit(B, a.npix, Cin, Cout);
    const int Ci32 = (Cin + 31) / 32 * 32;
    int rc = ws_reserve(ws, ws_align((size_t)B * a.nsplit * ntaps * Cout * Ci32 * 4));
    if (rc) return rc;
    a.part = ws_f(ws, (size_t)B * a.nsplit * ntaps * Cout * Ci32);
    a.x[0] = (const rbf16*)x; a.dy[0] = (const rbf16*)dy; a.x_stride = a.npix * Cin; a.dy_stride = a.npix * Cout;
    TRY(launch_rn_wgrad(st, a));
    return launch_rn_wgrad_reduce(st, B, a.nsplit, ntaps, Cout, Cin, Cin, a.part, dW, (long)Cout * Cin * ntaps);
}

}  // extern "C"
