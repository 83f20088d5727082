"""Teacher-forced stage checks of the bf16 ResNet-12 meta-step (BASELINE.json configs[4]) -- TEST INFRASTRUCTURE ONLY.

A chain of bf16 roundings decorrelates: a value that rounds the other way in the engine than in the oracle perturbs everything
downstream, so a whole-step comparison at depth 4 / several inner steps can only carry a loose bound.  Here
every STAGE of the sweep -- each convolution, BN + LeakyReLU pass, residual join, their backward and tangent forms, the head, every
weight / BN-parameter gradient, every parameter-space update -- is checked on its own: ``oracle/resnet12_manual.py``'s formula for
the stage is fed the ENGINE's stored upstream maps (``fumi_hip_rn12_probe``) and only that stage's output is compared.  Nothing
accumulates: what is left is the stage's own output rounding (one bf16 ulp where fp32 summation order crosses a rounding boundary),
so every map must agree to 2^-7 of its maximum and 2e-3 in relative L2 (measured ~1e-4), every fp32 sum to 2e-4.  A wrong operand,
sign, coefficient or buffer in any stage of any block fails its own check by O(1).

Decisions taken at fp32 noise level are excluded where they are compared, not where they are used: elements whose LeakyReLU
argument is within 1e-5 of zero and pooling windows whose two largest entries tie within 1e-5 (the engine and the float64 formula
may legitimately decide them differently; the forward value does not move, a routed gradient does).

The algebra is oracle/resnet12_manual.py (equal to autograd at 1e-9 in float64, tests/test_resnet12_manual.py); the seam is
fumi/models/fumi.py:89-100, the loop fumi.py:146-192 / maml.py:156-191.
"""
import torch
import torch.nn.functional as F

from oracle import resnet12_manual as M
from oracle.resnet12_ref import SLOPE

RCF = dict(MU=0, R=1, A=2, C0=3, D1=4, D2=5, TB=6, TC=7, M1=8, M2=9, K0=10, DD1=11, E12=12)
RCF_N = 13
TIE = 1e-5
rnd = M.bf16_round


class Layout:
    """Offsets of the per-episode parameter slab -- mirrors net_init of fumi_amd/csrc/rn12.hip."""

    def __init__(self, channels, Cimg, H):
        po, cin, self.off, self.shape, self.H, self.channels = 0, Cimg, {}, {}, [], list(channels)
        for l, c in enumerate(channels):
            self.H.append(H)
            for k in range(4):
                ci = cin if k in (0, 3) else c
                ks = 1 if k == 3 else 3
                nW = c * ci * ks * ks
                self.off[(l, k)] = (po, po + nW, po + nW + c)
                self.shape[(l, k)] = (c, ci, ks)
                po += nW + 2 * c
            cin = c
            H //= 2
        self.H.append(H)
        self.PSZ = (po + 63) // 64 * 64
        self.nblk = len(channels)

    def W(self, vec, l, k):
        c, ci, ks = self.shape[(l, k)]
        o = self.off[(l, k)][0]
        return vec[o:o + c * ci * ks * ks].reshape(c, ci, ks, ks)

    def G(self, vec, l, k):
        o = self.off[(l, k)][1]
        return vec[o:o + self.shape[(l, k)][0]]

    def Bt(self, vec, l, k):
        o = self.off[(l, k)][2]
        return vec[o:o + self.shape[(l, k)][0]]

    def theta(self, vec):
        out = []
        for l in range(self.nblk):
            for k in range(4):
                out += [self.W(vec, l, k), self.G(vec, l, k), self.Bt(vec, l, k)]
        return out


class Checker:
    def __init__(self):
        self.rows, self.bad = [], []

    def maps(self, name, got, ref, excl=None):
        d = got - ref
        if excl is not None:
            d = d.masked_fill(excl, 0.0)
        mx = float(ref.abs().max())
        e_max = float(d.abs().max()) / max(mx, 1e-300)
        e_l2 = float(d.norm()) / max(float(ref.norm()), 1e-300)
        ok = (mx > 0) and e_max <= 2.0 ** -7 and e_l2 <= 2e-3
        self.rows.append((name, e_max, e_l2, ok))
        if not ok:
            self.bad.append(f"{name}: max {e_max:.2e} l2 {e_l2:.2e} (|ref|max {mx:.2e})")

    def vec(self, name, got, ref, tol=2e-4, scale=0.0, slack=None):
        """`scale`: magnitude of the terms the value is a difference of, where it can cancel to (nearly) nothing.  `slack` (per element,
        absolute): what the decisions taken at noise level (pooling arg-max, LeakyReLU sign: the `tie` masks) can move the value by --
        a sum over pixels has no element to set aside, so the bound of the tied terms is granted instead."""
        mx = max(float(ref.abs().max()), scale)
        d = (got - ref).abs()
        if slack is not None:
            d = (d - slack.reshape(d.shape)).clamp_min(0.0)
        e = float(d.max()) / max(mx, 1e-300)
        ok = (mx > 0) and e <= tol
        self.rows.append((name, e, e, ok))
        if not ok:
            self.bad.append(f"{name}: max {e:.2e} > {tol:.0e} (|ref|max {mx:.2e})")

    def assert_ok(self):
        assert not self.bad, f"{len(self.bad)} of {len(self.rows)} stage checks failed:\n" + "\n".join(self.bad[:40])


class Engine:
    """Reads the stored intermediates of the last probe-mode step."""

    def __init__(self, hip, ws, dev, lay, B, N, S, Qn, T, Cimg):
        self.hip, self.ws, self.dev, self.lay = hip, ws, dev, lay
        self.B, self.N, self.S, self.Qn, self.T, self.Cimg = B, N, S, Qn, T, Cimg
        self.cache = {}

    def raw(self, pas, kind, l=0, idx=0):
        key = (pas, kind, l, idx)
        if key not in self.cache:
            self.cache[key] = self.hip.rn12_probe(self.ws, self.dev, pas, kind, l, idx).cpu()
        return self.cache[key]

    def M_of(self, pas):
        return self.Qn if pas == self.T else self.S

    def map(self, pas, kind, l, idx=0, pooled=False):
        """[B, M, C, H, W] float64 interior of a padded channels-last bf16 map"""
        Mi, H, C = self.M_of(pas), self.lay.H[l + 1 if pooled else l], self.lay.channels[l]
        t = self.raw(pas, kind, l, idx).double().reshape(self.B, Mi, H + 2, H + 2, C)
        return t[:, :, 1:-1, 1:-1].permute(0, 1, 4, 2, 3).contiguous()

    def border(self, pas, kind, l, idx=0, pooled=False):
        Mi, H, C = self.M_of(pas), self.lay.H[l + 1 if pooled else l], self.lay.channels[l]
        t = self.raw(pas, kind, l, idx).double().reshape(self.B, Mi, H + 2, H + 2, C)
        return float(t[:, :, 0].abs().max() + t[:, :, -1].abs().max() + t[:, :, :, 0].abs().max() + t[:, :, :, -1].abs().max())

    def img(self, query):
        Mi, H = (self.Qn if query else self.S), self.lay.H[0]
        t = self.raw(-1, 11 if query else 10).double().reshape(self.B, Mi, H + 2, H + 2, 16)
        return t[:, :, 1:-1, 1:-1, :self.Cimg].permute(0, 1, 4, 2, 3).contiguous()

    def coef(self, pas, l, k):
        return self.raw(pas, 6, l, k).double().reshape(self.B, RCF_N, self.lay.channels[l])

    def f32(self, pas, kind, shape, idx=0):
        return self.raw(pas, kind, 0, idx).double().reshape(shape)


def _cc(v):
    return v.view(1, -1, 1, 1)


def fwd_block(E, chk, tag, pas, b, l, th, xin):
    """Forward stages of block l on the engine's maps of pass `pas`, episode b -> the block's tape (built from ENGINE values)."""
    lay = E.lay
    W = [rnd(lay.W(th, l, k)) for k in range(4)]
    g = [lay.G(th, l, k) for k in range(4)]
    be = [lay.Bt(th, l, k) for k in range(4)]
    u = [E.map(pas, 0, l, k)[b] for k in range(4)]
    a = [E.map(pas, 1, l, k)[b] for k in range(2)]
    out = E.map(pas, 2, l, pooled=True)[b]
    srcs = [xin, a[0], a[1], xin]
    bn, v = [], []
    for k in range(4):
        chk.maps(f"{tag} u{k}", u[k], rnd(M.conv(srcs[k], W[k])))
        vk, tk = M.bn_fwd(u[k], g[k], be[k])
        cf = E.coef(pas, l, k)[b]
        mu = u[k].mean((0, 2, 3))
        chk.vec(f"{tag} coef{k}.mu", cf[RCF["MU"]], mu, 2e-4, scale=float(u[k].abs().max()))
        chk.vec(f"{tag} coef{k}.r", cf[RCF["R"]], tk["r"].reshape(-1))
        chk.vec(f"{tag} coef{k}.A", cf[RCF["A"]], (tk["g"] * tk["r"]).reshape(-1))
        bn.append(tk); v.append(vk)
        if k < 2:
            chk.maps(f"{tag} a{k}", a[k], rnd(F.leaky_relu(vk, SLOPE)))
    s = v[2] + v[3]
    ls = F.leaky_relu(s, SLOPE)
    lw = M._windows(ls)
    mx = lw.max(-1)[0]
    first = ((lw == mx[..., None]).to(torch.int8).cumsum(-1) == 1) & (lw == mx[..., None])
    arg = first.to(torch.int8).argmax(-1)
    chk.maps(f"{tag} out", out, rnd(mx))
    top2 = lw.topk(2, -1)[0]
    thr = TIE * float(ls.abs().max())
    tie_w = ((top2[..., 0] - top2[..., 1]) < thr) | (mx.abs() < thr)                      # window decided at noise level
    H = s.shape[-1]
    tie_s = M._unwindows(tie_w[..., None].expand(*tie_w.shape, 4).double(), H, H) > 0
    tie = [vk.abs() < TIE * float(vk.abs().max()) for vk in v[:2]]
    return dict(x=xin, W=W, bn=bn, a=a, m=[M._lmask(v[0]), M._lmask(v[1])], ms=M._lmask(s), arg=arg, shape=s.shape, u=u, out=out,
                tie=tie, tie_s=tie_s, tie_w=tie_w)


def bwd_block(E, chk, tag, pas, b, l, tp, Gv):
    """Backward stages of block l: engine's dout[l] in, du / da / dout[l-1] and the 12 parameter gradients out."""
    lay = E.lay
    W, a, x = tp["W"], tp["a"], tp["x"]
    do = E.map(pas, 5, l, pooled=True)[b]
    du = [E.map(pas, 3, l, k)[b] for k in range(4)]
    da = [E.map(pas, 4, l, k)[b] for k in range(2)]
    ds = M._scatter(do, tp)
    tied = do.abs() * tp["tie_w"]                                      # windows whose arg-max / sign was decided at noise level
    for k, nm in ((2, "3"), (3, "s")):
        d_o, dg, db = M.bn_bwd(ds, tp["bn"][k])
        chk.maps(f"{tag} du{nm}", du[k], rnd(d_o), excl=tp["tie_s"])
        xh_w = M._windows(tp["bn"][k]["xh"]).abs().max(-1)[0]
        chk.vec(f"{tag} dg{nm}", lay.G(Gv, l, k), dg, slack=2.0 * (tied * xh_w).sum((0, 2, 3)))
        chk.vec(f"{tag} db{nm}", lay.Bt(Gv, l, k), db, slack=tied.sum((0, 2, 3)))
    chk.vec(f"{tag} dW3", lay.W(Gv, l, 2), M.conv_bwd_weight(a[1], du[2], 3))
    for k in (1, 0):                                                   # BN2 / BN1 behind c3 / c2
        chk.maps(f"{tag} da{k}", da[k], rnd(M.conv_bwd_data(du[k + 1], W[k + 1])))
        d_o, dg, db = M.bn_bwd(da[k] * tp["m"][k], tp["bn"][k])
        chk.maps(f"{tag} du{k}", du[k], rnd(d_o), excl=tp["tie"][k])
        tied_k = da[k].abs() * tp["tie"][k]                             # pixels whose LeakyReLU sign was decided at noise level
        chk.vec(f"{tag} dg{k}", lay.G(Gv, l, k), dg, slack=(tied_k * tp["bn"][k]["xh"].abs()).sum((0, 2, 3)))
        chk.vec(f"{tag} db{k}", lay.Bt(Gv, l, k), db, slack=tied_k.sum((0, 2, 3)))
        chk.vec(f"{tag} dW{k}", lay.W(Gv, l, k), M.conv_bwd_weight(a[k - 1] if k else x, du[k], 3))
    chk.vec(f"{tag} dWs", lay.W(Gv, l, 3), M.conv_bwd_weight(x, du[3], 1))
    if l:
        dx = E.map(pas, 5, l - 1, pooled=True)[b]
        chk.maps(f"{tag} dout[{l - 1}]", dx, rnd(M.conv_bwd_data(du[0], W[0]) + M.conv_bwd_data(du[3], W[3])))
    tp.update(du=du, da=da, do=do)
    for k in (0, 2, 3):                                                # zero borders the next products rely on
        assert E.border(pas, 3, l, k) == 0.0, f"{tag} du{k} border"


def check_pass(E, chk, tag, pas, b, th, head, y, scale, Gv, dh, z_ext=None):
    """One forward + backward pass (support step `pas` < T or the query pass) of episode b, stage by stage."""
    lay, Mi = E.lay, E.M_of(pas)
    x = E.img(pas == E.T)[b]
    tapes = []
    for l in range(lay.nblk):
        tp = fwd_block(E, chk, f"{tag} b{l}", pas, b, l, th, x)
        tapes.append(tp)
        x = tp["out"]
        for k in range(2):
            assert E.border(pas, 1, l, k) == 0.0
        assert E.border(pas, 2, l, pooled=True) == 0.0
    Fd, N = lay.channels[-1], E.N
    f = E.f32(pas, 7, (E.B, Mi, Fd))[b]
    chk.vec(f"{tag} f", f, x.mean((2, 3)), 1e-5)
    z_o = f @ head[:, :-1].t() + head[:, -1]
    z = z_ext if z_ext is not None else E.f32(pas, 9, (E.B, Mi, N))[b]
    chk.vec(f"{tag} z", z, z_o, 1e-5)
    p = torch.softmax(z, -1)
    chk.vec(f"{tag} p", E.f32(pas, 10, (E.B, Mi, N))[b], p, 1e-5)
    dz_o = (p - F.one_hot(y, N).double()) * scale
    dz = E.f32(pas, 11, (E.B, Mi, N))[b]
    chk.vec(f"{tag} dz", dz, dz_o, 2e-5)            # (fp32 soft-max with the fast exponential: 1.2e-5 of max |dz| was seen at T = 5)
    if Gv is None:
        return dict(blocks=tapes, f=f, h=head, p=p, dz=dz)
    chk.vec(f"{tag} dh", dh, torch.cat([dz.t() @ f, dz.sum(0)[:, None]], 1), 1e-5)
    df = E.f32(pas, 8, (E.B, Mi, Fd))[b]
    chk.vec(f"{tag} df", df, dz @ head[:, :-1], 1e-5)
    Ho = lay.H[-1]
    do_last = E.map(pas, 5, lay.nblk - 1, pooled=True)[b]
    chk.maps(f"{tag} dout[last]", do_last, rnd((df / (Ho * Ho))[:, :, None, None].expand(Mi, Fd, Ho, Ho)))
    for l in reversed(range(lay.nblk)):
        bwd_block(E, chk, f"{tag} b{l}", pas, b, l, tapes[l], Gv)
    return dict(blocks=tapes, f=f, h=head, p=p, dz=dz)


def check_hvp(E, chk, tag, b, tape, Vv, Vh, scale, HVv, HVh):
    """Tangent forward + tangent backward over the (engine-built) tape of one support step, direction (Vv, Vh)."""
    lay, T, S = E.lay, E.T, E.S
    TP = T + 1
    xd = None
    for l, tp in enumerate(tape["blocks"]):
        tg = f"{tag} b{l}"
        W, x, a = tp["W"], tp["x"], tp["a"]
        Wd = [rnd(lay.W(Vv, l, k)) for k in range(4)]
        gd = [lay.G(Vv, l, k) for k in range(4)]
        bd = [lay.Bt(Vv, l, k) for k in range(4)]
        ud = [E.map(TP, 0, l, k)[b] for k in range(4)]
        ad = [E.map(TP, 1, l, k)[b] for k in range(2)]
        outd = E.map(TP, 2, l, pooled=True)[b]

        def two(xa, Wa_d, xa_d, Wa):
            yv = M.conv(xa, Wa_d)
            return yv if xa_d is None else yv + M.conv(xa_d, Wa)
        chk.maps(f"{tg} u0'", ud[0], rnd(two(x, Wd[0], xd, W[0])))
        chk.maps(f"{tg} a0'", ad[0], rnd(tp["m"][0] * M.bn_tan_fwd(ud[0], gd[0], bd[0], tp["bn"][0])), excl=tp["tie"][0])
        chk.maps(f"{tg} u1'", ud[1], rnd(two(a[0], Wd[1], ad[0], W[1])))
        chk.maps(f"{tg} a1'", ad[1], rnd(tp["m"][1] * M.bn_tan_fwd(ud[1], gd[1], bd[1], tp["bn"][1])), excl=tp["tie"][1])
        chk.maps(f"{tg} u2'", ud[2], rnd(two(a[1], Wd[2], ad[1], W[2])))
        v3d = M.bn_tan_fwd(ud[2], gd[2], bd[2], tp["bn"][2])
        chk.maps(f"{tg} u3'", ud[3], rnd(two(x, Wd[3], xd, W[3])))
        vsd = M.bn_tan_fwd(ud[3], gd[3], bd[3], tp["bn"][3])
        chk.maps(f"{tg} out'", outd, rnd(M._gather(v3d + vsd, tp)), excl=tp["tie_w"])
        tp.update(xd=xd, Wd=Wd, ad=ad, ud=ud, outd=outd)
        xd = outd
    Fd, N = lay.channels[-1], E.N
    f, h, p, dz = tape["f"], tape["h"], tape["p"], tape["dz"]
    fd = E.f32(TP, 7, (E.B, S, Fd))[b]
    chk.vec(f"{tag} f'", fd, xd.mean((2, 3)), 1e-5)
    zd = fd @ h[:, :-1].t() + f @ Vh[:, :-1].t() + Vh[:, -1]
    dzd_o = p * (zd - (p * zd).sum(-1, keepdim=True)) * scale
    dzd = E.f32(TP, 11, (E.B, S, N))[b]
    chk.vec(f"{tag} dz'", dzd, dzd_o, 2e-5, scale=float(zd.abs().max()) * scale)      # p (z' - <p, z'>): cancels when z' is flat
    chk.vec(f"{tag} HV_h", HVh, torch.cat([dzd.t() @ f + dz.t() @ fd, dzd.sum(0)[:, None]], 1), 2e-5)
    dfd = E.f32(TP, 8, (E.B, S, Fd))[b]
    chk.vec(f"{tag} df'", dfd, dzd @ h[:, :-1] + dz @ Vh[:, :-1], 2e-5)
    Ho = lay.H[-1]
    chk.maps(f"{tag} dout'[last]", E.map(TP, 5, lay.nblk - 1, pooled=True)[b],
             rnd((dfd / (Ho * Ho))[:, :, None, None].expand(S, Fd, Ho, Ho)))
    for l in reversed(range(lay.nblk)):
        tp = tape["blocks"][l]
        tg = f"{tag} b{l}"
        W, Wd, x, xd, a, ad, du = tp["W"], tp["Wd"], tp["x"], tp["xd"], tp["a"], tp["ad"], tp["du"]
        dod = E.map(TP, 5, l, pooled=True)[b]
        dud = [E.map(TP, 3, l, k)[b] for k in range(4)]
        dad = [E.map(TP, 4, l, k)[b] for k in range(2)]
        dsd = M._scatter(dod, tp)
        for k, nm in ((2, "3"), (3, "s")):
            d_o, dgd, dbd = M.bn_tan_bwd(dsd, tp["bn"][k])
            chk.maps(f"{tg} du{nm}'", dud[k], rnd(d_o), excl=tp["tie_s"])
            chk.vec(f"{tg} dg{nm}'", lay.G(HVv, l, k), dgd)
            chk.vec(f"{tg} db{nm}'", lay.Bt(HVv, l, k), dbd)
        chk.vec(f"{tg} dW3'", lay.W(HVv, l, 2), M.conv_bwd_weight(a[1], dud[2], 3) + M.conv_bwd_weight(ad[1], du[2], 3))
        for k in (1, 0):
            chk.maps(f"{tg} da{k}'", dad[k], rnd(M.conv_bwd_data(dud[k + 1], W[k + 1]) + M.conv_bwd_data(du[k + 1], Wd[k + 1])))
            d_o, dgd, dbd = M.bn_tan_bwd(dad[k] * tp["m"][k], tp["bn"][k])
            chk.maps(f"{tg} du{k}'", dud[k], rnd(d_o), excl=tp["tie"][k])
            chk.vec(f"{tg} dg{k}'", lay.G(HVv, l, k), dgd)
            chk.vec(f"{tg} db{k}'", lay.Bt(HVv, l, k), dbd)
            if k:
                chk.vec(f"{tg} dW1'", lay.W(HVv, l, 1), M.conv_bwd_weight(a[0], dud[1], 3) + M.conv_bwd_weight(ad[0], du[1], 3))
        dW0 = M.conv_bwd_weight(x, dud[0], 3)
        dWs = M.conv_bwd_weight(x, dud[3], 1)
        if xd is not None:
            dW0 = dW0 + M.conv_bwd_weight(xd, du[0], 3)
            dWs = dWs + M.conv_bwd_weight(xd, du[3], 1)
        chk.vec(f"{tg} dW0'", lay.W(HVv, l, 0), dW0)
        chk.vec(f"{tg} dWs'", lay.W(HVv, l, 3), dWs)
        if l:
            ref = (M.conv_bwd_data(dud[0], W[0]) + M.conv_bwd_data(du[0], Wd[0]) + M.conv_bwd_data(dud[3], W[3])
                   + M.conv_bwd_data(du[3], Wd[3]))
            chk.maps(f"{tg} dout'[{l - 1}]", E.map(TP, 5, l - 1, pooled=True)[b], rnd(ref))


def check_step(hip, ws, dev, run, ep, theta, head0, channels, T, alpha, hvp_steps, episodes=None, logits=None):
    """Runs `run()` (a second-order MAML / FuMI ResNet-12 step on `ep`) once per entry of `hvp_steps` in probe mode with the reverse
    sweep stopped after that inner step and checks every stage.  head0 [B, N, F+1] float64: the heads the episodes start from.
    Returns (checker, per-episode bar_0 [B, PSZ], bar_h [B, N, F+1] of the run with hvp_stop == 0 or None)."""
    B, S = ep["x_s"].shape[:2]
    Qn, Cimg, H = ep["x_q"].shape[1], ep["x_s"].shape[2], ep["x_s"].shape[3]
    N = head0.shape[1]
    lay = Layout(channels, Cimg, H)
    Fd = channels[-1]
    chk = Checker()
    episodes = range(B) if episodes is None else episodes
    final = None
    prev_bar = None
    hip.resnet12_set_option(0, 1)
    try:
        for ri, stop in enumerate(hvp_steps):
            hip.resnet12_set_option(1, stop)
            out = run()
            assert ws.read_status() == 0
            E = Engine(hip, ws, dev, lay, B, N, S, Qn, T, Cimg)
            P = [E.f32(-1, 0, (B, lay.PSZ), idx=t) for t in range(T + 1)]
            Hd = [E.f32(-1, 1, (B, N, Fd + 1), idx=t) for t in range(T + 1)]
            Gs = [E.f32(-1, 2, (B, lay.PSZ), idx=t) for t in range(T)]
            dhs = [E.f32(-1, 3, (B, N, Fd + 1), idx=t) for t in range(T)]
            bar, barh = E.f32(-1, 4, (B, lay.PSZ)), E.f32(-1, 5, (B, N, Fd + 1))
            HV, HVh = E.f32(-1, 6, (B, lay.PSZ)), E.f32(-1, 7, (B, N, Fd + 1))
            V, Vh = E.f32(-1, 8, (B, lay.PSZ)), E.f32(-1, 9, (B, N, Fd + 1))
            zq = out["logits"].cpu().double()
            th0 = torch.cat([t.reshape(-1).double() for t in theta])
            for b in episodes:
                tg = f"r{ri} e{b}"
                chk.vec(f"{tg} slot0", P[0][b][:th0.numel()], th0, 1e-7)
                chk.vec(f"{tg} head0", Hd[0][b], head0[b], 1e-6)
                tapes = {}
                full = ri == 0                                          # the passes themselves are identical in every run
                for t in range(T):
                    if full or t == stop:
                        tapes[t] = check_pass(E, chk, f"{tg} t{t}", t, b, P[t][b], Hd[t][b], ep["y_s"][b], 1.0 / S, Gs[t][b], dhs[t][b])
                    chk.vec(f"{tg} slot{t + 1}", P[t + 1][b], P[t][b] - alpha * Gs[t][b], 1e-6)
                    chk.vec(f"{tg} head{t + 1}", Hd[t + 1][b], Hd[t][b] - alpha * dhs[t][b], 1e-6)
                if stop == T - 1:
                    # the direction of the first Hessian-vector product IS the query pass's gradient
                    check_pass(E, chk, f"{tg} q", T, b, P[T][b], Hd[T][b], ep["y_q"][b], 1.0 / Qn, V[b], Vh[b], z_ext=zq[b])
                elif prev_bar is not None and stop == prev_bar[0] - 1:
                    chk.vec(f"{tg} V = previous bar", V[b], prev_bar[1][b], 1e-6)
                    chk.vec(f"{tg} V_h = previous bar_h", Vh[b], prev_bar[2][b], 1e-6)
                check_hvp(E, chk, f"{tg} hvp{stop}", b, tapes[stop], V[b], Vh[b], 1.0 / S, HV[b], HVh[b])
                chk.vec(f"{tg} bar{stop}", bar[b], V[b] - alpha * HV[b], 1e-6)
                chk.vec(f"{tg} bar_h{stop}", barh[b], Vh[b] - alpha * HVh[b], 1e-6)
            prev_bar = (stop, bar, barh)
            if stop == 0:
                final = (out, bar, barh)
    finally:
        hip.resnet12_set_option(1, 0)
        hip.resnet12_set_option(0, 0)
    return chk, lay, final


def check_first_order(hip, ws, dev, run, ep, theta, head0, channels, T, alpha):
    """First-order step (MAML `--first_order`, maml.py:173-177) or T = 0: one tape reused by every inner step, two alternating
    parameter slots, no Hessian-vector products -- the last support pass, the query pass and `bar` = the query pass's gradient."""
    B, S = ep["x_s"].shape[:2]
    Qn, Cimg, H = ep["x_q"].shape[1], ep["x_s"].shape[2], ep["x_s"].shape[3]
    N, Fd = head0.shape[1], channels[-1]
    lay = Layout(channels, Cimg, H)
    chk = Checker()
    hip.resnet12_set_option(0, 1)
    try:
        out = run()
        assert ws.read_status() == 0
        E = Engine(hip, ws, dev, lay, B, N, S, Qn, T, Cimg)
        slots = [E.f32(-1, 0, (B, lay.PSZ), idx=i) for i in range(2)]
        heads = [E.f32(-1, 1, (B, N, Fd + 1), idx=i) for i in range(2)]
        bar, barh = E.f32(-1, 4, (B, lay.PSZ)), E.f32(-1, 5, (B, N, Fd + 1))
        zq = out["logits"].cpu().double()
        cur = T % 2                                                    # the slot the query pass reads
        for b in range(B):
            tg = f"fo e{b}"
            if T > 0:                                                  # the tape holds the LAST inner step (slot (T - 1) % 2 -> slot T % 2)
                prev = (T - 1) % 2
                Gl, dhl = E.f32(-1, 2, (B, lay.PSZ), idx=0), E.f32(-1, 3, (B, N, Fd + 1), idx=0)
                E.T = T
                check_pass(E, chk, f"{tg} t{T - 1}", 0, b, slots[prev][b], heads[prev][b], ep["y_s"][b], 1.0 / S, Gl[b], dhl[b])
                chk.vec(f"{tg} slot", slots[cur][b], slots[prev][b] - alpha * Gl[b], 1e-6)
                chk.vec(f"{tg} head", heads[cur][b], heads[prev][b] - alpha * dhl[b], 1e-6)
            else:
                th0 = torch.cat([t.reshape(-1).double() for t in theta])
                chk.vec(f"{tg} slot0", slots[0][b][:th0.numel()], th0, 1e-7)
            check_pass(E, chk, f"{tg} q", T, b, slots[cur][b], heads[cur][b], ep["y_q"][b], 1.0 / Qn, bar[b], barh[b], z_ext=zq[b])
    finally:
        hip.resnet12_set_option(0, 0)
    return chk, lay, (out, bar, barh)
