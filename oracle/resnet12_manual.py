"""Autograd-free statement of the ResNet-12 meta-step: the algebra the bf16 HIP kernels implement -- TEST INFRASTRUCTURE ONLY.

``oracle/resnet12_ref.py`` lets autograd differentiate twice through the inner loop; the engine runs a hand-written sweep
(forward-over-reverse second order, as ``oracle/conv4_manual.py`` explains for Conv4).  This module is that sweep in plain tensor
operations, with one extra ingredient: a rounding hook ``rnd``.

  * ``rnd = identity`` (float64): the sweep equals autograd to 1e-9 -- the DERIVATION is checked (tests/test_resnet12_manual.py);
  * ``rnd = bf16 round-to-nearest-even``: the sweep rounds exactly where the engine stores bf16 -- images, the weights handed to
    a matrix product (fp32 masters are kept and updated in fp32), every conv output ``u`` / ``u'``, every activation, every
    gradient map ``da`` / ``du`` and their tangents -- with fp32-or-better arithmetic in between.  Batch statistics are taken from
    the STORED (rounded) map, as the engine's conv epilogue does.  The GPU tests compare against this form tightly (what is left
    is summation order and the rare one-ulp flip of a bf16 rounding) and against the float64 form at bf16's own noise level.

Per block (resnet12_ref.py): a1 = lrelu(BN1(conv(x, W1))), a2 = lrelu(BN2(conv(a1, W2))), s = BN3(conv(a2, W3)) + BNs(conv1x1(x, Ws)),
out = maxpool2(lrelu(s)).  LeakyReLU and max-pool have constant masks / arg-max along a tangent; BN's tangent formulas are those of
conv4_manual.py; the residual join adds the two BN outputs, so both receive the same incoming gradient.
"""
import torch
import torch.nn.functional as F

from .resnet12_ref import BN_EPS, SLOPE, PER_BLOCK


def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _id(t):
    return t


# ---- convolutions (k = 3: pad 1; k = 1: no pad) ---------------------------------------------------------------------------------
def conv(x, W):
    return F.conv2d(x, W, None, padding=W.shape[-1] // 2)


def conv_bwd_data(dy, W):
    k = W.shape[-1]
    return F.conv2d(dy, W.flip(2, 3).transpose(0, 1), None, padding=k // 2)


def conv_bwd_weight(x, dy, k):
    if k == 1:
        return torch.einsum("mohw,mihw->oi", dy, x)[:, :, None, None]
    M, Ci, H, Wd = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    cols = torch.stack([xp[:, :, ky:ky + H, kx:kx + Wd] for ky in range(3) for kx in range(3)], 2)
    return torch.einsum("mohw,mikhw->oik", dy, cols).reshape(dy.shape[1], Ci, 3, 3)


def _mean(t):
    return t.mean((0, 2, 3), keepdim=True)


def _c(v):
    return v.view(1, -1, 1, 1)


# ---- batch-statistic BN ---------------------------------------------------------------------------------------------------------
def bn_fwd(u, g, b):
    mu = _mean(u)
    r = (u.var((0, 2, 3), unbiased=False, keepdim=True) + BN_EPS).rsqrt()
    xh = (u - mu) * r
    return _c(g) * xh + _c(b), dict(r=r, xh=xh, g=_c(g))


def bn_bwd(dv, tp):
    d1, d2 = _mean(dv), _mean(dv * tp["xh"])
    du = tp["g"] * tp["r"] * (dv - d1 - tp["xh"] * d2)
    tp.update(dv=dv, d1=d1, d2=d2)
    return du, (dv * tp["xh"]).sum((0, 2, 3)), dv.sum((0, 2, 3))


def bn_tan_fwd(ud, gd, bd, tp):
    m1, m2 = _mean(ud), _mean(tp["xh"] * ud)
    xhd = tp["r"] * (ud - m1 - tp["xh"] * m2)
    tp.update(gd=_c(gd), xhd=xhd, m2=m2)
    return _c(gd) * tp["xh"] + tp["g"] * xhd + _c(bd)


def bn_tan_bwd(dvd, tp):
    xh, xhd, dv, r, g = tp["xh"], tp["xhd"], tp["dv"], tp["r"], tp["g"]
    dd1, e1, e2 = _mean(dvd), _mean(dvd * xh), _mean(dv * xhd)
    rd = -r * r * tp["m2"]
    dud = (tp["gd"] * r + g * rd) * (dv - tp["d1"] - xh * tp["d2"]) + g * r * (dvd - dd1 - xhd * tp["d2"] - xh * (e1 + e2))
    n = dv.shape[0] * dv.shape[2] * dv.shape[3]
    return dud, n * (e1 + e2).reshape(-1), n * dd1.reshape(-1)


# ---- max-pool 2 (floor) with the FIRST maximum of a window ---------------------------------------------------------------------------
def _windows(v):
    M, C, H, W = v.shape
    Ho, Wo = H // 2, W // 2
    return v[:, :, :2 * Ho, :2 * Wo].reshape(M, C, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(M, C, Ho, Wo, 4)


def _unwindows(w, H, W):
    M, C, Ho, Wo, _ = w.shape
    out = w.new_zeros(M, C, H, W)
    out[:, :, :2 * Ho, :2 * Wo] = w.reshape(M, C, Ho, Wo, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(M, C, 2 * Ho, 2 * Wo)
    return out


def _lmask(v):
    return torch.where(v > 0, torch.ones_like(v), torch.full_like(v, SLOPE))


# ---- one residual block ---------------------------------------------------------------------------------------------------------
def block_fwd(x, p, rnd):
    W1, g1, b1, W2, g2, b2, W3, g3, b3, Ws, gs, bs = p
    tp = dict(x=x, W=[rnd(W1), rnd(W2), rnd(W3), rnd(Ws)])
    u1 = rnd(conv(x, tp["W"][0])); v1, t1 = bn_fwd(u1, g1, b1); a1 = rnd(F.leaky_relu(v1, SLOPE))
    u2 = rnd(conv(a1, tp["W"][1])); v2, t2 = bn_fwd(u2, g2, b2); a2 = rnd(F.leaky_relu(v2, SLOPE))
    u3 = rnd(conv(a2, tp["W"][2])); v3, t3 = bn_fwd(u3, g3, b3)
    us = rnd(conv(x, tp["W"][3])); vs, ts = bn_fwd(us, gs, bs)
    s = v3 + vs
    ls = F.leaky_relu(s, SLOPE)
    lw = _windows(ls)
    mx = lw.max(-1)[0]
    first = ((lw == mx[..., None]).to(torch.int8).cumsum(-1) == 1) & (lw == mx[..., None])
    arg = first.to(torch.int8).argmax(-1)
    out = rnd(mx)
    tp.update(bn=[t1, t2, t3, ts], a=[a1, a2], m=[_lmask(v1), _lmask(v2)], ms=_lmask(s), arg=arg, shape=s.shape, u=[u1, u2, u3, us],
              out=out)
    return out, tp


def _scatter(dxo, tp):
    M, C, H, W = tp["shape"]
    w = dxo.new_zeros(*dxo.shape, 4)
    w.scatter_(-1, tp["arg"][..., None], dxo[..., None])
    return _unwindows(w, H, W) * tp["ms"]


def _gather(v, tp):
    return torch.gather(_windows(v * tp["ms"]), -1, tp["arg"][..., None])[..., 0]


def block_bwd(do, tp, rnd, need_dx=True):
    """-> (dx or None, 12 parameter gradients)."""
    W = tp["W"]
    ds = _scatter(do, tp)
    du3, dg3, db3 = bn_bwd(ds, tp["bn"][2]); du3 = rnd(du3)
    dus, dgs, dbs = bn_bwd(ds, tp["bn"][3]); dus = rnd(dus)
    dW3 = conv_bwd_weight(tp["a"][1], du3, 3)
    da2 = rnd(conv_bwd_data(du3, W[2]))
    du2, dg2, db2 = bn_bwd(da2 * tp["m"][1], tp["bn"][1]); du2 = rnd(du2)
    dW2 = conv_bwd_weight(tp["a"][0], du2, 3)
    da1 = rnd(conv_bwd_data(du2, W[1]))
    du1, dg1, db1 = bn_bwd(da1 * tp["m"][0], tp["bn"][0]); du1 = rnd(du1)
    dW1 = conv_bwd_weight(tp["x"], du1, 3)
    dWs = conv_bwd_weight(tp["x"], dus, 1)
    dx = rnd(conv_bwd_data(du1, W[0]) + conv_bwd_data(dus, W[3])) if need_dx else None
    tp.update(du=[du1, du2, du3, dus], da=[da1, da2], do=do)
    return dx, [dW1, dg1, db1, dW2, dg2, db2, dW3, dg3, db3, dWs, dgs, dbs]


def block_tan_fwd(xd, pd, tp, rnd):
    """Tangent of block_fwd along (x', the 12 parameter directions pd); x' = None for the first block (images are constants)."""
    W1d, g1d, b1d, W2d, g2d, b2d, W3d, g3d, b3d, Wsd, gsd, bsd = pd
    W, x = tp["W"], tp["x"]
    Wd = [rnd(W1d), rnd(W2d), rnd(W3d), rnd(Wsd)]

    def two(xa, Wa_d, xa_d, Wa):
        y = conv(xa, Wa_d)
        return y if xa_d is None else y + conv(xa_d, Wa)
    u1d = rnd(two(x, Wd[0], xd, W[0])); a1d = rnd(tp["m"][0] * bn_tan_fwd(u1d, g1d, b1d, tp["bn"][0]))
    u2d = rnd(two(tp["a"][0], Wd[1], a1d, W[1])); a2d = rnd(tp["m"][1] * bn_tan_fwd(u2d, g2d, b2d, tp["bn"][1]))
    u3d = rnd(two(tp["a"][1], Wd[2], a2d, W[2])); v3d = bn_tan_fwd(u3d, g3d, b3d, tp["bn"][2])
    usd = rnd(two(x, Wd[3], xd, W[3])); vsd = bn_tan_fwd(usd, gsd, bsd, tp["bn"][3])
    outd = rnd(_gather(v3d + vsd, tp))
    tp.update(xd=xd, Wd=Wd, ad=[a1d, a2d], ud=[u1d, u2d, u3d, usd], outd=outd)
    return outd


def block_tan_bwd(dod, tp, rnd, need_dx=True):
    W, Wd, x, xd = tp["W"], tp["Wd"], tp["x"], tp["xd"]
    du1, du2, du3, dus = tp["du"]
    dsd = _scatter(dod, tp)
    du3d, dg3d, db3d = bn_tan_bwd(dsd, tp["bn"][2]); du3d = rnd(du3d)
    dusd, dgsd, dbsd = bn_tan_bwd(dsd, tp["bn"][3]); dusd = rnd(dusd)
    dW3d = conv_bwd_weight(tp["a"][1], du3d, 3) + conv_bwd_weight(tp["ad"][1], du3, 3)
    da2d = rnd(conv_bwd_data(du3d, W[2]) + conv_bwd_data(du3, Wd[2]))
    du2d, dg2d, db2d = bn_tan_bwd(da2d * tp["m"][1], tp["bn"][1]); du2d = rnd(du2d)
    dW2d = conv_bwd_weight(tp["a"][0], du2d, 3) + conv_bwd_weight(tp["ad"][0], du2, 3)
    da1d = rnd(conv_bwd_data(du2d, W[1]) + conv_bwd_data(du2, Wd[1]))
    du1d, dg1d, db1d = bn_tan_bwd(da1d * tp["m"][0], tp["bn"][0]); du1d = rnd(du1d)
    dW1d = conv_bwd_weight(x, du1d, 3)
    dWsd = conv_bwd_weight(x, dusd, 1)
    if xd is not None:
        dW1d = dW1d + conv_bwd_weight(xd, du1, 3)
        dWsd = dWsd + conv_bwd_weight(xd, dus, 1)
    dxd = None
    if need_dx:
        dxd = rnd(conv_bwd_data(du1d, W[0]) + conv_bwd_data(du1, Wd[0]) + conv_bwd_data(dusd, W[3]) + conv_bwd_data(dus, Wd[3]))
    tp.update(dud=[du1d, du2d, du3d, dusd], dad=[da1d, da2d], dod=dod)
    return dxd, [dW1d, dg1d, db1d, dW2d, dg2d, db2d, dW3d, dg3d, db3d, dWsd, dgsd, dbsd]


# ---- the network, the head, one episode ---------------------------------------------------------------------------------------------
def net_fwd(x, theta, h, rnd):
    tapes = []
    x = rnd(x)
    for i in range(0, len(theta), PER_BLOCK):
        x, tp = block_fwd(x, theta[i:i + PER_BLOCK], rnd)
        tapes.append(tp)
    f = x.mean((2, 3))
    z = f @ h[:, :-1].t() + h[:, -1]
    return z, dict(blocks=tapes, f=f, h=h, oshape=x.shape)


def net_bwd(z, y, tape, scale, rnd):
    p = torch.softmax(z, -1)
    dz = (p - F.one_hot(y, z.shape[1]).to(z.dtype)) * scale
    f, h = tape["f"], tape["h"]
    dh = torch.cat([dz.t() @ f, dz.sum(0)[:, None]], 1)
    M, C, Ho, Wo = tape["oshape"]
    dx = rnd(((dz @ h[:, :-1]) / (Ho * Wo))[:, :, None, None].expand(M, C, Ho, Wo))
    tape.update(p=p, dz=dz)
    g = []
    for i in reversed(range(len(tape["blocks"]))):
        dx, gb = block_bwd(dx, tape["blocks"][i], rnd, need_dx=i > 0)
        g = gb + g
    return g, dh


def net_hvp(tape, vth, vh, scale, rnd):
    xd = None
    for i, tp in enumerate(tape["blocks"]):
        xd = block_tan_fwd(xd, vth[PER_BLOCK * i:PER_BLOCK * (i + 1)], tp, rnd)
    f, h, p, dz = tape["f"], tape["h"], tape["p"], tape["dz"]
    fd = xd.mean((2, 3))
    zd = fd @ h[:, :-1].t() + f @ vh[:, :-1].t() + vh[:, -1]
    dzd = p * (zd - (p * zd).sum(-1, keepdim=True)) * scale
    dhd = torch.cat([dzd.t() @ f + dz.t() @ fd, dzd.sum(0)[:, None]], 1)
    M, C, Ho, Wo = tape["oshape"]
    dfd = dzd @ h[:, :-1] + dz @ vh[:, :-1]
    dxd = rnd((dfd / (Ho * Wo))[:, :, None, None].expand(M, C, Ho, Wo))
    tape.update(fd=fd, dzd=dzd, dfd=dfd)
    out = []
    for i in reversed(range(len(tape["blocks"]))):
        dxd, gb = block_tan_bwd(dxd, tape["blocks"][i], rnd, need_dx=i > 0)
        out = gb + out
    return out, dhd


def episode_grads(theta, h0, x_s, y_s, x_q, y_q, T, alpha, first_order=False, rnd=_id, trace=None):
    """(query logits, query loss, d loss / d theta, d loss / d h0) of one episode, no autograd anywhere."""
    th, h, tapes = [t for t in theta], h0, []
    S = x_s.shape[0]
    steps = []                                             # (theta_t, head_t, g_t, dh_t) of every inner step, for `trace`
    for _ in range(T):
        z, tape = net_fwd(x_s, th, h, rnd)
        g, dh = net_bwd(z, y_s, tape, 1.0 / S, rnd)
        tapes.append(tape)
        steps.append((list(th), h, list(g), dh))
        th = [p - alpha * gi for p, gi in zip(th, g)]
        h = h - alpha * dh
    zq, tq = net_fwd(x_q, th, h, rnd)
    loss = F.cross_entropy(zq, y_q)
    bar_th, bar_h = net_bwd(zq, y_q, tq, 1.0 / x_q.shape[0], rnd)
    if trace is not None:
        trace.update(tapes=tapes, query=tq, bar_T=(list(bar_th), bar_h), theta_T=list(th), head_T=h, hv=[], steps=steps, zq=zq, bars=[])
    if not first_order:
        stop = 0 if trace is None else trace.get("hvp_stop", 0)
        for t in reversed(range(stop, len(tapes))):
            hv_th, hv_h = net_hvp(tapes[t], bar_th, bar_h, 1.0 / S, rnd)
            if trace is not None:
                trace["hv"].append((hv_th, hv_h))
                trace["V"] = (list(bar_th), bar_h)              # direction of the LAST Hessian-vector product
            bar_th = [b - alpha * v for b, v in zip(bar_th, hv_th)]
            bar_h = bar_h - alpha * hv_h
    return zq, loss, bar_th, bar_h


def fumi_meta_step(theta, phi, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head, rnd=_id):
    """FuMI meta-step through the manual sweep (the hypernetwork and its gradient by autograd on the tiny head path): the same
    dict as resnet12_ref.fumi_meta_step."""
    from . import fumi_ref as R
    B = x_s.shape[0]
    phis = [p.detach().clone().requires_grad_(True) for p in phi]
    g_theta = [torch.zeros_like(t) for t in theta]
    logits, loss_b, hb_terms = [], [], []
    for b in range(B):
        c = R.class_text_select(text_s[b], y_s[b], n_way)
        h = R.hyper_net(c, phis, tanh_head)
        zq, loss, bth, bh = episode_grads(theta, h.detach(), x_s[b], y_s[b], x_q[b], y_q[b], T, alpha, rnd=rnd)
        logits.append(zq); loss_b.append(loss)
        for a, g in zip(g_theta, bth):
            a += g / B
        hb_terms.append((h * bh).sum() / B)
    g_phi = torch.autograd.grad(torch.stack(hb_terms).sum(), phis, allow_unused=True)
    out = R._pack(logits, loss_b, y_q)
    out["loss"] = (torch.stack(loss_b).sum() / B).detach()
    out["g_theta"] = g_theta
    out["g_phi"] = [torch.zeros_like(p) if g is None else g for g, p in zip(g_phi, phis)]
    return out
