"""Autograd-free statement of the Conv4 meta-step: the algebra the HIP kernels implement -- TEST INFRASTRUCTURE ONLY.

``oracle/conv4_ref.py`` lets autograd differentiate twice through the inner loop.  The engine cannot (no autograd graph
inside kernels); it runs a hand-written sweep.  This module is that sweep in plain tensor operations, so that (a) the
derivation is checked against autograd in float64 (tests/test_conv4_manual.py) and (b) every kernel has a one-line
counterpart here to be compared with.

Second order by forward-over-reverse (Pearlmutter's R-operator).  With theta_{t+1} = theta_t - alpha * grad L_s(theta_t) the
adjoint recursion of the outer gradient is

    bar_t = bar_{t+1} - alpha * H_t bar_{t+1},            H_t = Hessian of the support loss at theta_t (symmetric),

and H_t v is the directional derivative of the gradient computation along v: one TANGENT forward pass (every activation's
derivative along v) and one TANGENT backward pass (every backward quantity's derivative along v) over the tape of step t.
Every op is bilinear (conv: forward / backward-data / backward-weight -- a set closed under differentiation) or elementwise
with a constant mask (ReLU, max-pool arg-max), except
  * soft-max cross-entropy:   dz = (p - y) / S,       tangent  dz' = p * (z' - <p, z'>) / S
  * batch-statistic BatchNorm: with xh = (u - mean u) r, r = (var u + eps)^-1/2,
        forward tangent   xh' = r (u' - mean u' - xh mean(xh u'))                      (the BN Jacobian is symmetric)
        backward          du  = g r (dv - mean dv - xh mean(dv xh)),  dg = sum dv xh,  db = sum dv
        backward tangent  du' = (g' r + g r') (dv - d1 - xh d2) + g r (dv' - mean dv' - xh' d2 - xh (mean(dv' xh) + mean(dv xh')))
                          r' = -r^2 mean(xh u'),   d1 = mean dv,  d2 = mean(dv xh)
                          dg' = sum(dv' xh + dv xh'),  db' = sum dv'
Counting conv-sized products per support image and step: forward 1, backward 2, tangent forward 2, tangent backward 4 = 9 (the
first block has no input gradient: 1 + 1 + 1 + 1 = 4): SURVEY.md section 8(d)'s c_s = 9.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def conv(x, W):
    return F.conv2d(x, W, None, padding=1)


def conv_bwd_data(dy, W):
    """dx = dy (*) W^T: a forward 3x3 convolution of dy with the kernel flipped and its channel axes swapped."""
    return F.conv2d(dy, W.flip(2, 3).transpose(0, 1), None, padding=1)


def conv_bwd_weight(x, dy):
    """dW[co, ci, ky, kx] = sum_{m,y,x} dy[m, co, y, x] * xpad[m, ci, y + ky, x + kx]"""
    M, Ci, H, Wd = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    cols = torch.stack([xp[:, :, ky:ky + H, kx:kx + Wd] for ky in range(3) for kx in range(3)], 2)    # [M, Ci, 9, H, W]
    return torch.einsum("mohw,mikhw->oik", dy, cols).reshape(dy.shape[1], Ci, 3, 3)


def _windows(v):
    M, C, H, W = v.shape
    Ho, Wo = H // 2, W // 2
    return v[:, :, :2 * Ho, :2 * Wo].reshape(M, C, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(M, C, Ho, Wo, 4)


def _unwindows(w, H, W):
    M, C, Ho, Wo, _ = w.shape
    out = w.new_zeros(M, C, H, W)
    out[:, :, :2 * Ho, :2 * Wo] = w.reshape(M, C, Ho, Wo, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(M, C, 2 * Ho, 2 * Wo)
    return out


def block_fwd(x, W, g, b, flips=None):
    """conv . BN(batch statistics) . ReLU . max-pool 2 -> (pooled output, tape).
    flips: [(kind, flat window index)] decisions to take the OTHER way ("relu": the unit's ReLU mask, "arg": the window's
    second-largest element instead of the largest) -- used by the parity tests to enumerate decisions that lie within fp32
    noise of a tie (see ambiguous_decisions)."""
    u = conv(x, W)
    mu = u.mean((0, 2, 3), keepdim=True)
    r = (u.var((0, 2, 3), unbiased=False, keepdim=True) + BN_EPS).rsqrt()
    xh = (u - mu) * r
    v = g.view(1, -1, 1, 1) * xh + b.view(1, -1, 1, 1)
    vw = _windows(v)
    mx = vw.max(-1)[0]
    first = ((vw == mx[..., None]).to(torch.int8).cumsum(-1) == 1) & (vw == mx[..., None])     # FIRST maximum of the window
    arg = first.to(torch.int8).argmax(-1)
    mask = (mx > 0).to(x.dtype)
    top2v, top2i = vw.topk(2, dim=-1)
    relu_margin, arg_margin = mx.abs(), top2v[..., 0] - top2v[..., 1]
    margin = float(torch.minimum(relu_margin.min(), arg_margin.min()))                       # distance from a ReLU / arg-max tie
    if flips:
        arg, mask = arg.clone(), mask.clone()
        for kind, idx in flips:
            if kind == "relu":
                mask.view(-1)[idx] = 1.0 - mask.view(-1)[idx]
            else:
                arg.view(-1)[idx] = top2i[..., 1].reshape(-1)[idx]
    return torch.relu(mx), dict(margin=margin, relu_margin=relu_margin, arg_margin=arg_margin, x=x, W=W, g=g.view(1, -1, 1, 1), r=r, xh=xh, arg=arg, mask=mask, shape=u.shape, u=u, mu=mu,
                                xo=torch.relu(mx))


def _scatter(dxo, tp):
    M, C, H, W = tp["shape"]
    w = dxo.new_zeros(*dxo.shape, 4)
    w.scatter_(-1, tp["arg"][..., None], (dxo * tp["mask"])[..., None])
    return _unwindows(w, H, W)


def _gather(v, tp):
    return torch.gather(_windows(v), -1, tp["arg"][..., None])[..., 0] * tp["mask"]


def _mean(t):
    return t.mean((0, 2, 3), keepdim=True)


def block_bwd(dxo, tp, need_dx=True):
    """First-order backward of a block: returns (dx or None, dW, dg, db) and extends the tape with dv / d1 / d2 / du."""
    dv = _scatter(dxo, tp)
    d1, d2 = _mean(dv), _mean(dv * tp["xh"])
    du = tp["g"] * tp["r"] * (dv - d1 - tp["xh"] * d2)
    tp.update(dv=dv, d1=d1, d2=d2, du=du, dxo=dxo)
    dg, db = (dv * tp["xh"]).sum((0, 2, 3)), dv.sum((0, 2, 3))
    return (conv_bwd_data(du, tp["W"]) if need_dx else None), conv_bwd_weight(tp["x"], du), dg, db


def block_tan_fwd(xd, Wd, gd, bd, tp):
    """Tangent of block_fwd along (x', W', g', b') (x' = None for the first block: images are constants)."""
    ud = conv(tp["x"], Wd)
    if xd is not None:
        ud = ud + conv(xd, tp["W"])
    m1, m2 = _mean(ud), _mean(tp["xh"] * ud)
    xhd = tp["r"] * (ud - m1 - tp["xh"] * m2)
    vd = gd.view(1, -1, 1, 1) * tp["xh"] + tp["g"] * xhd + bd.view(1, -1, 1, 1)
    tp.update(xd=xd, Wd=Wd, gd=gd.view(1, -1, 1, 1), xhd=xhd, m2=m2, ud=ud)
    tp["xod"] = _gather(vd, tp)
    return tp["xod"]


def block_tan_bwd(dxod, tp, need_dx=True):
    """Tangent of block_bwd: returns (dx', dW', dg', db')."""
    dvd = _scatter(dxod, tp)
    xh, xhd, dv, r, g = tp["xh"], tp["xhd"], tp["dv"], tp["r"], tp["g"]
    dd1, e1, e2 = _mean(dvd), _mean(dvd * xh), _mean(dv * xhd)
    rd = -r * r * tp["m2"]
    dud = (tp["gd"] * r + g * rd) * (dv - tp["d1"] - xh * tp["d2"]) + g * r * (dvd - dd1 - xhd * tp["d2"] - xh * (e1 + e2))
    n = dv.shape[0] * dv.shape[2] * dv.shape[3]
    dgd, dbd = n * (e1 + e2).reshape(-1), n * dd1.reshape(-1)
    dWd = conv_bwd_weight(tp["x"], dud)
    if tp["xd"] is not None:
        dWd = dWd + conv_bwd_weight(tp["xd"], tp["du"])
    dxd = (conv_bwd_data(dud, tp["W"]) + conv_bwd_data(tp["du"], tp["Wd"])) if need_dx else None
    tp.update(dud=dud, dxod=dxod)
    return dxd, dWd, dgd, dbd


def net_fwd(x, theta, h, flips=None):
    tapes = []
    for i in range(0, len(theta), 3):
        x, tp = block_fwd(x, theta[i], theta[i + 1], theta[i + 2], None if not flips else flips.get(i // 3))
        tapes.append(tp)
    f = x.reshape(x.shape[0], -1)
    z = f @ h[:, :-1].t() + h[:, -1]
    return z, dict(blocks=tapes, f=f, h=h, fshape=x.shape)


def net_bwd(z, y, tape, scale):
    """Gradient of scale * sum_rows CE(z, y) w.r.t. (theta, h); extends the tape with p / dz."""
    p = torch.softmax(z, -1)
    dz = (p - F.one_hot(y, z.shape[1]).to(z.dtype)) * scale
    f, h = tape["f"], tape["h"]
    dh = torch.cat([dz.t() @ f, dz.sum(0)[:, None]], 1)
    dx = (dz @ h[:, :-1]).reshape(tape["fshape"])
    tape.update(p=p, dz=dz)
    g = []
    for i in reversed(range(len(tape["blocks"]))):
        dx, dW, dg, db = block_bwd(dx, tape["blocks"][i], need_dx=i > 0)
        g = [dW, dg, db] + g
    return g, dh


def net_hvp(tape, vth, vh, scale):
    """(H v) for v = (vth, vh): derivative of net_bwd's outputs along v, by one tangent forward + one tangent backward pass."""
    xd = None
    for i, tp in enumerate(tape["blocks"]):
        xd = block_tan_fwd(xd, vth[3 * i], vth[3 * i + 1], vth[3 * i + 2], tp)
    f, h, p, dz = tape["f"], tape["h"], tape["p"], tape["dz"]
    fd = xd.reshape(f.shape)
    zd = fd @ h[:, :-1].t() + f @ vh[:, :-1].t() + vh[:, -1]
    dzd = p * (zd - (p * zd).sum(-1, keepdim=True)) * scale
    dhd = torch.cat([dzd.t() @ f + dz.t() @ fd, dzd.sum(0)[:, None]], 1)
    dxd = (dzd @ h[:, :-1] + dz @ vh[:, :-1]).reshape(tape["fshape"])
    out = []
    for i in reversed(range(len(tape["blocks"]))):
        dxd, dWd, dgd, dbd = block_tan_bwd(dxd, tape["blocks"][i], need_dx=i > 0)
        out = [dWd, dgd, dbd] + out
    return out, dhd


def episode_grads(theta, h0, x_s, y_s, x_q, y_q, T, alpha, first_order=False, trace=None, flips=None):
    """(query logits, query loss, d loss / d theta, d loss / d h0) of one episode, no autograd anywhere.
    trace (dict): receives every intermediate (tapes of the inner steps, the query pass, gradients, adjoints, H v).
    flips: {pass (inner step t, or T = query): {block: [(kind, index)]}} decisions taken the other way (block_fwd)."""
    th, h, tapes = [t for t in theta], h0, []
    if trace is not None:
        trace.update(tapes=tapes, params=[list(th)], heads=[h], grads=[], dh=[], hv=[])
    S = x_s.shape[0]
    for t in range(T):
        z, tape = net_fwd(x_s, th, h, None if not flips else flips.get(t))
        g, dh = net_bwd(z, y_s, tape, 1.0 / S)
        tapes.append(tape)
        th = [p - alpha * gi for p, gi in zip(th, g)]
        h = h - alpha * dh
        if trace is not None:
            trace["grads"].append(g); trace["dh"].append(dh); trace["params"].append(list(th)); trace["heads"].append(h)
    zq, tq = net_fwd(x_q, th, h, None if not flips else flips.get(T))
    loss = F.cross_entropy(zq, y_q)
    bar_th, bar_h = net_bwd(zq, y_q, tq, 1.0 / x_q.shape[0])
    if trace is not None:
        trace.update(query=tq, bar_T=(list(bar_th), bar_h))
    if not first_order:
        for tape in reversed(tapes):
            hv_th, hv_h = net_hvp(tape, bar_th, bar_h, 1.0 / S)
            if trace is not None:
                trace["hv"].append((hv_th, hv_h))
            bar_th = [b - alpha * v for b, v in zip(bar_th, hv_th)]
            bar_h = bar_h - alpha * hv_h
    return zq, loss, bar_th, bar_h


def min_decision_margin(theta, h0, x_s, y_s, x_q, y_q, T, alpha):
    """Smallest distance of any ReLU / max-pool decision of the episode (all inner steps + the query pass) from a tie.  A
    decision closer to a tie than the fp32 noise of its input (~1e-6) may fall either way in two correct implementations; the
    forward value barely moves, but the gradient routed through it switches on or off -- an O(1/sqrt(#terms)) change of a
    weight gradient.  Parity cases are drawn so that this margin is comfortably above the noise."""
    tr = {}
    episode_grads(theta, h0, x_s, y_s, x_q, y_q, T, alpha, first_order=True, trace=tr)
    return min(tp["margin"] for tape in tr["tapes"] + [tr["query"]] for tp in tape["blocks"])


def ambiguous_decisions(trace, tol, limit=8):
    """[(margin, pass, block, kind, flat window index)] of the ReLU / arg-max decisions of an episode (trace of episode_grads)
    that lie within `tol` of a tie, smallest margins first, at most `limit`."""
    out = []
    for p, tape in enumerate(trace["tapes"] + [trace["query"]]):
        for blk, tp in enumerate(tape["blocks"]):
            for kind, m in (("relu", tp["relu_margin"]), ("arg", tp["arg_margin"])):
                idx = (m.reshape(-1) < tol).nonzero(as_tuple=True)[0]
                out += [(float(m.reshape(-1)[i]), p, blk, kind, int(i)) for i in idx]
    return sorted(out)[:limit]


def flips_of(subset):
    """{pass: {block: [(kind, index)]}} from a subset of ambiguous_decisions entries."""
    fl = {}
    for _, p, blk, kind, idx in subset:
        fl.setdefault(p, {}).setdefault(blk, []).append((kind, idx))
    return fl
