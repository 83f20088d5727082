"""GPU parity of the Conv4 image encoder at the im_net seam (csrc/conv_gemm.hip, conv_ew.hip, conv4.hip), through the C ABI.

PARITY UNPINNED for the convolutional part (the reference has no such encoder: oracle/conv4_ref.py's header).  Checkers:
  * the three convolution products against torch.nn.functional on the host;
  * EVERY intermediate tensor of a meta-step (fumi_hip_conv4_probe) against the autograd-free sweep oracle/conv4_manual.py,
    which tests/test_conv4_manual.py ties to autograd at 1e-9 in float64 -- so a failure names the kernel;
  * whole meta-steps (FuMI and MAML heads, T = 0..3, first order, evaluation mode) against autograd through
    oracle/conv4_ref.py: logits 1e-4 of max|logit|, integer predictions bit-exact outside the 1e-5 margin, every meta-gradient
    1e-3 of the tensor's scale; one full-size 84 x 84 5-way episode pair.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import casegen as cg
from oracle import conv4_manual as M
from oracle import conv4_ref as C
from oracle import fumi_ref as R
from helpers import rel_to_max, safe_margin_mask

pytestmark = pytest.mark.gpu
LOGIT_TOL, GRAD_TOL, MARGIN = 1e-4, 1e-3, 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ws(dev):
    from fumi_amd import hip
    return hip.Workspace.get(dev)


def _g(t, dev):
    return t.to(dev).contiguous()


# ---- the three convolution products ------------------------------------------------------------------------------------
@pytest.mark.parametrize("Mi,H,W", [(3, 10, 10), (2, 21, 21), (5, 42, 42), (1, 7, 13), (40, 5, 5)])
def test_conv3x3_products_match_torch(Mi, H, W, dev, ws):
    from fumi_amd import hip
    g = torch.Generator().manual_seed(Mi * 100 + H)
    x = torch.randn(Mi, 64, H, W, generator=g)
    Wt = torch.randn(64, 64, 3, 3, generator=g) / 24
    dy = torch.randn(Mi, 64, H, W, generator=g)
    cl = lambda t: t.permute(0, 2, 3, 1).contiguous()
    y = hip.conv3x3_fwd(ws, _g(cl(x), dev), _g(Wt, dev)).cpu().permute(0, 3, 1, 2)
    assert rel_to_max(y, F.conv2d(x.double(), Wt.double(), padding=1)) <= 2e-6
    dx = hip.conv3x3_bwd_data(ws, _g(cl(dy), dev), _g(Wt, dev)).cpu().permute(0, 3, 1, 2)
    assert rel_to_max(dx, M.conv_bwd_data(dy.double(), Wt.double())) <= 2e-6
    dW = hip.conv3x3_bwd_weight(ws, _g(cl(x), dev), _g(cl(dy), dev)).cpu()
    assert rel_to_max(dW, M.conv_bwd_weight(x.double(), dy.double())) <= 5e-6


def test_conv3x3_asymmetric_identity(dev, ws):
    """A one-hot kernel (output channel o copies input channel pi(o) shifted by one tap) catches a transposed channel map, a
    flipped tap order and a swapped row / column shift."""
    from fumi_amd import hip
    H, W = 9, 12
    x = torch.arange(2 * 64 * H * W, dtype=torch.float32).reshape(2, 64, H, W) % 251 - 100.0
    Wt = torch.zeros(64, 64, 3, 3)
    for o in range(64):
        Wt[o, (o * 7 + 3) % 64, o % 3, (o // 3) % 3] = 1.0
    y = hip.conv3x3_fwd(ws, _g(x.permute(0, 2, 3, 1), dev), _g(Wt, dev)).cpu().permute(0, 3, 1, 2)
    assert torch.equal(y, F.conv2d(x, Wt, padding=1))


def test_small_exported_ops(dev, ws):
    from fumi_amd import hip
    g = torch.Generator().manual_seed(3)
    z, y = torch.randn(37, 5, generator=g), torch.randint(0, 5, (37,), generator=g)
    loss, dz, preds = hip.ce_fwd_bwd(ws, _g(z, dev), _g(y, dev))
    zz = z.clone().requires_grad_(True)
    ref = F.cross_entropy(zz, y)
    assert abs(float(loss) - float(ref.detach())) < 1e-6 and rel_to_max(dz.cpu(), torch.autograd.grad(ref, zz)[0]) < 1e-5
    assert torch.equal(preds.cpu(), z.max(-1)[1])
    p, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    assert torch.allclose(hip.sgd_axpy(ws, _g(p, dev), 0.01, _g(gr, dev)).cpu(), p - 0.01 * gr, atol=1e-7)
    x, ys = torch.randn(3, 12, 20, generator=g), torch.randint(0, 4, (3, 12), generator=g)
    ys[0][ys[0] == 2] = 1                                               # an empty class -> count clamped to 1 -> zeros
    out = hip.proto_reduce(ws, _g(x, dev), _g(ys, dev), 4).cpu()
    ref = torch.stack([torch.stack([x[b][ys[b] == n].sum(0) / max(int((ys[b] == n).sum()), 1) for n in range(4)]) for b in range(3)])
    assert torch.allclose(out, ref, atol=1e-6)
    assert ws.read_status() == 0


# ---- every intermediate of a meta-step ------------------------------------------------------------------------------------
def _unpad(flat, B, Mi, H, W):
    """padded channels-last [B*M][(H+2)(W+2)][64] -> NCHW [B, M, 64, H, W]; also returns the largest |border| value."""
    t = flat.reshape(B, Mi, H + 2, W + 2, 64)
    inner = t[:, :, 1:-1, 1:-1, :]
    frame = t.clone()
    frame[:, :, 1:-1, 1:-1, :] = 0
    return inner.permute(0, 1, 4, 2, 3), float(frame.abs().max())


def _case(seed, B, N, K, Q, Cin, H, W, nblk):
    ep = C.make_image_episodes(seed, B, N, K, Q, Cin, H, W, 12)
    theta = C.make_conv4_params(seed, Cin, 64, nblk)
    Fd = C.feature_dim(H, W, 64, nblk)
    return ep, theta, Fd


def _slab_to_torch(slab, Cin, nblk):
    """One episode's parameter slab in the engine's canonical layouts (csrc/conv4.hip: W1 [64][32], W_l [tap][co][ci], BN weight,
    BN bias per block) -> tensors in torch layouts."""
    out, off = [], 0
    for l in range(nblk):
        if l == 0:
            out.append(slab[off:off + 2048].reshape(64, 32)[:, :Cin * 9].reshape(64, Cin, 3, 3)); off += 2048
        else:
            out.append(slab[off:off + 36864].reshape(9, 64, 64).permute(1, 2, 0).reshape(64, 64, 3, 3)); off += 36864
        out += [slab[off:off + 64], slab[off + 64:off + 128]]; off += 128
    return out


TIE_TOL = 3e-6


def _match_episodes(hip, ws, dev, logits, theta, heads, ep, T, alpha, first_order, need_grad=True):
    """Per episode: the float64 autograd-free sweep (oracle/conv4_manual.py, tied to autograd at 1e-9) against the engine's query
    logits and its per-episode meta-gradients (fumi_hip_conv4_probe).  A ReLU / arg-max decision within fp32 noise of a tie may
    fall either way in two correct fp32 implementations: the forward value does not move, but the gradient routed through that
    unit switches on or off and moves a weight gradient by O(1/sqrt(#terms)) (tests/dev/probe_conv4_flip.py).  With ~5e5
    decisions per case such a tie (margin < 3e-6) is present in most cases, so the checker enumerates the decisions of the
    float64 sweep that lie within TIE_TOL of a tie (at most 6) and requires the engine to agree with ONE assignment of them, at
    the usual tolerances.  Returns the matched per-episode (logits, loss, d theta, d head, trace)."""
    import itertools
    B, Cin, nblk = logits.shape[0], ep["x_s"].shape[2], len(theta) // 3
    th64 = [t.double() for t in theta]
    if need_grad:
        bar = hip.conv4_probe(ws, dev, -1, 4).cpu().reshape(B, -1)
        barh = hip.conv4_probe(ws, dev, -1, 5).cpu().reshape(B, heads.shape[1], -1)
    out = []
    for b in range(B):
        args = (th64, heads[b].double(), ep["x_s"][b].double(), ep["y_s"][b], ep["x_q"][b].double(), ep["y_q"][b], T, alpha, first_order)
        tr0 = {}
        M.episode_grads(*args, trace=tr0)
        amb = M.ambiguous_decisions(tr0, TIE_TOL, limit=6)
        subsets = [()] + [c for k in range(1, len(amb) + 1) for c in itertools.combinations(amb, k)]
        got = _slab_to_torch(bar[b], Cin, nblk) + [barh[b]] if need_grad else None
        errs = None
        for sub in subsets:
            tr = {}
            zq, loss, bth, bh = M.episode_grads(*args, trace=tr, flips=M.flips_of(sub))
            e_log = rel_to_max(logits[b], zq)
            e_g = 0.0
            if need_grad:
                ref = bth + [bh]
                floor = max(0.02 * max(float(r.abs().max()) for r in ref), 1e-9)
                e_g = max(rel_to_max(a, r, floor) for a, r in zip(got, ref))
            errs = errs or (e_log, e_g)
            if e_log <= LOGIT_TOL and e_g <= GRAD_TOL:
                out.append((zq, loss, bth, bh, tr))
                break
        else:
            raise AssertionError(f"episode {b}: logits / gradient error {errs[0]:.2e} / {errs[1]:.2e} against the float64 sweep, and no "
                                 f"assignment of its {len(amb)} near-tie decisions (margins {[f'{m[0]:.1e}' for m in amb]}) explains it")
    return out


def _cmp(name, got, ref, tol=2e-4):
    e = rel_to_max(got, ref, floor=1e-6)
    assert e <= tol, f"{name}: rel-to-max error {e:.3e}"


@pytest.fixture(params=["fused-block1", "plain"])
def c1mode(request):
    """Block 1 either recomputed band by band (csrc/conv_first.hip, the default) or through the plain passes."""
    from fumi_amd import hip
    hip.conv4_set_option(0, 1 if request.param == "fused-block1" else 0)
    yield request.param
    hip.conv4_set_option(0, 1)


@pytest.mark.parametrize("shape", [(3, 12, 12, 2), (1, 20, 20, 3)])
def test_every_intermediate_matches_the_manual_sweep(shape, dev, ws, c1mode):
    from fumi_amd import hip
    Cin, H, W, nblk = shape
    fused = c1mode == "fused-block1"
    B, N, K, Q, T, alpha = 2, 3, 2, 3, 2, 0.05

    ep, theta, Fd = _case(11, B, N, K, Q, Cin, H, W, nblk)
    S, Qn = N * K, N * Q
    g = torch.Generator().manual_seed(5)
    head = torch.cat([torch.randn(N, Fd, generator=g) * 0.2, torch.rand(N, 1, generator=g) * 0.2 - 0.1], 1)
    p = theta + [head[:, :-1].contiguous(), head[:, -1].contiguous()]
    out = hip.maml_conv4_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                              [_g(t, dev) for t in p], T, alpha)
    assert ws.read_status() == 0
    probe = lambda pa, kind, blk=0: hip.conv4_probe(ws, dev, pa, kind, blk).cpu()
    # the float64 sweep of each episode, with its near-tie decisions taken the way the engine took them
    traces = [m[4] for m in _match_episodes(hip, ws, dev, out["logits"].cpu(), theta, head[None].expand(B, -1, -1), ep, T, alpha, False)]
    geo = [(H >> l, W >> l) for l in range(nblk + 1)]
    for t in range(T + 1):                                              # support steps, then the query pass
        Mi = S if t < T else Qn
        tapes = [tr["tapes"][t] if t < T else tr["query"] for tr in traces]
        for l in range(nblk):
            Hl, Wl = geo[l]
            if not (fused and l == 0):                              # (the fused path never stores block 1's maps)
                u, bu = _unpad(probe(t, 0, l), B, Mi, Hl, Wl)
                _cmp(f"pass {t} block {l} u", u, torch.stack([tp["blocks"][l]["u"] for tp in tapes]))
                assert bu == 0.0, f"pass {t} block {l}: conv output border not zero"
            xo = probe(t, 1, l)
            ref_xo = torch.stack([tp["blocks"][l]["xo"] for tp in tapes])
            if l + 1 < nblk:
                xo, bx = _unpad(xo, B, Mi, *geo[l + 1])
                assert bx == 0.0
            else:
                xo = xo.reshape(B, Mi, -1); ref_xo = ref_xo.reshape(B, Mi, -1)
            _cmp(f"pass {t} block {l} pooled", xo, ref_xo)
        _cmp(f"pass {t} p", probe(t, 5).reshape(B, Mi, N), torch.stack([tp["p"] for tp in tapes]))
        _cmp(f"pass {t} dz", probe(t, 6).reshape(B, Mi, N), torch.stack([tp["dz"] for tp in tapes]))
        for l in reversed(range(nblk)):
            Hl, Wl = geo[l]
            dxo = probe(t, 3, l)
            ref = torch.stack([tp["blocks"][l]["dxo"] for tp in tapes])
            if l + 1 < nblk:
                dxo, _ = _unpad(dxo, B, Mi, *geo[l + 1])
            else:
                dxo = dxo.reshape(ref.shape)
            _cmp(f"pass {t} block {l} dxo", dxo, ref)
            if fused and l == 0:
                continue
            du, bd = _unpad(probe(t, 2, l), B, Mi, Hl, Wl)
            _cmp(f"pass {t} block {l} du", du, torch.stack([tp["blocks"][l]["du"] for tp in tapes]))
            assert bd == 0.0, f"pass {t} block {l}: du border not zero"
    # tangent scratch of the last Hessian-vector product (inner step 0)
    tp0 = [tr["tapes"][0] for tr in traces]
    for l in range(nblk):
        Hl, Wl = geo[l]
        if not (fused and l == 0):
            ud, _ = _unpad(probe(T + 1, 0, l), B, S, Hl, Wl)
            _cmp(f"tangent block {l} u'", ud, torch.stack([tp["blocks"][l]["ud"] for tp in tp0]))
        xod = probe(T + 1, 1, l)
        ref = torch.stack([tp["blocks"][l]["xod"] for tp in tp0])
        xod = _unpad(xod, B, S, *geo[l + 1])[0] if l + 1 < nblk else xod.reshape(ref.shape)
        _cmp(f"tangent block {l} x'", xod, ref)
    for l in reversed(range(nblk)):
        Hl, Wl = geo[l]
        dxod = probe(T + 1, 3, l)
        ref = torch.stack([tp["blocks"][l]["dxod"] for tp in tp0])
        dxod = _unpad(dxod, B, S, *geo[l + 1])[0] if l + 1 < nblk else dxod.reshape(ref.shape)
        _cmp(f"tangent block {l} dxo'", dxod, ref)
        if fused and l == 0:
            continue
        dud, bd = _unpad(probe(T + 1, 2, l), B, S, Hl, Wl)
        _cmp(f"tangent block {l} du'", dud, torch.stack([tp["blocks"][l]["dud"] for tp in tp0]))
        assert bd == 0.0


# ---- whole meta-steps --------------------------------------------------------------------------------------
def _check_grads(names, got, ref):
    floor = max(0.02 * max(float(r.abs().max()) for r in ref), 1e-7)
    for n, a, r in zip(names, got, ref):
        e = rel_to_max(a.cpu(), r, floor)
        assert e <= GRAD_TOL, f"grad {n}: rel-to-max error {e:.3e}"


@pytest.mark.parametrize("T,first_order,need_grad", [(1, False, True), (3, False, True), (2, True, True), (0, False, True), (2, False, False)])
def test_maml_conv4_step_matches_the_float64_sweep(T, first_order, need_grad, dev, ws, c1mode):
    from fumi_amd import hip
    B, N, K, Q, Cin, H, W, nblk, alpha = 3, 5, 1, 3, 3, 16, 16, 4, 0.05
    ep, theta, Fd = _case(21 + T, B, N, K, Q, Cin, H, W, nblk)
    g = torch.Generator().manual_seed(9)
    p = theta + [torch.randn(N, Fd, generator=g) * 0.2, torch.randn(N, generator=g) * 0.1]
    head = torch.cat([p[-2], p[-1][:, None]], 1)
    out = hip.maml_conv4_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                              [_g(t, dev) for t in p], T, alpha, first_order, need_grad=need_grad)
    assert ws.read_status() == 0
    ms = _match_episodes(hip, ws, dev, out["logits"].cpu(), theta, head[None].expand(B, -1, -1), ep, T, alpha, first_order, need_grad)
    zq = torch.stack([m[0] for m in ms])
    assert rel_to_max(out["loss_b"].cpu(), torch.stack([m[1] for m in ms])) <= LOGIT_TOL
    mask = safe_margin_mask(zq, MARGIN)
    assert torch.equal(out["preds"].cpu()[mask], zq.max(-1)[1][mask]) and float(mask.float().mean()) > 0.9
    if bool(mask.all()):
        assert torch.allclose(out["acc_b"].cpu().double(), zq.max(-1)[1].eq(ep["y_q"]).double().mean(-1), atol=1e-6)
    if need_grad:                                       # mean over the episodes, back in the parameters' own layouts
        ref = [sum(m[2][i] for m in ms) / B for i in range(len(theta))]
        hb = sum(m[3] for m in ms) / B
        _check_grads([str(i) for i in range(len(p))], out["g_params"], ref + [hb[:, :-1], hb[:, -1]])
    # and plain autograd through the same network in fp32 agrees on what no tie can move: the logits
    pl = [t.clone().requires_grad_(True) for t in p]
    r32 = C.maml_conv4_meta_step(pl, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], T, alpha, first_order, need_grad=False)
    assert rel_to_max(out["logits"].cpu(), r32["logits"]) <= 10 * LOGIT_TOL


@pytest.mark.parametrize("T,tanh", [(1, False), (2, True)])
def test_fumi_conv4_step_matches_the_float64_sweep(T, tanh, dev, ws, c1mode):
    from fumi_amd import hip
    B, N, K, Q, Cin, H, W, nblk, Dt, Ht, alpha = 2, 5, 2, 3, 3, 20, 20, 4, 12, 16, 0.05
    ep, theta, Fd = _case(31 + T, B, N, K, Q, Cin, H, W, nblk)
    _, phi = cg.make_fumi_params(31, 8, [Fd], Dt, Ht, head_scale=0.3)
    stats = torch.zeros(2, device=dev)
    out = hip.fumi_conv4_step(ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                              [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, alpha, tanh, text_s=_g(ep["text_s"], dev),
                              stats=stats)
    assert ws.read_status() == 0
    # the hypernetwork in float64 (pinned part of the path: fumi.py:70-86,104-113,207-210), the encoder by the float64 sweep
    ph = [t.double().requires_grad_(True) for t in phi]
    heads = torch.stack([R.hyper_net(R.class_text_select(ep["text_s"][b].double(), ep["y_s"][b], N), ph, tanh) for b in range(B)])
    ms = _match_episodes(hip, ws, dev, out["logits"].cpu(), theta, heads.detach(), ep, T, alpha, False)
    zq = torch.stack([m[0] for m in ms])
    mask = safe_margin_mask(zq, MARGIN)
    assert torch.equal(out["preds"].cpu()[mask], zq.max(-1)[1][mask])
    loss = float(torch.stack([m[1] for m in ms]).mean())
    assert abs(float(stats[0]) - loss) <= LOGIT_TOL * max(1.0, loss)
    g_theta = [sum(m[2][i] for m in ms) / B for i in range(len(theta))]
    g_phi = torch.autograd.grad(heads, ph, grad_outputs=torch.stack([m[3] for m in ms]) / B)
    _check_grads([f"theta{i}" for i in range(len(theta))] + [f"phi{i}" for i in range(4)], out["g_theta"] + out["g_phi"],
                 g_theta + list(g_phi))


def test_conv4_full_size_episode_84x84(dev, ws):
    """BASELINE.json configs[1] as worded, one episode pair: 5-way 5-shot, 3 x 84 x 84 images, Conv4 (1600 features), 1 inner
    step, second-order meta-gradients (3 query images per class keep the host-side autograd oracle to seconds).
    Tolerance semantics: the checker is the FLOAT64 oracle.  At this size (2.8 M pooling windows and ReLUs per
    block-1 pass; gradients that are sums of 176 400 cancelling terms) two correct fp32 implementations differ from float64 --
    and from each other -- by up to a few 1e-3 of a tensor's scale in the early blocks' meta-gradients: a single arg-max or
    ReLU decision that falls the other way in fp32 moves a weight gradient by 1/sqrt(#terms) = 0.24 % (measured:
    tests/dev/probe_conv4_precision.py; the fp32 host oracle shows the same, 0.1-0.9 % per tensor, different tensors on different
    runs).  So logits / loss are held to 1e-4 and every gradient to max(1e-2, 4 x the fp32 host oracle's own distance from
    float64): what this test adds over the small cases (1e-3 on every gradient, 2e-4 on every intermediate) is the full-size
    geometry -- tiling, halos, 64-bit offsets -- whose mistakes are O(1), not O(1e-3)."""
    from fumi_amd import hip
    B, N, K, Q, Cin, H, W, nblk, Dt, Ht, alpha, T = 2, 5, 5, 3, 3, 84, 84, 4, 12, 24, 0.01, 1
    ep, theta, Fd = _case(77, B, N, K, Q, Cin, H, W, nblk)
    assert Fd == 1600
    _, phi = cg.make_fumi_params(77, 8, [Fd], Dt, Ht, head_scale=0.03)
    out = hip.fumi_conv4_step(ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                              [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, alpha, False, text_s=_g(ep["text_s"], dev))
    assert ws.read_status() == 0
    rg = lambda ts, dt: [t.to(dt).clone().requires_grad_(True) for t in ts]
    r32 = C.fumi_conv4_meta_step(rg(theta, torch.float32), rg(phi, torch.float32), ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"],
                                 ep["y_q"], N, T, alpha, False)
    ref = C.fumi_conv4_meta_step(rg(theta, torch.float64), rg(phi, torch.float64), ep["text_s"].double(), ep["x_s"].double(), ep["y_s"],
                                 ep["x_q"].double(), ep["y_q"], N, T, alpha, False)
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    mask = safe_margin_mask(ref["logits"], 1e-4)
    assert float(mask.float().mean()) > 0.9 and torch.equal(out["preds"].cpu()[mask], ref["preds"][mask])
    names = [f"theta{i}" for i in range(len(theta))] + [f"phi{i}" for i in range(4)]
    floor = 0.02 * max(float(r.abs().max()) for r in ref["g_theta"] + ref["g_phi"])
    for n, a, b32, b64 in zip(names, out["g_theta"] + out["g_phi"], r32["g_theta"] + r32["g_phi"], ref["g_theta"] + ref["g_phi"]):
        e, e32 = rel_to_max(a.cpu(), b64, floor), rel_to_max(b32, b64, floor)
        assert e <= max(1e-2, 4 * e32), f"grad {n}: {e:.3e} from float64 (fp32 host oracle: {e32:.3e})"


def test_conv4_as_worded_bench_shape_properties(dev, ws):
    """BASELINE.json configs[1] AS WORDED at bench.py's full shape (32 episodes, 5-way 5-shot, 32 query images per class, 3 x 84 x 84,
    1 inner step, second order): too large for the host oracle, so the size-independent properties of a meta-step are checked --
      * episodes are independent: the meta-batch's outputs are the two half-batches' outputs, its meta-gradient the mean of
        theirs (up to the fp32 summation order, which follows the batch size);
      * evaluation mode (need_grad = 0) computes the same logits / losses as the training step's forward;
      * the meta-gradient is linear in grad_scale;
      * the loss of every episode is the cross-entropy of its own logits (recomputed on the host in float64)."""
    from fumi_amd import hip
    B, N, K, Q, Cin, H, nblk, Dt, Ht, alpha, T = 32, 5, 5, 32, 3, 84, 4, 300, 256, 0.01, 1
    S, Qn = N * K, N * Q
    g = torch.Generator(device=dev).manual_seed(2024)
    cgen = torch.Generator().manual_seed(2024)
    y_s = torch.stack([torch.arange(N).repeat_interleave(K)[torch.randperm(S, generator=cgen)] for _ in range(B)]).to(dev)
    y_q = torch.stack([torch.arange(N).repeat_interleave(Q)[torch.randperm(Qn, generator=cgen)] for _ in range(B)]).to(dev)
    proto = torch.randn(B, N, Cin, H, H, device=dev, generator=g)                     # class means: the episodes are learnable
    x_s = proto.gather(1, y_s[:, :, None, None, None].expand(-1, -1, Cin, H, H)) + 0.5 * torch.randn(B, S, Cin, H, H, device=dev, generator=g)
    x_q = proto.gather(1, y_q[:, :, None, None, None].expand(-1, -1, Cin, H, H)) + 0.5 * torch.randn(B, Qn, Cin, H, H, device=dev, generator=g)
    cls_text = torch.randn(B, N, Dt, device=dev, generator=g)
    _, theta, Fd = _case(78, 1, N, 1, 1, Cin, 12, 12, nblk)
    assert hip.conv4_feature_dim(nblk, H, H) == 1600
    _, phi = cg.make_fumi_params(78, 8, [1600], Dt, Ht, head_scale=0.03)
    theta, phi = [_g(t, dev) for t in theta], [_g(t, dev) for t in phi]

    def run(sl, **kw):
        o = hip.fumi_conv4_step(ws, N, x_s[sl].contiguous(), y_s[sl].contiguous(), x_q[sl].contiguous(), y_q[sl].contiguous(), theta, phi,
                                T, alpha, False, cls_text=cls_text[sl].contiguous(), **kw)
        assert ws.read_status() == 0
        return {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in o.items() if v is not None}
    full = run(slice(0, B))
    lo, hi = run(slice(0, B // 2)), run(slice(B // 2, B))
    # tiles, pixel slabs and split-K groups depend on the batch size, so the fp32 sums run in another order: at this size (2.8 M
    # pooling windows and ReLUs per block-1 pass) that moves logits by ~2e-4 and re-routes a few arg-max / ReLU decisions (the
    # full-size test above states the same for two fp32 implementations): forward 1e-3, gradients 2e-2 of each tensor's scale
    halves = {k: torch.cat([lo[k], hi[k]]) for k in ("logits", "loss_b", "acc_b", "preds")}
    assert rel_to_max(full["logits"].cpu(), halves["logits"].cpu()) <= 1e-3 and rel_to_max(full["loss_b"].cpu(), halves["loss_b"].cpu()) <= 1e-3
    safe = safe_margin_mask(full["logits"].double().cpu(), 1e-2)
    assert float(safe.float().mean()) > 0.8 and torch.equal(full["preds"].cpu()[safe], halves["preds"].cpu()[safe])
    names = [f"theta{i}" for i in range(len(theta))] + [f"phi{i}" for i in range(4)]
    # (a tensor whose gradient cancels to fp32 noise -- the head's bias row sums to zero over the classes -- is held to the scale of
    # the largest gradient, not to its own)
    floor = 1e-3 * max(float(t.abs().max()) for t in full["g_theta"] + full["g_phi"])
    for n, a, b0, b1 in zip(names, full["g_theta"] + full["g_phi"], lo["g_theta"] + lo["g_phi"], hi["g_theta"] + hi["g_phi"]):
        assert torch.isfinite(a).all(), n
        assert rel_to_max(a.cpu(), (0.5 * (b0 + b1)).cpu(), floor) <= 2e-2, n
    ev = run(slice(0, B), need_grad=False)                              # same batch size, same forward kernels
    assert rel_to_max(ev["logits"].cpu(), full["logits"].cpu()) <= 1e-5 and rel_to_max(ev["loss_b"].cpu(), full["loss_b"].cpu()) <= 1e-5
    sc = run(slice(0, B), grad_scale=3.0 / B)
    for n, a, b in zip(names, sc["g_theta"] + sc["g_phi"], full["g_theta"] + full["g_phi"]):
        assert rel_to_max(a.cpu(), 3.0 * b.cpu(), 3.0 * floor) <= 1e-5, n
    z = full["logits"].double().cpu()
    ce = torch.stack([F.cross_entropy(z[b], y_q[b].cpu()) for b in range(B)])
    assert rel_to_max(full["loss_b"].cpu().double(), ce) <= 1e-5
    acc = (z.max(-1)[1] == y_q.cpu()).double().mean(-1)
    mask = safe_margin_mask(z, MARGIN).all(-1)
    assert torch.allclose(full["acc_b"].cpu().double()[mask], acc[mask], atol=1e-6)
    assert float(acc.mean()) > 1.5 / N                  # one inner step on separable classes: clearly above chance


def test_lanes_give_the_single_stream_step(dev, ws):
    """The parts of a meta-batch run on concurrent streams (lanes, csrc/conv4.hip) -- with one, two and three lanes the same meta-step
    must come out: per-episode outputs to fp32 summation order (a lane of 4 episodes tiles its maps differently from a batch of 12),
    meta-gradients to 1e-3 of the largest gradient."""
    from fumi_amd import hip
    B, N, K, Q, Cin, H, W, nblk, Dt, Ht, alpha, T = 12, 5, 2, 3, 3, 20, 20, 4, 12, 16, 0.05, 2
    ep, theta, Fd = _case(91, B, N, K, Q, Cin, H, W, nblk)
    _, phi = cg.make_fumi_params(91, 8, [Fd], Dt, Ht, head_scale=0.3)
    th, ph = [_g(t, dev) for t in theta], [_g(t, dev) for t in phi]
    args = (ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev), th, ph, T, alpha, True)
    outs = []
    try:
        for lanes in (1, 2, 3):
            hip.conv4_set_option(1, lanes)
            o = hip.fumi_conv4_step(*args, text_s=_g(ep["text_s"], dev))
            assert ws.read_status() == 0
            outs.append({k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in o.items() if v is not None})
    finally:
        hip.conv4_set_option(1, 0)
    one = outs[0]
    gmax = max(float(t.abs().max()) for t in one["g_theta"] + one["g_phi"])
    for o in outs[1:]:
        assert rel_to_max(o["logits"].cpu(), one["logits"].cpu()) <= 1e-4 and rel_to_max(o["loss_b"].cpu(), one["loss_b"].cpu()) <= 1e-4
        for a, b in zip(o["g_theta"] + o["g_phi"], one["g_theta"] + one["g_phi"]):
            assert rel_to_max(a.cpu(), b.cpu(), 1e-1 * gmax) <= 1e-2


# ---- the module surface (--im_encoder conv4) on the GPU -------------------------------------------------------------------
def test_conv4_features_op(dev, ws):
    from fumi_amd import hip
    ep, theta, Fd = _case(5, 3, 4, 2, 3, 3, 28, 28, 4)
    f = hip.conv4_features(ws, _g(ep["x_q"], dev), [_g(t, dev) for t in theta]).cpu()
    ref = torch.stack([C.conv4_features(ep["x_q"][b], theta) for b in range(3)])
    assert f.shape == (3, 12, Fd) and rel_to_max(f, ref) <= 1e-5


def test_fumi_conv4_evaluate_on_gpu_equals_the_cpu_oracle_engine(dev):
    """FUMI(im_encoder='conv4').evaluate: one training step (meta-gradients -> optimizer) and a test step on the HIP engine
    against the same model driven by the autograd oracle on the host (tests/oracle_engine.py).  This checks the module plumbing
    (parameter order, gradient views, head width, loss / accuracy read-back); kernel parity is what the tests above are for, so
    the update is compared at 5 % of its size: a ReLU / arg-max decision within fp32 noise of a tie legitimately moves a conv
    tensor's gradient by a percent or two (see _match_episodes), and longer trajectories amplify that without bound."""
    from types import SimpleNamespace
    from fumi_amd import engine
    from fumi_amd.models.fumi import FUMI
    from oracle_engine import OracleEngine
    ep = C.make_image_episodes(8, 4, 5, 2, 3, 3, 20, 20, 12)
    batch = cg.to_batch(ep)

    def run(device, eng):
        old = engine.set_engine(eng)
        try:
            torch.manual_seed(1)
            m = FUMI(n_way=5, im_encoder="conv4", image_size=20, text_emb_dim=12, text_hid_dim=16, norm_hypernet=True).to(device)
            args = SimpleNamespace(device=device, num_train_adapt_steps=2, num_test_adapt_steps=2, step_size=0.05,
                                   first_order=False, num_ways=5, batch_size=4)
            opt = torch.optim.SGD(m.parameters(), lr=0.02)
            p0 = torch.cat([p.detach().reshape(-1).cpu() for p in m.parameters()])
            tr = m.evaluate(args, batch, opt, "train")
            p1 = torch.cat([p.detach().reshape(-1).cpu() for p in m.parameters()])
            te = m.evaluate(args, batch, None, "test")
            return [float(tr[0]), float(tr[1]), float(te[0])], p1 - p0, te[2].cpu()
        finally:
            engine.set_engine(old)
    l_gpu, d_gpu, pr_gpu = run(dev, None)
    l_cpu, d_cpu, pr_cpu = run(torch.device("cpu"), OracleEngine())
    assert abs(l_gpu[0] - l_cpu[0]) < 1e-4 and abs(l_gpu[1] - l_cpu[1]) < 1e-6        # loss / accuracy of the first meta-batch
    assert float((d_gpu - d_cpu).abs().max()) <= 0.05 * float(d_cpu.abs().max())       # the SGD update = -lr * meta-gradient
    assert abs(l_gpu[2] - l_cpu[2]) < 5e-3
    assert float((pr_gpu == pr_cpu).float().mean()) > 0.9


def test_cli_fumi_conv4_end_to_end_on_gpu(dev, tmp_path, monkeypatch):
    """`python -m fumi_amd.main --model fumi --im_encoder conv4 --dataset synthetic` (BASELINE.json configs[1] as worded, shortened):
    image loader -> Conv4 + hypernetwork -> meta-training on the HIP engine -> checkpoint -> test."""
    from fumi_amd import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ["--model", "fumi", "--dataset", "synthetic", "--im_encoder", "conv4", "--image_size", "28", "--text_encoder", "BERT",
            "--text_emb_dim", "32", "--batch_size", "8", "--num_shots", "5", "--num_ways", "5", "--num_shots_test", "5",
            "--epochs", "30", "--eval_freq", "15", "--num_ep_test", "16", "--num_train_adapt_steps", "1",
            "--num_test_adapt_steps", "1", "--lr", "1e-3", "--step_size", "0.05", "--dropout", "0", "--log_dir", str(tmp_path / "res"),
            "--synthetic_classes", "16", "--wandb_offline"]
    args = cli.parse_args(argv)
    assert args.device.type == "cuda"
    res = cli.main(args)
    assert np.isfinite(res["test_loss"]) and 0.0 <= res["test_acc"] <= 1.0
    assert res["test_acc"] > 0.3                                        # chance = 0.2: the engine's gradients train the encoder


# ---- AM3 with the Conv4 backbone at the image_encoder seam (BASELINE.json configs[3] as worded) ------------------------------
@pytest.mark.parametrize("lamda_fixed,Ht", [(None, 64), (1, 24)])
def test_am3_conv4_step_matches_autograd(lamda_fixed, Ht, dev):
    """encode (tape kept in the encoder's own workspace) -> fumi_hip_am3_step_dx -> encode_bwd against float64 autograd through
    oracle/conv4_ref.am3_conv4_step: features, loss, predictions, the ten AM3 gradients and the twelve backbone gradients."""
    from fumi_amd import hip
    B, N, K, Q, Cin, H, W, Dt, P = 3, 4, 2, 3, 3, 20, 20, 12, 32
    ep, theta, Fd = _case(21, B, N, K, Q, Cin, H, W, 4)
    w = cg.make_am3_params(21, Fd, Dt, Ht, P)
    wl = [w[k] for k in hip.AM3_KEYS]
    ws_main, ws_enc = hip.Workspace.get(dev), hip.Workspace.get(dev, "encoder")
    th_d = [_g(t, dev) for t in theta]
    xs, xq = _g(ep["x_s"], dev), _g(ep["x_q"], dev)
    f_s, f_q = hip.conv4_encode(ws_enc, xs, xq, th_d, keep_tape=True)
    out = hip.am3_step(ws_main, f_s, _g(ep["y_s"], dev), f_q, _g(ep["y_q"], dev), _g(ep["text_s"], dev), [_g(t, dev) for t in wl],
                       N, lamda_fixed, want_dx=True)
    g_th = hip.conv4_encode_bwd(ws_enc, xs, xq, out["dx_s"], out["dx_q"], th_d)
    assert ws_main.read_status() == 0
    d64 = lambda t: t.double()
    ref = C.am3_conv4_step([d64(t) for t in theta], {k: d64(v) for k, v in w.items()}, d64(ep["text_s"]), d64(ep["x_s"]), ep["y_s"],
                           d64(ep["x_q"]), ep["y_q"], N, lamda_fixed)
    assert rel_to_max(f_s.cpu().double(), ref["feats_s"]) <= 1e-5 and rel_to_max(f_q.cpu().double(), ref["feats_q"]) <= 1e-5
    assert abs(float(out["loss"]) - float(ref["loss"])) <= LOGIT_TOL * max(1.0, abs(float(ref["loss"])))
    d2 = ref["dist"].transpose(1, 2).topk(2, dim=-1, largest=False)[0]
    mask = (d2[..., 1] - d2[..., 0]) > 1e-4 * d2[..., 1].abs().clamp_min(1.0)
    assert torch.equal(out["preds"].cpu()[mask], ref["preds"][mask])
    for k, g in zip(hip.AM3_KEYS, out["grads"]):
        if float(ref["grads"][k].abs().max()) < 1e-9:      # identically zero (the image bias cancels in every distance when lamda = 1)
            assert float(g.abs().max()) < 1e-6, k
        else:
            assert rel_to_max(g.cpu().double(), ref["grads"][k]) <= GRAD_TOL, k
    names = [f"block{i}.{p}" for i in range(4) for p in ("W", "g", "b")]
    for nme, g, r in zip(names, g_th, ref["grads_theta"]):
        # a ReLU / arg-max decision within round-off of a tie moves an early block's gradient by a percent (the header of
        # test_conv4_full_size_episode_84x84): 1e-2 of the tensor's scale
        assert rel_to_max(g.cpu().double(), r) <= 1e-2, (nme, rel_to_max(g.cpu().double(), r))
    # the tape is consumed: a second backward without a new encode is refused
    with pytest.raises(hip.FumiHipError):
        hip.conv4_encode_bwd(ws_enc, xs, xq, out["dx_s"], out["dx_q"], th_d)


def test_cli_am3_conv4_end_to_end_on_gpu(dev, tmp_path, monkeypatch):
    """`python -m fumi_amd.main --model am3 --im_encoder conv4 --dataset synthetic` (BASELINE.json configs[3] as worded, shortened)."""
    from fumi_amd import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ["--model", "am3", "--dataset", "synthetic", "--im_encoder", "conv4", "--image_size", "28", "--text_encoder", "BERT",
            "--text_emb_dim", "32", "--batch_size", "8", "--num_shots", "5", "--num_ways", "5", "--num_shots_test", "5",
            "--epochs", "30", "--eval_freq", "15", "--num_ep_test", "16", "--lr", "1e-3", "--dropout", "0.25",
            "--log_dir", str(tmp_path / "res"), "--synthetic_classes", "16", "--wandb_offline"]
    args = cli.parse_args(argv)
    assert args.device.type == "cuda"
    res = cli.main(args)
    assert np.isfinite(res["test_loss"]) and 0.0 <= res["test_acc"] <= 1.0
    assert res["test_acc"] > 0.3                                        # chance = 0.2
