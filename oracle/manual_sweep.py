"""Hand-derived, autograd-free statement of ONE episode in the exact algebraic form the HIP kernels use --
TEST INFRASTRUCTURE ONLY (it is the CPU blueprint/cross-check of ``fumi_amd/csrc``; never shipped in the product path).

Why a second oracle: the HIP path does not differentiate a graph, it runs a hand-written forward tape and reverse
sweep, and it never materialises the per-episode fast weight of layer 0.  With dz0_t the support pre-activation
gradient of layer 0 at inner step t,

    W0_t = W0 - alpha * D_t^T Xs,   D_t = sum_{tau<t} dz0_tau  in R^{S x h0},     b0_t = b0 - alpha * colsum(D_t)
    z0_t(X) = X W0^T  - alpha * (X Xs^T) D_t + b0_t  =  A0(X) - alpha * G(X) D_t + b0_t

so the whole meta-step needs X only in two shared GEMMs (A0|G = X [W0;Xs]^T forward, gW0 = Abar0^T X backward) and the
state of layer 0 is the small matrix D_t.  This file checks that algebra (and the second-order reverse sweep through
softmax-CE, ReLU masks and the SGD updates) against ``oracle/fumi_ref.py``'s autograd (tests/test_manual_sweep.py).

Reference lines restated: fumi/models/fumi.py:156-193 (episode), fumi/models/maml.py:158-191.
"""
import torch


def _softmax(l):
    return torch.softmax(l, dim=-1)


def episode_manual(theta, h0, x_s, y_s, x_q, y_q, T, alpha, second_order=True, need_grad=True):
    """theta = [W0,b0,...,W_{L-1},b_{L-1}] (L>=1, each followed by ReLU); h0 [N,H+1] = initial head [Wh | bh].
    Returns dict(logits, loss, A0bar_s [S,h0], A0bar_q [Qn,h0], g_b0, g_W[i>=1], g_b[i>=1], g_h0 [N,H+1]):
    gradients of THIS episode's query loss w.r.t. the meta-parameters (g_W0 = A0bar_s^T Xs + A0bar_q^T Xq)."""
    L = len(theta) // 2
    W = [theta[2 * i] for i in range(L)]
    b = [theta[2 * i + 1] for i in range(L)]
    S, Qn, N = x_s.shape[0], x_q.shape[0], h0.shape[0]
    dt = x_s.dtype
    Y = torch.zeros(S, N, dtype=dt); Y[torch.arange(S), y_s] = 1
    Yq = torch.zeros(Qn, N, dtype=dt); Yq[torch.arange(Qn), y_q] = 1

    # ---- shared GEMM 1 (per-episode slice): A0 = X W0^T, G = X Xs^T
    A0s, A0q = x_s @ W[0].t(), x_q @ W[0].t()
    Gss, Gqs = x_s @ x_s.t(), x_q @ x_s.t()

    # ---- adapt: T inner steps on the support set, tape kept
    D = torch.zeros(S, W[0].shape[0], dtype=dt)
    Wt = [None] + [W[i].clone() for i in range(1, L)]
    bt = [None] + [b[i].clone() for i in range(1, L)]
    Wh, bh = h0[:, :-1].clone(), h0[:, -1].clone()
    tape = []
    for _ in range(T):
        a = [None] * L
        a[0] = torch.relu(A0s - alpha * (Gss @ D) + (b[0] - alpha * D.sum(0)))
        for i in range(1, L):
            a[i] = torch.relu(a[i - 1] @ Wt[i].t() + bt[i])
        p = _softmax(a[L - 1] @ Wh.t() + bh)
        e = (p - Y) / S
        gWh, gbh = e.t() @ a[L - 1], e.sum(0)
        da = e @ Wh
        dz, gW, gb = [None] * L, [None] * L, [None] * L
        for i in range(L - 1, 0, -1):
            dz[i] = da * (a[i] > 0)
            gW[i], gb[i] = dz[i].t() @ a[i - 1], dz[i].sum(0)
            da = dz[i] @ Wt[i]
        dz[0] = da * (a[0] > 0)
        tape.append(dict(a=a, p=p, e=e, dz=dz, W=[w.clone() if w is not None else None for w in Wt], Wh=Wh.clone()))
        Wh, bh = Wh - alpha * gWh, bh - alpha * gbh
        for i in range(1, L):
            Wt[i], bt[i] = Wt[i] - alpha * gW[i], bt[i] - alpha * gb[i]
        D = D + dz[0]

    # ---- query forward with (theta_T, h_T)
    aq = [None] * L
    aq[0] = torch.relu(A0q - alpha * (Gqs @ D) + (b[0] - alpha * D.sum(0)))
    for i in range(1, L):
        aq[i] = torch.relu(aq[i - 1] @ Wt[i].t() + bt[i])
    logits = aq[L - 1] @ Wh.t() + bh
    logp = torch.log_softmax(logits, -1)
    loss = -(logp * Yq).sum() / Qn
    out = dict(logits=logits, loss=loss)
    if not need_grad:
        return out

    # ---- query backward (first-order through the query graph)
    lbar = (_softmax(logits) - Yq) / Qn
    Whb, bhb = lbar.t() @ aq[L - 1], lbar.sum(0)
    ab = lbar @ Wh
    Wb, bb = [None] * L, [None] * L
    for i in range(L - 1, 0, -1):
        zb = ab * (aq[i] > 0)
        Wb[i], bb[i] = zb.t() @ aq[i - 1], zb.sum(0)
        ab = zb @ Wt[i]
    z0b = ab * (aq[0] > 0)
    A0bar_q = z0b
    b0bar = z0b.sum(0)
    Db = -alpha * (Gqs.t() @ z0b + z0b.sum(0)[None, :])
    A0bar_s = torch.zeros_like(A0s)

    # ---- reverse sweep over the inner steps (second order)
    if second_order:
        for t in range(T - 1, -1, -1):
            tp = tape[t]
            a, p, e, dz, Wc, Whc = tp["a"], tp["p"], tp["e"], tp["dz"], tp["W"], tp["Wh"]
            gWhb, gbhb = -alpha * Whb, -alpha * bhb
            gWb = [None] + [-alpha * Wb[i] for i in range(1, L)]
            gbb = [None] + [-alpha * bb[i] for i in range(1, L)]
            abar = [torch.zeros_like(a[i]) for i in range(L)]
            # reverse of the backward pass
            dab = Db * (a[0] > 0)                                   # d(abar) of da_0   (dz0bar = Dbar')
            for i in range(1, L):
                dzb = dab @ Wc[i].t()
                Wb[i] = Wb[i] + dz[i].t() @ dab
                dzb = dzb + gbb[i][None, :] + a[i - 1] @ gWb[i].t()
                abar[i - 1] = abar[i - 1] + dz[i] @ gWb[i]
                dab = dzb * (a[i] > 0)
            eb = dab @ Whc.t() + gbhb[None, :] + a[L - 1] @ gWhb.t()
            Whb = Whb + e.t() @ dab
            abar[L - 1] = abar[L - 1] + e @ gWhb
            pb = eb / S
            lb = p * (pb - (p * pb).sum(-1, keepdim=True))
            # reverse of the forward pass
            abar[L - 1] = abar[L - 1] + lb @ Whc
            Whb = Whb + lb.t() @ a[L - 1]
            bhb = bhb + lb.sum(0)
            for i in range(L - 1, 0, -1):
                zb = abar[i] * (a[i] > 0)
                abar[i - 1] = abar[i - 1] + zb @ Wc[i]
                Wb[i] = Wb[i] + zb.t() @ a[i - 1]
                bb[i] = bb[i] + zb.sum(0)
            z0b = abar[0] * (a[0] > 0)
            A0bar_s = A0bar_s + z0b
            b0bar = b0bar + z0b.sum(0)
            Db = Db - alpha * (Gss @ z0b + z0b.sum(0)[None, :])
    out.update(A0bar_s=A0bar_s, A0bar_q=A0bar_q, g_b0=b0bar, g_W=Wb, g_b=bb,
               g_h0=torch.cat([Whb, bhb[:, None]], 1))
    return out


def hyper_net_manual(c, phi, tanh_head):
    """Forward of the hypernetwork keeping what its backward needs."""
    A0, a0, A1, a1 = phi
    u = torch.relu(c @ A0.t() + a0)
    hp = u @ A1.t() + a1
    return (torch.tanh(hp) if tanh_head else hp), u


def hyper_net_backward(c, u, h, hbar, phi, tanh_head):
    """Grads of (A0,a0,A1,a1) given hbar = dL/dh for rows c [R,Dt] (R = B*N rows when batched)."""
    A0, a0, A1, a1 = phi
    hpb = hbar * (1 - h * h) if tanh_head else hbar
    gA1, ga1 = hpb.t() @ u, hpb.sum(0)
    ub = (hpb @ A1) * (u > 0)
    return [ub.t() @ c, ub.sum(0), gA1, ga1]


def fumi_meta_step_manual(theta, phi, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head):
    """Whole FuMI meta-step in kernel form; returns the same dict as fumi_ref.fumi_meta_step (mean-loss gradients)."""
    from .fumi_ref import class_text_select
    B = x_s.shape[0]
    c = torch.stack([class_text_select(text_s[b], y_s[b], n_way) for b in range(B)])       # [B,N,Dt]
    h, u = hyper_net_manual(c.reshape(B * n_way, -1), phi, tanh_head)
    h = h.reshape(B, n_way, -1)
    L = len(theta) // 2
    gW0 = torch.zeros_like(theta[0]); gb0 = torch.zeros_like(theta[1])
    gW = [None] + [torch.zeros_like(theta[2 * i]) for i in range(1, L)]
    gb = [None] + [torch.zeros_like(theta[2 * i + 1]) for i in range(1, L)]
    hbar = torch.zeros_like(h)
    logits, losses = [], []
    for b in range(B):
        o = episode_manual(theta, h[b], x_s[b], y_s[b], x_q[b], y_q[b], T, alpha)
        logits.append(o["logits"]); losses.append(o["loss"])
        gW0 += o["A0bar_s"].t() @ x_s[b] + o["A0bar_q"].t() @ x_q[b]
        gb0 += o["g_b0"]
        for i in range(1, L):
            gW[i] += o["g_W"][i]; gb[i] += o["g_b"][i]
        hbar[b] = o["g_h0"]
    g_phi = hyper_net_backward(c.reshape(B * n_way, -1), u, h.reshape(B * n_way, -1), hbar.reshape(B * n_way, -1), phi, tanh_head)
    g_theta = [gW0 / B, gb0 / B]
    for i in range(1, L):
        g_theta += [gW[i] / B, gb[i] / B]
    logits = torch.stack(logits)
    return dict(logits=logits, loss=torch.stack(losses).mean(), preds=logits.max(-1)[1],
                g_theta=g_theta, g_phi=[g / B for g in g_phi])
