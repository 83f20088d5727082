"""CPU oracle of the Conv4 image encoder at the ``im_net`` seam -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the convolutional part: the reference (s-a-malik/fumi) has only the seam -- ``im_net`` is "any module
with forward(x, params) and meta_named_parameters()" (fumi/models/fumi.py:89-100; ``--im_encoder resnet`` is a ``# TODO``
at fumi/models/am3.py:41-46) -- and BASELINE.json's configs are worded with a Conv4 encoder on 84x84 images that the
reference never implements.  There is nothing to import, no golden vector, no fixture: this file restates the standard
few-shot Conv4 (four blocks of conv3x3(64, pad 1) . BatchNorm2d with BATCH statistics (torchmeta's
``MetaBatchNorm2d(momentum=1., track_running_stats=False)``: batch statistics in training AND evaluation) . ReLU .
MaxPool2d(2)) with ``torch.nn.functional`` ops and lets autograd differentiate it (``create_graph=True`` for the second-order
meta-gradient).  Everything from the feature vector onwards -- hypernetwork head, inner SGD update of (theta, head), query
cross-entropy, arg-max -- is the reference's algorithm (fumi/models/fumi.py:146-192; maml.py:156-191) and shares the
pinned restatement in ``oracle/fumi_ref.py``.

Conv layers carry no bias: batch-statistic normalisation removes any per-channel constant, so a conv bias has no effect on
the output and an exactly zero gradient.

theta = [W1, g1, b1, W2, g2, b2, W3, g3, b3, W4, g4, b4]   W_l [C, C_in, 3, 3], g_l = BN weight [C], b_l = BN bias [C].
Images are NCHW; features are ``x.view(M, -1)`` of the last block's [M, C, h, w] output (PyTorch's flatten order).
All functions are dtype-generic (float64 for a high-precision oracle).
"""
import torch
import torch.nn.functional as F

from . import fumi_ref as R

BN_EPS = 1e-5


def conv4_features(x, theta):
    """x [M, Cin, H, W] -> [M, C * h * w].  One block = conv3x3 pad 1 (no bias), batch-stat BN, ReLU, max-pool 2 (floor)."""
    for i in range(0, len(theta), 3):
        x = F.conv2d(x, theta[i], None, padding=1)
        x = F.batch_norm(x, None, None, theta[i + 1], theta[i + 2], training=True, momentum=1.0, eps=BN_EPS)
        x = F.max_pool2d(F.relu(x), 2)
    return x.reshape(x.shape[0], -1)


def feature_dim(H, W, C, n_blocks=4):
    for _ in range(n_blocks):
        H, W = H // 2, W // 2
    return C * H * W


def conv4_forward(x, theta, h):
    """logits = features @ h[:, :-1].T + h[:, -1]   (fumi.py:214-218 with the Conv4 ``im_net``)."""
    return conv4_features(x, theta) @ h[:, :-1].t() + h[:, -1]


def episode(theta, h, x_s, y_s, x_q, T, alpha, first_order=False):
    """Inner loop of one episode on (theta, head) and the query logits with the adapted parameters (graph kept):
    fumi.py:160-178 / maml.py:166-176 (torchmeta gradient_update_parameters: p - step_size * grad)."""
    th = list(theta)
    for _ in range(T):
        inner = F.cross_entropy(conv4_forward(x_s, th, h), y_s)
        grads = torch.autograd.grad(inner, [h] + th, create_graph=not first_order)
        h = h - alpha * grads[0]
        th = [p - alpha * g for p, g in zip(th, grads[1:])]
    return conv4_forward(x_q, th, h)


def fumi_conv4_meta_step(theta, phi, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head, need_grad=True,
                         first_order=False, extra=None):
    """FuMI meta-step (fumi.py:115-196) with the Conv4 encoder: x_s [B,S,Cin,H,W], x_q [B,Qn,Cin,H,W]; the hypernetwork emits
    [N, F+1] head rows (F = Conv4 feature width).  Returns what fumi_ref.fumi_meta_step returns."""
    B = x_s.shape[0]
    logits, loss_b = [], []
    for b in range(B):
        c = R.class_text_select(text_s[b], y_s[b], n_way)
        h = R.hyper_net(c, phi, tanh_head)
        lq = episode(theta, h, x_s[b], y_s[b], x_q[b], T, alpha, first_order)
        logits.append(lq)
        loss_b.append(F.cross_entropy(lq, y_q[b]))
    loss = torch.stack(loss_b).sum() / B
    out = R._pack(logits, loss_b, y_q)
    out["loss"] = loss.detach()
    if need_grad:
        ps = list(theta) + list(phi) + list(extra or [])            # extra: tensors text_s depends on (fumi_ref.fumi_meta_step)
        g = torch.autograd.grad(loss, ps, allow_unused=True)
        g = [torch.zeros_like(p) if gi is None else gi for gi, p in zip(g, ps)]
        out["g_theta"], out["g_phi"] = g[:len(theta)], g[len(theta):len(theta) + len(phi)]
        out["g_extra"] = g[len(theta) + len(phi):]
    return out


def maml_conv4_meta_step(params, x_s, y_s, x_q, y_q, T, alpha, first_order=False, need_grad=True):
    """MAML meta-step (maml.py:134-193) with the Conv4 encoder: params = theta (12 tensors) + [lin_final W [N,F], b [N]]."""
    theta, Wf, bf = list(params[:-2]), params[-2], params[-1]
    B = x_s.shape[0]
    logits, loss_b = [], []
    for b in range(B):
        h = torch.cat([Wf, bf[:, None]], 1)
        lq = episode(theta, h, x_s[b], y_s[b], x_q[b], T, alpha, first_order)
        logits.append(lq)
        loss_b.append(F.cross_entropy(lq, y_q[b]))
    loss = torch.stack(loss_b).sum() / B
    out = R._pack(logits, loss_b, y_q)
    out["loss"] = loss.detach()
    if need_grad:
        g = torch.autograd.grad(loss, list(params), allow_unused=True)
        out["g_params"] = [torch.zeros_like(p) if gi is None else gi for gi, p in zip(g, params)]
    return out


def am3_conv4_step(theta, w, text_s, x_s, y_s, x_q, y_q, n_way, lamda_fixed=None, need_grad=True, masks=None):
    """AM3 (am3.py:128-212 through oracle/fumi_ref.am3_step) with the Conv4 backbone in front of ``image_encoder`` (the seam at
    am3.py:41-46): x_s [B,S,C,H,W], x_q [B,Qn,C,H,W]; every episode's support set and query set is one batch-statistics group.
    Returns fumi_ref.am3_step's dict with ``grads_theta`` (list like theta) added."""
    B = x_s.shape[0]
    th = [t.detach().clone().requires_grad_(need_grad) for t in theta]
    f_s = torch.stack([conv4_features(x_s[b], th) for b in range(B)])
    f_q = torch.stack([conv4_features(x_q[b], th) for b in range(B)])
    wl = {k: v.detach().clone().requires_grad_(need_grad) for k, v in w.items()}
    out = R.am3_step(wl, text_s, f_s, y_s, f_q, y_q, n_way, lamda_fixed=lamda_fixed, need_grad=False, masks=masks)
    out["feats_s"], out["feats_q"] = f_s.detach(), f_q.detach()
    if need_grad:
        # am3_step detaches its loss: restate the loss on the graph that includes the backbone
        im_s = F.linear(f_s, wl["Wi"], wl["bi"]); im_q = F.linear(f_q, wl["Wi"], wl["bi"])
        t1 = torch.relu(F.linear(text_s, wl["G0"], wl["g0"]))
        if masks is not None:
            t1 = t1 * masks[0].view_as(t1)
        tx = F.linear(t1, wl["G1"], wl["g1"])
        l1 = torch.relu(F.linear(tx, wl["H0"], wl["h0"]))
        if masks is not None:
            l1 = l1 * masks[1].view_as(l1)
        lam = torch.sigmoid(F.linear(l1, wl["H1"], wl["h1"]))
        if lamda_fixed == 0:
            lam = torch.zeros_like(lam)
        elif lamda_fixed == 1:
            lam = torch.ones_like(lam)
        loss = R.prototypical_loss(R.get_prototypes(im_s, tx, lam, y_s, n_way), im_q, y_q)
        names = ["Wi", "bi", "G0", "g0", "G1", "g1", "H0", "h0", "H1", "h1"]
        g = torch.autograd.grad(loss, [wl[k] for k in names] + th, allow_unused=True)
        out["grads"] = {k: (torch.zeros_like(wl[k]) if gi is None else gi) for k, gi in zip(names, g[:10])}
        out["grads_theta"] = [torch.zeros_like(t) if gi is None else gi for t, gi in zip(th, g[10:])]
    return out


def make_conv4_params(seed, Cin=3, C=64, n_blocks=4, dtype=torch.float32):
    """Deterministic parameters for the parity cases (uniform +-1/sqrt(fan_in) conv weights like nn.Conv2d's default scale,
    BN weight around 1, BN bias around 0 -- not exactly 1 / 0 so their gradients' paths are exercised)."""
    import numpy as np
    rs = np.random.RandomState(seed + 15485863)
    theta, ci = [], Cin
    for _ in range(n_blocks):
        bound = 1.0 / np.sqrt(ci * 9)
        theta.append(torch.from_numpy(rs.uniform(-bound, bound, (C, ci, 3, 3))).to(dtype))
        theta.append(torch.from_numpy(1.0 + 0.1 * rs.standard_normal(C)).to(dtype))
        theta.append(torch.from_numpy(0.1 * rs.standard_normal(C)).to(dtype))
        ci = C
    return theta


def make_image_episodes(seed, B, N, K, Q, Cin, H, W, Dt, learnable=True):
    """Synthetic image episodes in the loader's batch contract (images in place of embeddings): class-dependent low-frequency
    pattern + noise, per-class text rows, shuffled balanced labels."""
    import numpy as np
    from . import casegen as cg
    rs = np.random.RandomState(seed)
    S, Qn = N * K, N * Q
    y_s = cg.make_targets(rs, B, N, K, False)
    y_q = cg.make_targets(rs, B, N, Q, False)
    mu = rs.standard_normal((B, N, Cin, H, W)) if learnable else np.zeros((B, N, Cin, H, W))
    noise = 1.0 if learnable else 1.0
    x_s = np.take_along_axis(mu, y_s[:, :, None, None, None], 1) * 0.5 + noise * rs.standard_normal((B, S, Cin, H, W))
    x_q = np.take_along_axis(mu, y_q[:, :, None, None, None], 1) * 0.5 + noise * rs.standard_normal((B, Qn, Cin, H, W))
    cls_text = rs.standard_normal((B, N, Dt))
    text_s = np.take_along_axis(cls_text, y_s[..., None], 1)
    text_q = np.take_along_axis(cls_text, y_q[..., None], 1)
    t = lambda a, d=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    return dict(x_s=t(x_s), y_s=t(y_s, torch.int64), x_q=t(x_q), y_q=t(y_q, torch.int64), text_s=t(text_s), text_q=t(text_q),
                idx_s=torch.arange(B * S).view(B, S), idx_q=torch.arange(B * Qn).view(B, Qn) + B * S)
