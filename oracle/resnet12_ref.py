"""CPU oracle of the ResNet-12 image encoder at the ``im_net`` seam (BASELINE.json configs[4]) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the convolutional part: the reference (s-a-malik/fumi) has only the seam -- ``im_net`` is "any module with
forward(x, params) and meta_named_parameters()" (fumi/models/fumi.py:89-100; ``--im_encoder resnet`` is a ``# TODO`` at
fumi/models/am3.py:41-46) -- and no ResNet-12, no bf16, no image input anywhere.  There is nothing to import, no golden vector, no
fixture.  This file restates the ResNet-12 of the few-shot literature (TADAM / MetaOptNet form, no DropBlock) with
``torch.nn.functional`` ops and lets autograd differentiate it (``create_graph=True`` for the second-order meta-gradient):

    block(x):  a1 = lrelu(BN1(conv3x3(x,  W1)));  a2 = lrelu(BN2(conv3x3(a1, W2)));  v3 = BN3(conv3x3(a2, W3))
               vs = BNs(conv1x1(x, Ws));          out = maxpool2(lrelu(v3 + vs))
    features = global average pool of block 4's output  ->  [M, 640]      (channels 64, 160, 320, 640; LeakyReLU slope 0.1)

BatchNorm uses BATCH statistics in training and evaluation (torchmeta's ``MetaBatchNorm2d(track_running_stats=False)``, the
few-shot convention the Conv4 path of this repository follows as well); convolutions carry no bias (batch-statistic
normalisation removes it).  Everything from the feature vector onwards -- hypernetwork head, inner SGD update of (theta, head),
query cross-entropy, arg-max -- is the reference's algorithm (fumi/models/fumi.py:146-192, maml.py:156-191) and shares the pinned
restatement in ``oracle/fumi_ref.py``.

theta = 12 tensors per block: [W1, g1, b1, W2, g2, b2, W3, g3, b3, Ws, gs, bs]  (W OIHW, g = BN weight, b = BN bias).
All functions are dtype-generic (float64 for a high-precision oracle).  The engine computes in bf16 with fp32 accumulation; the
rounding points it has are restated in ``oracle/resnet12_manual.py``.
"""
import torch
import torch.nn.functional as F

from . import fumi_ref as R

BN_EPS = 1e-5
SLOPE = 0.1
CHANNELS = (64, 160, 320, 640)
PER_BLOCK = 12


def _bn(u, g, b):
    return F.batch_norm(u, None, None, g, b, training=True, momentum=1.0, eps=BN_EPS)


def block(x, p):
    W1, g1, b1, W2, g2, b2, W3, g3, b3, Ws, gs, bs = p
    a = F.leaky_relu(_bn(F.conv2d(x, W1, None, padding=1), g1, b1), SLOPE)
    a = F.leaky_relu(_bn(F.conv2d(a, W2, None, padding=1), g2, b2), SLOPE)
    v3 = _bn(F.conv2d(a, W3, None, padding=1), g3, b3)
    vs = _bn(F.conv2d(x, Ws, None), gs, bs)
    return F.max_pool2d(F.leaky_relu(v3 + vs, SLOPE), 2)


def features(x, theta):
    """x [M, Cin, H, W] -> [M, C_last]."""
    for i in range(0, len(theta), PER_BLOCK):
        x = block(x, theta[i:i + PER_BLOCK])
    return x.mean((2, 3))


def forward(x, theta, h):
    return features(x, theta) @ h[:, :-1].t() + h[:, -1]


def episode(theta, h, x_s, y_s, x_q, T, alpha, first_order=False):
    th = list(theta)
    for _ in range(T):
        inner = F.cross_entropy(forward(x_s, th, h), y_s)
        grads = torch.autograd.grad(inner, [h] + th, create_graph=not first_order)
        h = h - alpha * grads[0]
        th = [p - alpha * g for p, g in zip(th, grads[1:])]
    return forward(x_q, th, h)


def fumi_meta_step(theta, phi, text_s, x_s, y_s, x_q, y_q, n_way, T, alpha, tanh_head, need_grad=True, first_order=False,
                   extra=None):
    """FuMI meta-step (fumi.py:115-196) with the ResNet-12 encoder; returns what fumi_ref.fumi_meta_step returns."""
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


def maml_meta_step(params, x_s, y_s, x_q, y_q, T, alpha, first_order=False, need_grad=True):
    """MAML meta-step (maml.py:134-193) with the ResNet-12 encoder: params = theta + [lin_final W [N,F], b [N]]."""
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


def make_params(seed, Cin=3, channels=CHANNELS, dtype=torch.float32):
    """Deterministic parameters for the parity cases: uniform +-1/sqrt(fan_in) conv weights (nn.Conv2d's default scale), BN
    weight around 1 and bias around 0 (not exactly, so their gradient paths are exercised)."""
    import numpy as np
    rs = np.random.RandomState(seed + 32452843)
    theta, ci = [], Cin
    for c in channels:
        for (co, cin, k) in ((c, ci, 3), (c, c, 3), (c, c, 3), (c, ci, 1)):
            bound = 1.0 / np.sqrt(cin * k * k)
            theta.append(torch.from_numpy(rs.uniform(-bound, bound, (co, cin, k, k))).to(dtype))
            theta.append(torch.from_numpy(1.0 + 0.1 * rs.standard_normal(co)).to(dtype))
            theta.append(torch.from_numpy(0.1 * rs.standard_normal(co)).to(dtype))
        ci = c
    return theta


def feature_dim(channels=CHANNELS):
    return channels[-1]
