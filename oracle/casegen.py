"""Deterministic synthetic episodes + parameters for the parity cases -- TEST INFRASTRUCTURE ONLY.

Shared by ``oracle/refharness/gen_golden.py`` (which feeds them to the real reference) and by
``tests/`` (which feed the same values to the oracle restatement and to the HIP path).  Everything is
drawn from ``numpy.random.RandomState`` (stream frozen by NumPy's compatibility policy), so a fixture
only has to store the case's config + the reference's outputs; ``digest`` values stored next to the
outputs catch any drift of the regenerated inputs.

Batch layout follows the reference loader contract (fumi/dataset/data.py:571-581):
  batch = {'train': ([idx [B,S] i64, text [B,S,Dt] f32 | [B,S,L] i64, im [B,S,D] f32], targets [B,S] i64),
           'test' : same with Qn rows}
The text row is identical for all samples of a class (data.py:543-549).
"""
import numpy as np
import torch


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def make_targets(rs, B, N, per_class, blocked):
    """Balanced labels; class-blocked (loader order) or per-episode shuffled (the code must not rely on order)."""
    y = np.tile(np.repeat(np.arange(N), per_class)[None], (B, 1))
    if not blocked:
        for b in range(B):
            y[b] = y[b][rs.permutation(N * per_class)]
    return y.astype(np.int64)


def make_episodes(seed, B, N, K, Q, D, Dt, blocked=False, tokens=None, learnable=False):
    """Returns dict of CPU tensors: x_s,y_s,x_q,y_q,text_s,text_q (+ idx).  tokens=(V,L,pad) -> int64 token text."""
    rs = np.random.RandomState(seed)
    S, Qn = N * K, N * Q
    y_s = make_targets(rs, B, N, K, blocked)
    y_q = make_targets(rs, B, N, Q, blocked)
    if learnable:
        mu = rs.standard_normal((B, N, D))
        x_s = np.take_along_axis(mu, y_s[..., None], 1) + 2.0 * rs.standard_normal((B, S, D))
        x_q = np.take_along_axis(mu, y_q[..., None], 1) + 2.0 * rs.standard_normal((B, Qn, D))
    else:
        x_s = rs.standard_normal((B, S, D))
        x_q = rs.standard_normal((B, Qn, D))
    if tokens is None:
        cls_text = rs.standard_normal((B, N, Dt))
    else:
        V, L, pad = tokens
        cls_text = np.full((B, N, L), pad, dtype=np.int64)
        for b in range(B):
            for n in range(N):
                ln = rs.randint(1, L + 1)
                toks = rs.randint(0, V, size=ln)
                toks[toks == pad] = (pad + 1) % V
                cls_text[b, n, :ln] = toks
    text_s = np.take_along_axis(cls_text, y_s[..., None], 1)
    text_q = np.take_along_axis(cls_text, y_q[..., None], 1)
    tdt = torch.float32 if tokens is None else torch.int64
    return dict(x_s=_t(x_s), y_s=_t(y_s, torch.int64), x_q=_t(x_q), y_q=_t(y_q, torch.int64),
                text_s=_t(text_s, tdt), text_q=_t(text_q, tdt),
                idx_s=torch.arange(B * S).view(B, S), idx_q=torch.arange(B * Qn).view(B, Qn) + B * S)


def to_batch(ep):
    """The reference loader's batch dict."""
    return {'train': ([ep['idx_s'], ep['text_s'], ep['x_s']], ep['y_s']),
            'test': ([ep['idx_q'], ep['text_q'], ep['x_q']], ep['y_q'])}


def _linear(rs, out_f, in_f, scale=1.0):
    bound = scale / np.sqrt(in_f)
    return (rs.uniform(-bound, bound, (out_f, in_f)), rs.uniform(-bound, bound, (out_f,)))


def make_fumi_params(seed, D, hid, Dt, Ht, head_scale=1.0):
    """theta = [W0,b0,W1,b1,..] (im_net.linear{i}), phi = [A0,a0,A1,a1] (hyper_net.0 / hyper_net.2)."""
    rs = np.random.RandomState(seed + 7919)
    theta, d = [], D
    for h in hid:
        W, b = _linear(rs, h, d)
        theta += [_t(W), _t(b)]
        d = h
    A0, a0 = _linear(rs, Ht, Dt)
    A1, a1 = _linear(rs, hid[-1] + 1, Ht, head_scale)
    return theta, [_t(A0), _t(a0), _t(A1), _t(a1)]


def make_maml_params(seed, D, hid, N):
    rs = np.random.RandomState(seed + 104729)
    p, d = [], D
    for h in (hid or []):
        W, b = _linear(rs, h, d)
        p += [_t(W), _t(b)]
        d = h
    W, b = _linear(rs, N, d)
    return p + [_t(W), _t(b)]


def make_am3_params(seed, D, Dt, Ht, P):
    rs = np.random.RandomState(seed + 1299709)
    w = {}
    for (kw, kb, o, i) in (("Wi", "bi", P, D), ("G0", "g0", Ht, Dt), ("G1", "g1", P, Ht),
                            ("H0", "h0", Ht, P), ("H1", "h1", 1, Ht)):
        W, b = _linear(rs, o, i)
        w[kw], w[kb] = _t(W), _t(b)
    return w


def fumi_state_dict(theta, phi):
    """state_dict keys of the reference FUMI: im_net.linear{i}.*, hyper_net.{0,2}.*"""
    sd = {}
    for i in range(len(theta) // 2):
        sd[f"im_net.linear{i}.weight"], sd[f"im_net.linear{i}.bias"] = theta[2 * i], theta[2 * i + 1]
    sd["hyper_net.0.weight"], sd["hyper_net.0.bias"], sd["hyper_net.2.weight"], sd["hyper_net.2.bias"] = phi
    return sd


def maml_state_dict(p):
    sd, n = {}, len(p) // 2 - 1
    for i in range(n):
        sd[f"net.lin_{i}.weight"], sd[f"net.lin_{i}.bias"] = p[2 * i], p[2 * i + 1]
    sd["net.lin_final.weight"], sd["net.lin_final.bias"] = p[-2], p[-1]
    return sd


def am3_state_dict(w):
    return {"image_encoder.weight": w["Wi"], "image_encoder.bias": w["bi"],
            "g.0.weight": w["G0"], "g.0.bias": w["g0"], "g.3.weight": w["G1"], "g.3.bias": w["g1"],
            "h.0.weight": w["H0"], "h.0.bias": w["h0"], "h.3.weight": w["H1"], "h.3.bias": w["h1"]}


def digest(t):
    """Order-independent-ish summary of a tensor (float64): [sum, abs-sum, l2, strided sample...]."""
    a = t.detach().to(torch.float64).reshape(-1)
    step = max(1, a.numel() // 61)
    return torch.cat([torch.stack([a.sum(), a.abs().sum(), a.pow(2).sum().sqrt()]), a[::step][:64]]).numpy()


# ---- the parity case table (name -> config) -----------------------------------------------------------------
FUMI_CASES = {
    # name:            B  N  K  Q   D    hid        Dt  Ht  T  tanh  init_bias blocked
    "fumi_t1":        dict(B=4, N=5, K=5, Q=4, D=64, hid=[32, 16], Dt=24, Ht=20, T=1, tanh=False, init_bias=False, blocked=False),
    "fumi_t5":        dict(B=4, N=5, K=5, Q=4, D=64, hid=[32, 16], Dt=24, Ht=20, T=5, tanh=False, init_bias=False, blocked=False),
    "fumi_t5_tanh":   dict(B=4, N=5, K=5, Q=4, D=64, hid=[32, 16], Dt=24, Ht=20, T=5, tanh=True, init_bias=False, blocked=True),
    "fumi_1shot":     dict(B=4, N=5, K=1, Q=4, D=64, hid=[32, 16], Dt=24, Ht=20, T=5, tanh=True, init_bias=False, blocked=False),
    "fumi_initbias":  dict(B=3, N=5, K=5, Q=4, D=64, hid=[32, 16], Dt=24, Ht=20, T=1, tanh=False, init_bias=True, blocked=False),
    "fumi_1layer":    dict(B=3, N=4, K=3, Q=5, D=48, hid=[24], Dt=16, Ht=12, T=3, tanh=True, init_bias=False, blocked=False),
    "fumi_3layer":    dict(B=2, N=3, K=4, Q=3, D=40, hid=[24, 16, 8], Dt=16, Ht=12, T=2, tanh=False, init_bias=False, blocked=False),
    "fumi_20way":     dict(B=2, N=20, K=5, Q=3, D=96, hid=[48, 24], Dt=32, Ht=24, T=2, tanh=True, init_bias=False, blocked=False),
    "fumi_default":   dict(B=2, N=5, K=5, Q=4, D=2048, hid=[256, 64], Dt=768, Ht=256, T=5, tanh=False, init_bias=False, blocked=False),
    # BASELINE.json configs[2] per-rank episode shape: 32 query/class, 5 inner steps, BERT-width text
    "fumi_default_t5_q32": dict(B=2, N=5, K=5, Q=32, D=2048, hid=[256, 64], Dt=768, Ht=256, T=5, tanh=False, init_bias=False, blocked=False),
    "fumi_default_t1": dict(B=2, N=5, K=5, Q=32, D=2048, hid=[256, 64], Dt=768, Ht=256, T=1, tanh=True, init_bias=False, blocked=True),
}
MAML_CASES = {
    "maml_2nd":       dict(B=4, N=5, K=1, Q=4, D=64, hid=[32, 16], T=5, first_order=False),
    "maml_1st":       dict(B=4, N=5, K=1, Q=4, D=64, hid=[32, 16], T=5, first_order=True),
    "maml_5shot_t1":  dict(B=3, N=5, K=5, Q=4, D=64, hid=[32, 16], T=1, first_order=False),
    "maml_default":   dict(B=2, N=5, K=1, Q=8, D=2048, hid=[256, 64], T=5, first_order=False),
    # hidden_dims=None: the network is lin_final alone (maml.py:24-31)
    "maml_linear_2nd": dict(B=3, N=5, K=5, Q=6, D=128, hid=None, T=3, first_order=False),
    "maml_linear_1st": dict(B=3, N=4, K=1, Q=5, D=128, hid=None, T=2, first_order=True),
}
AM3_CASES = {
    "am3_lam":        dict(B=4, N=5, K=5, Q=4, D=64, Dt=24, Ht=20, P=16, lamda_fixed=None),
    "am3_lam0":       dict(B=4, N=5, K=5, Q=4, D=64, Dt=24, Ht=20, P=16, lamda_fixed=0),
    "am3_lam1":       dict(B=4, N=5, K=5, Q=4, D=64, Dt=24, Ht=20, P=16, lamda_fixed=1),
    "am3_default":    dict(B=2, N=5, K=5, Q=8, D=2048, Dt=768, Ht=256, P=64, lamda_fixed=None),
    # BASELINE.json configs[3] per-rank episode shape (32 query/class, reference widths) on a reduced meta-batch
    "am3_default_q32": dict(B=4, N=5, K=5, Q=32, D=2048, Dt=768, Ht=256, P=64, lamda_fixed=None),
}
ALPHA = 0.01     # --step_size default, fumi/utils/utils.py:164-167
