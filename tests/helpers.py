"""Shared helpers for the parity tests (oracle = checker, never the product path)."""
import os

import numpy as np
import torch

from oracle import casegen as cg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def case_seed(name):
    return sum(ord(ch) for ch in name)


FLOOR = 1e-5      # tensors that are exactly 0 in real arithmetic (e.g. d/d hyper_net.2.bias without tanh:
                  # the softmax gradient sums to zero over classes) carry only ~1e-8 round-off noise


def rel_to_max(a, b, floor=FLOOR):
    """|a-b|_inf / max(|b|_inf, floor): the tolerance semantics of SURVEY.md 7.3 / BASELINE.md."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def assert_close_max(a, b, tol, what=""):
    r = rel_to_max(a, b)
    assert r <= tol, f"{what}: rel-to-max error {r:.3e} > {tol:.1e}"


def grad_floor(gold):
    """Noise floor for gradient tensors that are exactly 0 in real arithmetic (e.g. d/d hyper_net.2.bias without
    tanh: the softmax gradient sums to zero over classes, so only ~1e-7 x (global gradient scale) of round-off
    is left).  5 % of the largest sampled |gradient| of the whole model."""
    m = max(float(abs(v[3:]).max()) for k, v in gold.items() if k.startswith("grad.") and k.endswith(".digest"))
    return max(0.05 * m, FLOOR)


def check_grad(gold, key, g, tol):
    """Compare a gradient with the fixture: full tensor when stored, digest (sum, abs-sum, l2, samples) otherwise."""
    floor = grad_floor(gold)
    if key in gold:
        r = rel_to_max(g, gold[key], floor)
        assert r <= tol, f"{key}: rel-to-max error {r:.3e} > {tol:.1e}"
    d_ref = gold[key + ".digest"]
    d = cg.digest(g)
    scale = max(abs(d_ref[3:]).max(), floor)
    assert abs(d[3:] - d_ref[3:]).max() <= tol * scale, f"{key} digest samples"
    assert abs(d[2] - d_ref[2]) <= tol * max(d_ref[2], floor * 10), f"{key} l2"


def safe_margin_mask(logits, min_margin):
    """Rows whose top-1/top-2 margin exceeds the noise floor (argmax must be bit-exact there)."""
    top2 = torch.as_tensor(logits).topk(2, dim=-1)[0]
    return (top2[..., 0] - top2[..., 1]) > min_margin
