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
    is left).  5 % of the largest sampled |gradient| of the whole model.

    What the floor lets through, checked against every reference-generated fixture: it only ever decides the comparison of tensors
    that are ANALYTICALLY zero (``hyper_net.2.bias`` without tanh: the soft-max gradient sums to zero over the classes) and of AM3's
    ``h.0.weight`` (largest entry 4.9e-4 of the model-wide maximum: compared at an effective 1e-4 / 4.9e-4 of its own scale); every other
    tensor's own maximum is above the floor, so its tolerance is the stated one relative to its own scale."""
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


# ---- the engine's counter-based dropout mask, restated in numpy (csrc/episode.hip: drop_mix / drop_key / drop_relu) ------------
def _mix(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout_mask(seed, p, b, call, layer, rows, width):
    """[rows, width] float32 mask in {0, 1/(1-p)} for episode b, forward call `call`, hidden layer `layer`."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    one = lambda v: _mix(np.array([v & 0xFFFFFFFF], dtype=np.uint64))[0]
    h = one(lo + 0x9E3779B9 * b)
    h = one(int(h) ^ ((hi + 0x85EBCA6B * call) & 0xFFFFFFFF))
    h = one(int(h) + 0xC2B2AE35 * layer)
    idx = np.arange(rows * width, dtype=np.uint64)
    u = _mix((np.uint64(int(h)) ^ idx) & 0xFFFFFFFF)
    thr = max(1, int(p * 4294967296.0))
    keep = (u >= thr).reshape(rows, width)
    return torch.from_numpy(keep.astype(np.float32) / np.float32(1.0 - p))


def dropout_mask_flat(seed, p, tag, rows, width):
    """[rows, width] mask of a GEMM epilogue (csrc/am3.hip: dkey(tag); csrc/gemm.hip epilogue)."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    one = lambda v: _mix(np.array([v & 0xFFFFFFFF], dtype=np.uint64))[0]
    key = one(int(one(lo ^ ((0x9E3779B9 * tag) & 0xFFFFFFFF))) ^ hi)
    u = _mix((np.uint64(int(key)) ^ np.arange(rows * width, dtype=np.uint64)) & 0xFFFFFFFF)
    keep = (u >= max(1, int(p * 4294967296.0))).reshape(rows, width)
    return torch.from_numpy(keep.astype(np.float32) / np.float32(1.0 - p))


# ---- the FUMI + trainable bi-LSTM fixture (tests/golden/fumi_rnn_finetune.npz, oracle/refharness/gen_golden.py) ----------------
RNN_KEYS = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
            "bias_ih_l0_reverse", "bias_hh_l0_reverse"]


def rnn_finetune_case():
    """(gold, dims dict, episodes with the fixture's token rows, theta, phi, word table, the 8 LSTM tensors in nn.LSTM's order)."""
    gold = load_golden("fumi_rnn_finetune")
    B, N, K, Q, D, h0, Dt, Ht, T, L, E = [int(v) for v in gold["dims"]]
    seed = int(gold["seed"])
    ep = cg.make_episodes(seed, B, N, K, Q, D, Dt, blocked=False)
    ep["text_s"], ep["text_q"] = torch.from_numpy(gold["text_s"]), torch.from_numpy(gold["text_q"])
    theta, phi = cg.make_fumi_params(seed, D, [h0], Dt, Ht)
    table = torch.from_numpy(gold["text_encoder.embed.weight"])
    lstm_w = [torch.from_numpy(gold["text_encoder.rnn." + k]) for k in RNN_KEYS]
    dims = dict(B=B, N=N, K=K, Q=Q, D=D, hid=[h0], Dt=Dt, Ht=Ht, T=T, L=L, E=E)
    return gold, dims, ep, theta, phi, table, lstm_w
