"""CPU suite: the autograd-free ResNet-12 sweep the bf16 kernels implement (oracle/resnet12_manual.py) equals autograd through
oracle/resnet12_ref.py in float64 -- the derivation (residual join, 1x1 shortcut, LeakyReLU masks, BN double backward, average
pool) is right before any kernel is compared with it.  ResNet-12 is "parity unpinned" (the reference has no such encoder)."""
import numpy as np
import pytest
import torch

from oracle import resnet12_manual as M
from oracle import resnet12_ref as RR
from oracle import conv4_ref as CR
from oracle import casegen as cg


def _case(seed, B, N, K, Q, H, channels, Dt=6, Ht=5):
    ep = CR.make_image_episodes(seed, B, N, K, Q, 3, H, H, Dt)
    theta = RR.make_params(seed, 3, channels, torch.float64)
    rs = np.random.RandomState(seed)
    F_ = channels[-1]
    phi = [torch.from_numpy(rs.standard_normal(s) * 0.3) for s in ((Ht, Dt), (Ht,), (F_ + 1, Ht), (F_ + 1,))]
    ep = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in ep.items()}
    return ep, theta, phi


@pytest.mark.parametrize("T,tanh", [(1, False), (2, True)])
def test_manual_sweep_equals_autograd_in_float64(T, tanh):
    ep, theta, phi = _case(3 + T, 2, 3, 2, 2, 16, (4, 6, 8, 10))
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = RR.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], 3, T, 0.05, tanh)
    man = M.fumi_meta_step(theta, phi, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], 3, T, 0.05, tanh)
    assert float((ref["logits"] - man["logits"]).abs().max()) < 1e-10
    for a, b in zip(ref["g_theta"] + ref["g_phi"], man["g_theta"] + man["g_phi"]):
        assert float((a - b).abs().max()) <= 1e-9 * max(1.0, float(a.abs().max()))


def test_bf16_rounding_points_cost_what_bf16_costs():
    """Forward: features within a few bf16 ulps of float64.  Backward rounding alone (on the float64 tape): gradients within 3 %.
    (Forward rounding ALSO flips LeakyReLU / arg-max decisions near ties, which re-routes gradients: a bf16 network's gradient is
    the gradient of the rounded function, so whole-step gradients are compared with the bf16-rounded sweep, not with float64.)"""
    ep, theta, phi = _case(11, 2, 3, 3, 3, 32, (16, 32, 32, 64))
    x, y = ep["x_s"][0], ep["y_s"][0]
    h = torch.from_numpy(np.random.RandomState(0).standard_normal((3, 65)) * 0.1)
    za, ta = M.net_fwd(x, theta, h, M._id)
    zb, tb = M.net_fwd(x, theta, h, M.bf16_round)
    assert float((ta["f"] - tb["f"]).abs().max() / ta["f"].abs().max()) < 0.06
    ga, _ = M.net_bwd(za, y, ta, 1.0 / 9, M._id)
    zc, tc = M.net_fwd(x, theta, h, M._id)
    gc, _ = M.net_bwd(zc, y, tc, 1.0 / 9, M.bf16_round)
    for a, c in zip(ga, gc):
        assert float((a - c).norm() / a.norm()) < 0.03
