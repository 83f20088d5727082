"""The hand-derived second-order sweep of the Conv4 meta-step (oracle/conv4_manual.py: what the HIP kernels implement)
against autograd through the same network (oracle/conv4_ref.py), in float64."""
import pytest
import torch
import torch.nn.functional as F

from oracle import conv4_manual as M
from oracle import conv4_ref as C


def _case(seed, B, N, K, Q, Cin, H, W, Cc, nb):
    ep = C.make_image_episodes(seed, B, N, K, Q, Cin, H, W, 8)
    theta = C.make_conv4_params(seed, Cin, Cc, nb, torch.float64)
    Fd = C.feature_dim(H, W, Cc, nb)
    g = torch.Generator().manual_seed(seed)
    head = torch.randn(B, N, Fd + 1, generator=g, dtype=torch.float64) * 0.3
    return {k: (v.double() if v.is_floating_point() else v) for k, v in ep.items()}, theta, head


@pytest.mark.parametrize("T,first_order", [(1, False), (3, False), (2, True), (0, False)])
@pytest.mark.parametrize("shape", [(3, 12, 12, 8, 2), (1, 10, 14, 6, 3), (3, 9, 9, 4, 1)])
def test_manual_sweep_equals_autograd(T, first_order, shape):
    Cin, H, W, Cc, nb = shape
    ep, theta, head = _case(7 + T, 2, 3, 2, 3, Cin, H, W, Cc, nb)
    alpha = 0.05
    for b in range(2):
        th = [t.clone().requires_grad_(True) for t in theta]
        h0 = head[b].clone().requires_grad_(True)
        lq = C.episode(th, h0, ep["x_s"][b], ep["y_s"][b], ep["x_q"][b], T, alpha, first_order)
        loss = F.cross_entropy(lq, ep["y_q"][b])
        g = torch.autograd.grad(loss, th + [h0])
        zq, l2, bar_th, bar_h = M.episode_grads(theta, head[b], ep["x_s"][b], ep["y_s"][b], ep["x_q"][b], ep["y_q"][b], T, alpha,
                                                first_order)
        assert torch.allclose(zq, lq.detach(), rtol=0, atol=1e-11)
        assert abs(float(l2 - loss.detach())) < 1e-12
        for a, r in zip(bar_th + [bar_h], g):
            assert float((a - r).abs().max()) <= 1e-9 * max(1.0, float(r.abs().max())), (a - r).abs().max()


def test_conv_identities():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 7, 6, generator=g, dtype=torch.float64, requires_grad=True)
    W = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(3, 4, 7, 6, generator=g, dtype=torch.float64)
    y = M.conv(x, W)
    gx, gW = torch.autograd.grad((y * dy).sum(), [x, W])
    assert torch.allclose(M.conv_bwd_data(dy, W.detach()), gx, atol=1e-12)
    assert torch.allclose(M.conv_bwd_weight(x.detach(), dy), gW, atol=1e-12)
