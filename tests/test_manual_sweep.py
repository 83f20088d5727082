"""CPU suite: the hand-derived low-rank forward tape + second-order reverse sweep (the algebra the HIP kernels
implement, oracle/manual_sweep.py) against the autograd oracle (oracle/fumi_ref.py), in float64."""
import pytest
import torch

from oracle import casegen as cg
from oracle import fumi_ref as R
from oracle import manual_sweep as M
from helpers import case_seed, rel_to_max


@pytest.mark.parametrize("name", [n for n in cg.FUMI_CASES if "default" not in n])
def test_fumi_manual_equals_autograd_fp64(name):
    c = cg.FUMI_CASES[name]
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    theta, phi = cg.make_fumi_params(seed, c["D"], c["hid"], c["Dt"], c["Ht"])
    d = lambda t: t.double()
    theta, phi = [d(t) for t in theta], [d(t) for t in phi]
    th_l = [t.clone().requires_grad_(True) for t in theta]
    ph_l = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th_l, ph_l, d(ep["text_s"]), d(ep["x_s"]), ep["y_s"], d(ep["x_q"]), ep["y_q"],
                           c["N"], c["T"], cg.ALPHA, c["tanh"])
    man = M.fumi_meta_step_manual(theta, phi, d(ep["text_s"]), d(ep["x_s"]), ep["y_s"], d(ep["x_q"]), ep["y_q"],
                                  c["N"], c["T"], cg.ALPHA, c["tanh"])
    assert rel_to_max(man["logits"], ref["logits"]) < 1e-12
    assert abs(float(man["loss"]) - float(ref["loss"])) < 1e-12
    for a, b in zip(man["g_theta"] + man["g_phi"], ref["g_theta"] + ref["g_phi"]):
        assert rel_to_max(a, b, floor=1e-6) < 1e-9


@pytest.mark.parametrize("second_order", [True, False])
def test_maml_manual_equals_autograd_fp64(second_order):
    c = cg.MAML_CASES["maml_2nd"]
    ep = cg.make_episodes(3, c["B"], c["N"], 2, c["Q"], c["D"], 8)
    p = [t.double() for t in cg.make_maml_params(3, c["D"], c["hid"], c["N"])]
    pl = [t.clone().requires_grad_(True) for t in p]
    ref = R.maml_meta_step(pl, ep["x_s"].double(), ep["y_s"], ep["x_q"].double(), ep["y_q"], 3, cg.ALPHA,
                           first_order=not second_order)
    B = c["B"]
    h0 = torch.cat([p[-2], p[-1][:, None]], 1)
    acc = None
    for b in range(B):
        o = M.episode_manual(p[:-2], h0, ep["x_s"][b].double(), ep["y_s"][b], ep["x_q"][b].double(), ep["y_q"][b],
                             3, cg.ALPHA, second_order=second_order)
        g = [o["A0bar_s"].t() @ ep["x_s"][b].double() + o["A0bar_q"].t() @ ep["x_q"][b].double(), o["g_b0"]]
        for i in range(1, len(p) // 2 - 1):
            g += [o["g_W"][i], o["g_b"][i]]
        g += [o["g_h0"][:, :-1], o["g_h0"][:, -1]]
        acc = g if acc is None else [x + y for x, y in zip(acc, g)]
        assert rel_to_max(o["logits"], ref["logits"][b]) < 1e-12
    for a, r in zip(acc, ref["g_params"]):
        assert rel_to_max(a / B, r, floor=1e-6) < 1e-9
