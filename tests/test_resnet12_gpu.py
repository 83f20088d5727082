"""GPU tests of the bf16 ResNet-12 encoder at the im_net seam (BASELINE.json configs[4]) through the C ABI.

"Parity unpinned": the reference has no ResNet-12 (fumi/models/fumi.py:89-100 is only the seam).  The oracle is this repository's
own restatement: oracle/resnet12_ref.py (autograd) and oracle/resnet12_manual.py (the autograd-free sweep the kernels implement,
equal to autograd at 1e-9 in float64 -- tests/test_resnet12_manual.py -- with a rounding hook that rounds to bf16 where the
engine stores bf16).

Tolerances come from bf16's 8-bit significand (spacing 2^-8 relative, round-off <= 2^-9 per stored value):
  * one matrix product against torch on the same bf16 inputs: the output rounding only, <= 2^-8 of max|y| (fp32 outputs: 1e-5);
  * a chain of L roundings decorrelates: a value that rounds the other way in the engine than in the oracle (fp32 summation order
    decides ties) perturbs everything downstream and, near a LeakyReLU / arg-max tie, re-routes a gradient.  One block deep with no
    inner step the engine and the bf16-rounded sweep agree to 1e-3; every further block and inner step multiplies the drift
    (measured 3e-2 after one inner step, 6e-2 after two, 1.5e-1 at 3 blocks).  The whole-step bounds below are those measured
    drifts with a factor 2-3 of head room; the SHARP checks are the one-block first-order step, the extracted Hessian-vector
    product and the unit products.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv4_ref as CR
from oracle import resnet12_manual as M
from oracle import resnet12_ref as RR
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

BF = 2.0 ** -8


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ws(dev):
    from fumi_amd import hip
    return hip.Workspace.get(dev)


def bf(t):
    return t.to(torch.bfloat16).to(t.dtype)


def to_cl(x):
    """[B, M, C, H, W] -> bf16 padded channels-last [B, M (H+2)(W+2), C]"""
    B, Mi, C, H, W = x.shape
    return F.pad(x, (1, 1, 1, 1)).permute(0, 1, 3, 4, 2).reshape(B, Mi * (H + 2) * (W + 2), C).contiguous().to(torch.bfloat16)


def from_cl(y, Mi, H, W):
    B, _, C = y.shape
    return y.float().reshape(B, Mi, H + 2, W + 2, C).permute(0, 1, 4, 2, 3)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(got, ref):
    num = sum(float(((a.cpu().double() - b.double()) ** 2).sum()) for a, b in zip(got, ref))
    return (num / sum(float((b.double() ** 2).sum()) for b in ref)) ** 0.5


def cosine(got, ref):
    dot = sum(float((a.cpu().double() * b.double()).sum()) for a, b in zip(got, ref))
    na = sum(float((a.cpu().double() ** 2).sum()) for a in got) ** 0.5
    return dot / (na * sum(float((b.double() ** 2).sum()) for b in ref) ** 0.5)


# ---- every matrix product against torch ------------------------------------------------------------------------------------------
SHAPES = [(2, 3, 10, 10, 16, 64, 3), (1, 5, 21, 21, 64, 160, 3), (2, 2, 12, 9, 160, 64, 1), (1, 4, 7, 7, 320, 320, 3),
          (1, 2, 42, 42, 64, 64, 3), (1, 7, 5, 5, 640, 160, 3), (2, 3, 8, 8, 32, 96, 3), (1, 3, 9, 11, 96, 128, 1),
          (1, 1, 84, 84, 16, 64, 3), (3, 1, 2, 2, 32, 32, 3)]


@pytest.mark.parametrize("B,Mi,H,W,Cin,Cout,k", SHAPES)
def test_conv_products_match_torch(B, Mi, H, W, Cin, Cout, k, dev, ws):
    """Forward conv (+ its batch statistics), input-gradient conv and weight gradient of one layer shape on raw bf16 maps."""
    from fumi_amd import hip
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
    x = bf(torch.randn(B, Mi, Cin, H, W, generator=g))
    Wt = torch.randn(B, Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    ref = torch.stack([F.conv2d(x[b], bf(Wt[b]), None, padding=k // 2) for b in range(B)])
    y, st = hip.rn12_conv(ws, to_cl(x).to(dev), Wt.to(dev), H, W, want_stats=True)
    yc = from_cl(y.cpu(), Mi, H, W)
    assert float(yc[..., 0, :].abs().max() + yc[..., -1, :].abs().max() + yc[..., :, 0].abs().max() + yc[..., :, -1].abs().max()) == 0.0
    yi = yc[..., 1:-1, 1:-1]
    assert rel(yi, ref) <= BF
    s_ref = torch.stack([yi.sum((1, 3, 4)), (yi * yi).sum((1, 3, 4))], 1)          # statistics of the STORED values
    assert rel(st.cpu(), s_ref) <= 1e-5
    if Cin % 32:
        return
    dy = bf(torch.randn(B, Mi, Cout, H, W, generator=g))
    ref = torch.stack([F.conv2d(dy[b], bf(Wt[b]).flip(2, 3).transpose(0, 1), None, padding=k // 2) for b in range(B)])
    dx = hip.rn12_conv(ws, to_cl(dy).to(dev), Wt.to(dev), H, W, transpose=True)
    assert rel(from_cl(dx.cpu(), Mi, H, W)[..., 1:-1, 1:-1], ref) <= BF
    dW = hip.rn12_wgrad(ws, to_cl(x).to(dev), to_cl(dy).to(dev), H, W, k)
    ref = torch.stack([M.conv_bwd_weight(x[b].double(), dy[b].double(), k) for b in range(B)]).float()
    assert rel(dW.cpu(), ref) <= 1e-5


# ---- whole steps -------------------------------------------------------------------------------------------------------------------
def case(seed, B, N, K, Q, H, channels, Dt=6, Ht=5):
    ep = CR.make_image_episodes(seed, B, N, K, Q, 3, H, H, Dt)
    theta = RR.make_params(seed, 3, channels, torch.float32)
    rs = np.random.RandomState(seed)
    F_ = channels[-1]
    phi = [torch.from_numpy((rs.standard_normal(s) * 0.3).astype(np.float32)) for s in ((Ht, Dt), (Ht,), (F_ + 1, Ht), (F_ + 1,))]
    return ep, theta, phi


def maml_sweep(ep, theta, h0, T, alpha, first_order, rnd):
    B = ep["x_s"].shape[0]
    th64 = [t.double() for t in theta]
    gs = [torch.zeros_like(t) for t in th64]; gh = torch.zeros_like(h0); zs = []
    for b in range(B):
        zq, _, bth, bh = M.episode_grads(th64, h0, ep["x_s"][b].double(), ep["y_s"][b], ep["x_q"][b].double(), ep["y_q"][b], T, alpha,
                                         first_order=first_order, rnd=rnd)
        zs.append(zq)
        for a, g_ in zip(gs, bth):
            a += g_ / B
        gh += bh / B
    return torch.stack(zs), gs, gh


def run_maml(ws, dev, ep, theta, Wf, bfin, T, alpha, first_order, **kw):
    from fumi_amd import hip
    out = hip.maml_resnet12_step(ws, ep["x_s"].to(dev), ep["y_s"].to(dev), ep["x_q"].to(dev), ep["y_q"].to(dev),
                                 [t.to(dev) for t in theta + [Wf, bfin]], T, alpha, first_order, **kw)
    assert ws.read_status() == 0
    return out


def head_of(N, Fdim, seed=1):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal((N, Fdim)) * 0.1).astype(np.float32)), torch.zeros(N)


@pytest.mark.parametrize("channels,H", [((32,), 8), ((64,), 12), ((32, 64), 16), ((32, 64, 64, 128), 32)])
def test_features_match_the_bf16_sweep(channels, H, dev, ws):
    from fumi_amd import hip
    ep, theta, _ = case(5, 2, 3, 3, 2, H, channels)
    f = hip.resnet12_features(ws, ep["x_s"].to(dev), [t.to(dev) for t in theta])
    th64 = [t.double() for t in theta]
    h = torch.zeros(3, channels[-1] + 1, dtype=torch.float64)
    tol = 4e-3 * len(channels) ** 2                    # one block: the roundings of three convolutions; drift grows with depth
    for b in range(2):
        _, tb = M.net_fwd(ep["x_s"][b].double(), th64, h, M.bf16_round)
        _, t64 = M.net_fwd(ep["x_s"][b].double(), th64, h, M._id)
        assert rel(f[b].cpu().double(), tb["f"]) <= tol
        assert rel(f[b].cpu().double(), t64["f"]) <= 2e-2 * len(channels)


def test_one_block_first_order_step_matches_the_sweep_tightly(dev, ws):
    """The sharp whole-step check: one block, no inner step -> forward, head, first-order backward through the residual join, the
    1x1 shortcut, both BN forms and all four weight gradients, with (almost) nothing to decorrelate."""
    channels, H, N = (32,), 8, 3
    ep, theta, _ = case(7, 2, N, 3, 2, H, channels)
    Wf, bfin = head_of(N, channels[-1])
    out = run_maml(ws, dev, ep, theta, Wf, bfin, 0, 0.05, False)
    zs, gs, gh = maml_sweep(ep, theta, torch.cat([Wf, bfin[:, None]], 1).double(), 0, 0.05, False, M.bf16_round)
    assert rel(out["logits"].cpu().double(), zs) <= 2e-3
    assert torch.equal(out["preds"].cpu(), zs.argmax(-1))
    for a, b in zip(out["g_params"][:-2], gs):
        assert float((a.cpu().double() - b).norm() / b.norm()) <= 1e-2
    ghg = torch.cat([out["g_params"][-2].cpu().double(), out["g_params"][-1].cpu().double()[:, None]], 1)
    assert rel(ghg, gh) <= 2e-3


@pytest.mark.parametrize("channels,H,tol", [((32,), 8, 0.04), ((32, 64), 16, 0.12)])
def test_hessian_vector_product_extracted_from_the_steps(channels, H, tol, dev, ws):
    """g_first_order - g_second_order = alpha * H bar on the SAME tape (both runs share every forward / backward value), so the
    difference isolates the tangent passes.  With a tiny alpha the tape is (nearly) the sweep's own: the extracted H bar of the
    engine and of the bf16-rounded sweep agree at bf16 level (`tol`: one block sharp, two blocks with the drift of the deeper net)."""
    N, alpha = 3, 1e-3
    ep, theta, _ = case(9, 2, N, 3, 2, H, channels)
    Wf, bfin = head_of(N, channels[-1])
    g1 = [t.cpu().double().clone() for t in run_maml(ws, dev, ep, theta, Wf, bfin, 1, alpha, True)["g_params"]]
    g2 = [t.cpu().double().clone() for t in run_maml(ws, dev, ep, theta, Wf, bfin, 1, alpha, False)["g_params"]]
    h0 = torch.cat([Wf, bfin[:, None]], 1).double()
    _, s1, _ = maml_sweep(ep, theta, h0, 1, alpha, True, M.bf16_round)
    _, s2, _ = maml_sweep(ep, theta, h0, 1, alpha, False, M.bf16_round)
    hv_e = [(a - b) / alpha for a, b in zip(g1[:-2], g2[:-2])]
    hv_s = [(a - b) / alpha for a, b in zip(s1, s2)]
    assert sum(float(h.norm()) for h in hv_s) > 1e-3                      # (there is a second-order term to compare)
    assert rel_l2(hv_e, hv_s) <= tol
    assert cosine(hv_e, hv_s) >= 1 - tol


@pytest.mark.parametrize("channels,H,T,first_order,tol", [
    ((32,), 8, 1, True, 0.08), ((32,), 8, 1, False, 0.08), ((32,), 8, 2, False, 0.16)])
def test_whole_steps_against_the_bf16_sweep(channels, H, T, first_order, tol, dev, ws):
    """Inner steps included (first and second order): logits and meta-gradients stay within the measured drift of a bf16 chain.  One
    block only: deeper nets decorrelate (0.4 rel-L2 at two blocks, 0.95 at four -- bounds that could not fail), so depth, the true
    channel widths and T up to 5 are covered STAGE BY STAGE on the engine's own stored maps instead, where nothing accumulates
    (tests/test_resnet12_probe.py, tests/rn12_stages.py: 2^-7 / 2e-3 per map, 2e-4 per sum)."""
    N = 3
    ep, theta, _ = case(7, 2, N, 3, 2, H, channels)
    Wf, bfin = head_of(N, channels[-1])
    out = run_maml(ws, dev, ep, theta, Wf, bfin, T, 0.05, first_order)
    zs, gs, _ = maml_sweep(ep, theta, torch.cat([Wf, bfin[:, None]], 1).double(), T, 0.05, first_order, M.bf16_round)
    assert rel(out["logits"].cpu().double(), zs) <= tol / 2
    assert rel_l2(out["g_params"][:-2], gs) <= tol
    assert cosine(out["g_params"][:-2], gs) >= 1 - tol


def test_fumi_step_with_the_hypernetwork(dev, ws):
    """FuMI form: text rows -> class select -> hypernetwork -> [N, F+1] heads; the head adjoints flow back into the four
    hypernetwork tensors (fumi.py:104-113,198-212)."""
    from fumi_amd import hip
    channels, H, N = (32,), 8, 3
    ep, theta, phi = case(13, 3, N, 3, 2, H, channels)
    out = hip.fumi_resnet12_step(ws, N, ep["x_s"].to(dev), ep["y_s"].to(dev), ep["x_q"].to(dev), ep["y_q"].to(dev),
                                 [t.to(dev) for t in theta], [t.to(dev) for t in phi], 1, 0.05, True, text_s=ep["text_s"].to(dev))
    assert ws.read_status() == 0
    ref = M.fumi_meta_step([t.double() for t in theta], [t.double() for t in phi], ep["text_s"].double(), ep["x_s"].double(), ep["y_s"],
                           ep["x_q"].double(), ep["y_q"], N, 1, 0.05, True, rnd=M.bf16_round)
    assert rel(out["logits"].cpu().double(), ref["logits"]) <= 0.04
    assert rel(out["loss_b"].cpu().double(), ref["loss_b"]) <= 0.04
    assert rel_l2(out["g_theta"], ref["g_theta"]) <= 0.08
    assert rel_l2(out["g_phi"], ref["g_phi"]) <= 0.08


def test_episode_chunks_sum_to_the_whole_meta_batch(dev, ws):
    """The meta-batch is processed in chunks of episodes (the tape of a 20-way ResNet-12 meta-batch does not fit HBM at once): a
    chunked run gives the same per-episode forward results and the same summed meta-gradient (summation order aside)."""
    channels, H, N = (32, 64), 16, 3
    ep, theta, _ = case(21, 5, N, 2, 2, H, channels)
    Wf, bfin = head_of(N, channels[-1])
    whole = run_maml(ws, dev, ep, theta, Wf, bfin, 0, 0.05, False, chunk=5)
    whole = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in whole.items() if v is not None}
    parts = run_maml(ws, dev, ep, theta, Wf, bfin, 0, 0.05, False, chunk=2)          # chunks of 2, 2, 1 episodes
    assert torch.equal(whole["logits"], parts["logits"]) and torch.equal(whole["preds"], parts["preds"])
    assert torch.equal(whole["loss_b"], parts["loss_b"])
    # an episode's launch geometry (reduction slabs, tile shapes, weight-gradient splits) depends on per-episode quantities only, so
    # chunking changes nothing but the order of the final sum over episodes
    assert rel_l2(parts["g_params"], [t.cpu() for t in whole["g_params"]]) <= 1e-5
    whole1 = run_maml(ws, dev, ep, theta, Wf, bfin, 1, 0.05, False, chunk=5)
    whole1 = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in whole1.items() if v is not None}
    parts1 = run_maml(ws, dev, ep, theta, Wf, bfin, 1, 0.05, False, chunk=3)
    assert torch.equal(parts1["logits"], whole1["logits"]) and torch.equal(parts1["loss_b"], whole1["loss_b"])
    assert rel_l2(parts1["g_params"], [t.cpu() for t in whole1["g_params"]]) <= 1e-5


def test_gradient_is_linear_in_grad_scale_and_eval_equals_train_forward(dev, ws):
    channels, H, N = (32, 64), 16, 3
    ep, theta, _ = case(3, 2, N, 2, 2, H, channels)
    Wf, bfin = head_of(N, channels[-1])
    a = run_maml(ws, dev, ep, theta, Wf, bfin, 1, 0.05, False, grad_scale=0.5)
    ga, la = [t.clone() for t in a["g_params"]], a["logits"].clone()
    b = run_maml(ws, dev, ep, theta, Wf, bfin, 1, 0.05, False, grad_scale=1.0)
    for x, y in zip(ga, b["g_params"]):
        assert torch.allclose(2 * x, y, rtol=1e-5, atol=1e-8)
    e = run_maml(ws, dev, ep, theta, Wf, bfin, 1, 0.05, False, need_grad=False)
    assert torch.equal(e["logits"], la)


def test_full_size_20way_episode_pair_against_the_stored_sweep(dev, ws):
    """BASELINE.json configs[4]'s episode shape at full size: 20-way 5-shot, 3 x 84 x 84 images, channels 64 / 160 / 320 / 640, one
    inner step, second-order FuMI meta-gradient, two episodes.  The expected values were produced by oracle/gen_resnet12_golden.py
    (minutes of host time); inputs are regenerated from its seeds.  A COARSE net (a bf16 chain this deep decorrelates: the bounds are
    the measured drift); the sharp checks of the same kernels at the same widths are the teacher-forced stage checks of
    tests/test_resnet12_probe.py, and the true T = 5 / Q = 15 shape runs there through size-independent properties."""
    from fumi_amd import hip
    from oracle import gen_resnet12_golden as G
    path = os.path.join(GOLDEN, "resnet12_20way.npz")
    gold = dict(np.load(path))
    ep, theta, phi = G.case()
    out = hip.fumi_resnet12_step(ws, G.N, ep["x_s"].to(dev), ep["y_s"].to(dev), ep["x_q"].to(dev), ep["y_q"].to(dev),
                                 [t.to(dev) for t in theta], [t.to(dev) for t in phi], G.T, G.ALPHA, False, text_s=ep["text_s"].to(dev))
    assert ws.read_status() == 0
    lg = out["logits"].cpu().double()
    for form, tol in (("bf16", 0.1), ("f32", 0.15)):
        ref = torch.from_numpy(gold[f"{form}.logits"]).double()
        assert rel(lg, ref) <= tol, form
        assert rel(out["loss_b"].cpu().double(), torch.from_numpy(gold[f"{form}.loss_b"]).double()) <= tol
    # meta-gradients: norms and strided samples of every tensor against the bf16-rounded sweep's
    num = den = 0.0
    for i, g in enumerate(out["g_theta"]):
        d = G.digest(g.cpu())
        r = gold[f"bf16.g_theta.{i}"]
        assert abs(d[0] - r[0]) <= 0.5 * r[0], f"norm of g_theta[{i}]"
        num += float(((d[2:] - r[2:]) ** 2).sum()); den += float((r[2:] ** 2).sum())
    assert (num / den) ** 0.5 <= 0.9
    assert rel_l2(out["g_phi"], [torch.from_numpy(gold[f"bf16.g_phi.{i}"]) for i in range(4)]) <= 0.5


# ---- the module surface (--im_encoder resnet12) on the GPU ----------------------------------------------------------------------
def test_fumi_resnet12_evaluate_applies_the_engines_gradient(dev, ws):
    """FUMI(im_encoder='resnet12').evaluate on the HIP engine: the plumbing around the step (parameter order of the 48 encoder
    tensors and the four hypernetwork tensors, gradient views, the loss / accuracy read-back, the optimizer step) -- one SGD
    training step must move every parameter by -lr x the meta-gradient a direct call of fumi_hip_fumi_resnet12_step returns for the
    same inputs (same kernels: compared at fp32 round-off), and the host oracle (fp32 autograd, no bf16 rounding) must see the same
    first-batch loss within bf16's forward error."""
    from types import SimpleNamespace
    from oracle import casegen as cg
    from fumi_amd import engine, hip
    from fumi_amd.models.fumi import FUMI
    from oracle_engine import OracleEngine
    N, Dt = 5, 12
    ep = CR.make_image_episodes(8, 3, N, 2, 3, 3, 16, 16, Dt)
    batch = cg.to_batch(ep)
    torch.manual_seed(1)
    m = FUMI(n_way=N, im_encoder="resnet12", image_size=16, text_emb_dim=Dt, text_hid_dim=16, norm_hypernet=False).to(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=1, num_test_adapt_steps=1, step_size=0.05, first_order=False, num_ways=N,
                           batch_size=3)
    theta, phi, _ = m._step_params(False)
    theta0, phi0 = [t.detach().clone() for t in theta], [t.detach().clone() for t in phi]
    state0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    direct = hip.fumi_resnet12_step(ws, N, ep["x_s"].to(dev), ep["y_s"].to(dev), ep["x_q"].to(dev), ep["y_q"].to(dev), theta0, phi0, 1, 0.05,
                                    False, text_s=ep["text_s"].to(dev))
    g_direct = [g.clone() for g in direct["g_theta"] + direct["g_phi"]]
    loss_direct = float(direct["loss_b"].mean())
    lr = 0.02
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    tr = m.evaluate(args, batch, opt, "train")
    assert ws.read_status() == 0
    assert abs(float(tr[0]) - loss_direct) <= 1e-5 * max(1.0, abs(loss_direct))
    theta1, phi1, _ = m._step_params(False)
    for i, (p0, p1, g) in enumerate(zip(theta0 + phi0, list(theta1) + list(phi1), g_direct)):
        upd = (p1.detach() - p0).cpu()
        assert torch.allclose(upd, -lr * g.cpu(), rtol=1e-4, atol=1e-7 + 1e-5 * float(g.abs().max()) * lr), f"parameter {i}"
    te = m.evaluate(args, batch, None, "test")
    assert np.isfinite(float(te[0])) and te[2].shape == (3, N * 3)
    # the host oracle on the same initial parameters: forward agreement at bf16 level
    old = engine.set_engine(OracleEngine())
    try:
        torch.manual_seed(1)
        mc = FUMI(n_way=N, im_encoder="resnet12", image_size=16, text_emb_dim=Dt, text_hid_dim=16, norm_hypernet=False)
        mc.load_state_dict({k: v.cpu() for k, v in state0.items()})
        argc = SimpleNamespace(**{**vars(args), "device": torch.device("cpu")})
        ref = mc.evaluate(argc, batch, None, "test")
    finally:
        engine.set_engine(old)
    m.load_state_dict(state0)
    te0 = m.evaluate(args, batch, None, "test")
    assert abs(float(te0[0]) - float(ref[0])) <= 0.05 * max(1.0, abs(float(ref[0])))


def test_cli_fumi_resnet12_end_to_end_on_gpu(dev, tmp_path, monkeypatch):
    """`python -m fumi_amd.main --model fumi --im_encoder resnet12 --dataset synthetic` (BASELINE.json configs[4]'s model, shortened):
    image loader -> ResNet-12 (bf16) + hypernetwork -> meta-training on the HIP engine -> checkpoint -> test."""
    from fumi_amd import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ["--model", "fumi", "--dataset", "synthetic", "--im_encoder", "resnet12", "--image_size", "16", "--text_encoder", "BERT",
            "--text_emb_dim", "32", "--batch_size", "8", "--num_shots", "5", "--num_ways", "5", "--num_shots_test", "5",
            "--epochs", "200", "--eval_freq", "100", "--num_ep_test", "16", "--num_train_adapt_steps", "1",
            "--num_test_adapt_steps", "1", "--lr", "1e-3", "--step_size", "0.01", "--dropout", "0", "--log_dir", str(tmp_path / "res"),
            "--synthetic_classes", "16", "--wandb_offline"]
    args = cli.parse_args(argv)
    assert args.device.type == "cuda"
    res = cli.main(args)
    assert np.isfinite(res["test_loss"]) and 0.0 <= res["test_acc"] <= 1.0
    # chance = 0.2.  200 meta-steps take the training episodes to loss 0.09 / accuracy 0.98 (16 synthetic classes: the 12-layer
    # encoder memorises them); on the held-out classes that leaves 0.30-0.32
    assert res["test_acc"] > 0.25
