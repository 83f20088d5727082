"""Sharp GPU checks of the bf16 ResNet-12 meta-step (BASELINE.json configs[4]) through the C ABI: every stage of the sweep, teacher-forced.

tests/test_resnet12_gpu.py compares whole steps with the bf16-rounded sweep and has to carry the decorrelation of a bf16 chain as its
bound.  Here ``fumi_hip_rn12_probe`` hands out every stored intermediate of a step and tests/rn12_stages.py checks each stage on the
engine's own upstream values (2^-7 of the maximum / 2e-3 relative L2 per map, 2e-4 per fp32 sum), for

  * the true channel widths 64 / 160 / 320 / 640 (every kernel's 160-, 320- and 640-channel forms: nch = 20 / 40 / 80 row groups,
    2.5-tile weight gradients), four blocks deep, two inner steps, second order, both Hessian-vector products;
  * five inner steps (the reference's default ``num_train_adapt_steps``, fumi/utils/utils.py:171-175) on a two-block net;
  * the FuMI form: the hypernetwork's gradients from the engine's head adjoints (fumi/models/fumi.py:104-113,198-212);

and the TRUE configs[4] episode shape (20-way 5-shot, 15 queries per class, 84 x 84, T = 5, second order) through size-independent
properties: chunks and lanes are bit-neutral, the gradient is linear in ``grad_scale``, evaluation equals the training forward.

"Parity unpinned": the reference has only the seam (fumi/models/fumi.py:89-100); the oracle is oracle/resnet12_manual.py.
"""
import numpy as np
import pytest
import torch

from oracle import conv4_ref as CR
from oracle import fumi_ref as R
from oracle import resnet12_ref as RR

import rn12_stages as ST

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ws(dev):
    from fumi_amd import hip
    return hip.Workspace.get(dev)


def _case(seed, B, N, K, Q, H, channels, Dt=6):
    ep = CR.make_image_episodes(seed, B, N, K, Q, 3, H, H, Dt)
    theta = RR.make_params(seed, 3, channels, torch.float32)
    rs = np.random.RandomState(seed + 1)
    Wf = torch.from_numpy((rs.standard_normal((N, channels[-1])) * 0.1).astype(np.float32))
    bfin = torch.from_numpy((rs.standard_normal(N) * 0.05).astype(np.float32))
    return ep, theta, Wf, bfin


def _maml_runner(hip, ws, dev, ep, theta, Wf, bfin, T, alpha):
    d = {k: ep[k].to(dev) for k in ("x_s", "y_s", "x_q", "y_q")}
    params = [t.to(dev) for t in theta + [Wf, bfin]]
    return lambda: hip.maml_resnet12_step(ws, d["x_s"], d["y_s"], d["x_q"], d["y_q"], params, T, alpha, False)


def _report(name, chk):
    """FUMI_RN12_STAGE_REPORT=<dir>: the stage table of a case (worst stages first) for profiles/ and DESIGN.md."""
    import json
    import os
    d = os.environ.get("FUMI_RN12_STAGE_REPORT")
    if not d:
        return
    os.makedirs(d, exist_ok=True)
    maps = [r for r in chk.rows if r[1] != r[2]]
    sums = [r for r in chk.rows if r[1] == r[2]]
    rep = dict(case=name, stages=len(chk.rows), failed=len(chk.bad), map_stages=len(maps), sum_stages=len(sums),
               worst_map_max=max(r[1] for r in maps), worst_map_l2=max(r[2] for r in maps), worst_sum=max(r[1] for r in sums),
               worst_maps=[dict(stage=r[0], max=r[1], l2=r[2]) for r in sorted(maps, key=lambda r: -r[2])[:12]],
               worst_sums=[dict(stage=r[0], err=r[1]) for r in sorted(sums, key=lambda r: -r[1])[:12]])
    with open(os.path.join(d, f"rn12_stage_report_{name}.json"), "w") as f:
        json.dump(rep, f, indent=1)


def _check_meta_gradient(chk, lay, final, B, n_theta):
    """g_params = 1/B x the sum over episodes of the engine's own bar_0 (parameter order of the 12 tensors per block + lin_final)."""
    out, bar, barh = final
    gsum = bar.sum(0) / B
    for i, (g, r) in enumerate(zip(out["g_params"][:n_theta], lay.theta(gsum))):
        chk.vec(f"meta-gradient {i}", g.cpu().double().reshape(-1), r.reshape(-1), 2e-6)
    hsum = barh.sum(0) / B
    chk.vec("meta-gradient lin_final.weight", out["g_params"][-2].cpu().double(), hsum[:, :-1], 2e-6)
    chk.vec("meta-gradient lin_final.bias", out["g_params"][-1].cpu().double(), hsum[:, -1], 2e-6)


def test_every_stage_of_a_four_block_second_order_step_at_the_true_widths(dev, ws):
    """Channels 64 / 160 / 320 / 640, four blocks, T = 2, second order: all forward / backward / tangent stages of both inner steps,
    the query pass, both Hessian-vector products, every parameter-space update and the final meta-gradient."""
    from fumi_amd import hip
    channels, H, B, N, T, alpha = (64, 160, 320, 640), 16, 2, 3, 2, 0.05
    ep, theta, Wf, bfin = _case(31, B, N, 2, 2, H, channels)
    head0 = torch.cat([Wf, bfin[:, None]], 1).double()[None].expand(B, -1, -1)
    run = _maml_runner(hip, ws, dev, ep, theta, Wf, bfin, T, alpha)
    chk, lay, final = ST.check_step(hip, ws, dev, run, ep, theta, head0, channels, T, alpha, hvp_steps=[1, 0])
    _check_meta_gradient(chk, lay, final, B, len(theta))
    assert len(chk.rows) > 1500
    _report("four_blocks_T2", chk)
    chk.assert_ok()


def test_every_stage_with_five_inner_steps(dev, ws):
    """T = 5 (the reference's default number of training adaptation steps): five taped forward / backward passes, the parameter
    chain, and the Hessian-vector products of inner steps 4, 3 and 0."""
    from fumi_amd import hip
    channels, H, B, N, T, alpha = (32, 64), 8, 2, 3, 5, 0.05
    ep, theta, Wf, bfin = _case(32, B, N, 2, 2, H, channels)
    head0 = torch.cat([Wf, bfin[:, None]], 1).double()[None].expand(B, -1, -1)
    run = _maml_runner(hip, ws, dev, ep, theta, Wf, bfin, T, alpha)
    chk, lay, final = ST.check_step(hip, ws, dev, run, ep, theta, head0, channels, T, alpha, hvp_steps=[4, 3, 0])
    _check_meta_gradient(chk, lay, final, B, len(theta))
    _report("two_blocks_T5", chk)
    chk.assert_ok()


def test_odd_sizes_and_three_blocks(dev, ws):
    """Odd map sizes (pooling leaves a row / column uncovered: the rim cells of the join's backward), 96-channel layers
    (NF = 3 column groups), three blocks."""
    from fumi_amd import hip
    channels, H, B, N, T, alpha = (32, 96, 160), 14, 2, 4, 1, 0.05
    ep, theta, Wf, bfin = _case(33, B, N, 1, 2, H, channels)
    head0 = torch.cat([Wf, bfin[:, None]], 1).double()[None].expand(B, -1, -1)
    run = _maml_runner(hip, ws, dev, ep, theta, Wf, bfin, T, alpha)
    chk, lay, final = ST.check_step(hip, ws, dev, run, ep, theta, head0, channels, T, alpha, hvp_steps=[0])
    _check_meta_gradient(chk, lay, final, B, len(theta))
    _report("three_blocks_odd", chk)
    chk.assert_ok()


def test_fumi_form_hypernetwork_gradients_from_the_engines_head_adjoints(dev, ws):
    """FuMI: text rows -> class select -> hypernetwork -> heads (fumi.py:104-113,198-212).  The heads the episodes start from and the
    gradients of the four hypernetwork tensors (the vector-Jacobian product of the engine's own head adjoints) at fp32 round-off;
    the encoder stages as above."""
    from fumi_amd import hip
    channels, H, B, N, T, alpha, Dt, Ht = (32, 64), 8, 3, 3, 1, 0.05, 6, 5
    ep, theta, _, _ = _case(34, B, N, 2, 2, H, channels, Dt)
    rs = np.random.RandomState(34)
    Fd = channels[-1]
    phi = [torch.from_numpy((rs.standard_normal(s) * 0.3).astype(np.float32)) for s in ((Ht, Dt), (Ht,), (Fd + 1, Ht), (Fd + 1,))]
    for tanh in (False, True):
        phis = [p.double().requires_grad_(True) for p in phi]
        heads = torch.stack([R.hyper_net(R.class_text_select(ep["text_s"][b].double(), ep["y_s"][b], N), phis, tanh) for b in range(B)])
        d = {k: ep[k].to(dev) for k in ("x_s", "y_s", "x_q", "y_q", "text_s")}
        th_d, phi_d = [t.to(dev) for t in theta], [t.to(dev) for t in phi]
        run = lambda: hip.fumi_resnet12_step(ws, N, d["x_s"], d["y_s"], d["x_q"], d["y_q"], th_d, phi_d, T, alpha, tanh, text_s=d["text_s"])
        chk, lay, final = ST.check_step(hip, ws, dev, run, ep, theta, heads.detach(), channels, T, alpha, hvp_steps=[0])
        out, bar, barh = final
        for i, (g, r) in enumerate(zip(out["g_theta"], lay.theta(bar.sum(0) / B))):
            chk.vec(f"meta-gradient {i}", g.cpu().double().reshape(-1), r.reshape(-1), 2e-6)
        g_phi = torch.autograd.grad((heads * barh).sum() / B, phis)
        floor = max(float(r.abs().max()) for r in g_phi)      # (the last bias's gradient is analytically 0 without tanh: soft-max shift)
        for i, (g, r) in enumerate(zip(out["g_phi"], g_phi)):
            chk.vec(f"g_phi {i} (tanh {tanh})", g.cpu().double(), r, 2e-5, scale=0.05 * floor)
        chk.assert_ok()


@pytest.mark.parametrize("T", [0, 2, 3])
def test_first_order_and_zero_step_forms(T, dev, ws):
    """MAML `--first_order` (maml.py:173-177) and T = 0: one reused tape and two alternating parameter slots instead of a tape per
    step -- the last support pass, the query pass with the slot it must read, and the meta-gradient = the query pass's own gradient."""
    from fumi_amd import hip
    channels, H, B, N, alpha = (32, 64, 96), 8, 2, 3, 0.05
    ep, theta, Wf, bfin = _case(36 + T, B, N, 2, 2, H, channels)
    head0 = torch.cat([Wf, bfin[:, None]], 1).double()[None].expand(B, -1, -1)
    d = {k: ep[k].to(dev) for k in ("x_s", "y_s", "x_q", "y_q")}
    params = [t.to(dev) for t in theta + [Wf, bfin]]
    run = lambda: hip.maml_resnet12_step(ws, d["x_s"], d["y_s"], d["x_q"], d["y_q"], params, T, alpha, True)
    chk, lay, final = ST.check_first_order(hip, ws, dev, run, ep, theta, head0, channels, T, alpha)
    _check_meta_gradient(chk, lay, final, B, len(theta))
    chk.assert_ok()


def test_probe_refuses_without_a_probe_mode_step(dev, ws):
    from fumi_amd import hip
    channels, H, B, N = (32,), 8, 2, 3
    ep, theta, Wf, bfin = _case(35, B, N, 2, 2, H, channels)
    _maml_runner(hip, ws, dev, ep, theta, Wf, bfin, 1, 0.05)()
    with pytest.raises(hip.FumiHipError):
        hip.rn12_probe(ws, dev, 0, 0, 0, 0)


# ---- the true configs[4] episode shape --------------------------------------------------------------------------------------------
def test_true_shape_chunks_lanes_scale_and_eval_are_consistent(dev, ws):
    """20-way 5-shot, 15 queries per class, 3 x 84 x 84 images, channels 64 / 160 / 320 / 640, T = 5, second order -- the workload
    bench.py's configs4 leg times, four episodes of it.  One chunk on one lane, two lanes of two episodes and four one-episode
    chunks on two lanes give the same forward bits and the same meta-gradient (an episode's launch geometry is independent of its
    chunk: only the order of the sum over episodes differs); the gradient is linear in grad_scale; the evaluation forward equals the
    training forward bit for bit; the status word stays 0."""
    from fumi_amd import hip
    B, N, K, Q, H, T, alpha, Dt, Ht = 4, 20, 5, 15, 84, 5, 0.01, 16, 8
    g = torch.Generator().manual_seed(77)
    x_s = torch.randn(B, N * K, 3, H, H, generator=g)
    x_q = torch.randn(B, N * Q, 3, H, H, generator=g)
    y_s = torch.stack([torch.arange(N).repeat_interleave(K)[torch.randperm(N * K, generator=g)] for _ in range(B)])
    y_q = torch.stack([torch.arange(N).repeat_interleave(Q)[torch.randperm(N * Q, generator=g)] for _ in range(B)])
    cls_text = torch.randn(B, N, Dt, generator=g)
    text_s = torch.gather(cls_text, 1, y_s[..., None].expand(-1, -1, Dt))
    theta = RR.make_params(77, 3, RR.CHANNELS, torch.float32)
    rs = np.random.RandomState(77)
    Fd = RR.CHANNELS[-1]
    phi = [torch.from_numpy((rs.standard_normal(s) * sc).astype(np.float32))
           for s, sc in (((Ht, Dt), 0.3), ((Ht,), 0.1), ((Fd + 1, Ht), 0.05), ((Fd + 1,), 0.02))]
    d = dict(x_s=x_s.to(dev), y_s=y_s.to(dev), x_q=x_q.to(dev), y_q=y_q.to(dev), text_s=text_s.to(dev))
    th_d, phi_d = [t.to(dev) for t in theta], [t.to(dev) for t in phi]

    def run(**kw):
        out = hip.fumi_resnet12_step(ws, N, d["x_s"], d["y_s"], d["x_q"], d["y_q"], th_d, phi_d, T, alpha, False, text_s=d["text_s"], **kw)
        torch.cuda.synchronize()
        assert ws.read_status() == 0
        return {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in out.items() if v is not None}

    def gdiff(a, b):
        num = sum(float(((x.double() - y.double()) ** 2).sum()) for x, y in zip(a, b))
        return (num / sum(float((y.double() ** 2).sum()) for y in b)) ** 0.5

    one = run(chunk=B)                                   # one chunk, one lane
    assert torch.isfinite(one["loss_b"]).all() and all(torch.isfinite(t).all() for t in one["g_theta"] + one["g_phi"])
    assert 0.5 < float(one["loss_b"].mean()) < 20.0 and sum(float(t.abs().sum()) for t in one["g_theta"]) > 0
    two = run()                                          # two lanes of two episodes
    four = run(chunk=1)                                  # four chunks on two lanes
    for o in (two, four):
        assert torch.equal(o["logits"], one["logits"]) and torch.equal(o["preds"], one["preds"]) and torch.equal(o["loss_b"], one["loss_b"])
        assert gdiff(o["g_theta"], one["g_theta"]) <= 1e-5 and gdiff(o["g_phi"], one["g_phi"]) <= 1e-5
    half = run(grad_scale=0.5 / B)
    for x, y in zip(half["g_theta"] + half["g_phi"], two["g_theta"] + two["g_phi"]):
        assert torch.allclose(2 * x, y, rtol=1e-5, atol=1e-9)
    ev = run(need_grad=False)
    assert torch.equal(ev["logits"], one["logits"]) and torch.equal(ev["loss_b"], one["loss_b"])
    # second order matters at this shape: the first-order gradient (MAML form shares the encoder path) is a different vector
    assert float(one["acc_b"].min()) >= 0.0 and float(one["acc_b"].max()) <= 1.0
