"""GPU parity tests: the HIP path, called through the C ABI (fumi_amd/hip.py -> libfumi_hip.so), against
(a) the golden vectors produced by the real reference and (b) the oracle restatement on the same seeded inputs.

Tolerances (BASELINE.md / SURVEY.md 7.3): integer predictions bit-exact (rows whose top-1/top-2 margin is below
10x the fp32 noise floor are exempt -- none occur in these cases, the test asserts that too); fp32 logits and
loss within 1e-4 relative to max|logit| per case; meta-gradients within 1e-4 of max|grad| of the tensor (measured ~1e-6; floored at
5 % of the model-wide gradient scale for tensors that are analytically zero)."""
import numpy as np
import pytest
import torch

from oracle import casegen as cg
from oracle import fumi_ref as R
from helpers import load_golden, case_seed, rel_to_max, grad_floor, safe_margin_mask

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
GRAD_TOL = 1e-4
MARGIN = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ws(dev):
    from fumi_amd import hip
    return hip.Workspace.get(dev)


def _g(t, dev):
    return t.to(dev).contiguous()


def _check_preds(preds, ref_logits, ref_preds, allow_ties=False):
    """Bit-exact integer predictions on every row whose top-1/top-2 margin is above the fp32 noise floor.  Only the
    --hypernet_bias_init case has such near-ties (its head weight is exactly 0, so all rows share one logit vector)."""
    mask = safe_margin_mask(ref_logits, MARGIN)
    if not allow_ties:
        assert bool(mask.all()), "a case has a near-tie; pick another seed"
    assert float(mask.float().mean()) > 0.5
    assert torch.equal(preds.cpu()[mask], torch.as_tensor(ref_preds)[mask]), "integer predictions differ"


def _check_grads(names, got, gold, ref):
    floor = grad_floor(gold) if gold is not None else max(0.05 * max(float(r.abs().max()) for r in ref), 1e-5)
    for n, a, r in zip(names, got, ref):
        e = rel_to_max(a.cpu(), r, floor)
        assert e <= GRAD_TOL, f"grad {n}: rel-to-max error {e:.3e}"
        if gold is not None and ("grad." + n) in gold:
            e = rel_to_max(a.cpu(), gold["grad." + n], floor)
            assert e <= GRAD_TOL, f"grad {n} vs golden: {e:.3e}"


@pytest.mark.parametrize("name", list(cg.FUMI_CASES))
def test_fumi_step_matches_reference(name, dev, ws):
    from fumi_amd import hip
    c, gold = cg.FUMI_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    theta, phi = cg.make_fumi_params(seed, c["D"], c["hid"], c["Dt"], c["Ht"])
    if c["init_bias"]:
        phi[2], phi[3] = torch.from_numpy(gold["init_head_weight"]), torch.from_numpy(gold["init_head_bias"])
    out = hip.fumi_step_select(ws, c["N"], _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi],
                               c["T"], cg.ALPHA, c["tanh"])
    assert ws.read_status() == 0
    # (a) golden vectors from the real reference
    assert rel_to_max(out["logits"].cpu(), gold["logits_q"]) <= LOGIT_TOL
    _check_preds(out["preds"], gold["logits_q"], gold["preds"], allow_ties=c["init_bias"])
    assert abs(float(out["loss_b"].mean()) - float(gold["loss"])) <= LOGIT_TOL * max(1.0, abs(float(gold["loss"])))
    if not c["init_bias"]:
        assert abs(float(out["acc_b"].mean()) - float(gold["acc"])) < 1e-6
    # (b) oracle on the same inputs (all gradients in full)
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], c["T"], cg.ALPHA, c["tanh"])
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    names = [f"im_net.linear{i}.{k}" for i in range(len(c["hid"] or [])) for k in ("weight", "bias")]
    names += ["hyper_net.0.weight", "hyper_net.0.bias", "hyper_net.2.weight", "hyper_net.2.bias"]
    _check_grads(names, out["g_theta"] + out["g_phi"], gold, ref["g_theta"] + ref["g_phi"])


@pytest.mark.parametrize("name", list(cg.FUMI_CASES)[:3])
def test_fumi_eval_mode_no_grad(name, dev, ws):
    """task='test' (fumi.py:126-127,154): forward only; same logits as the gradient-carrying call."""
    from fumi_amd import hip
    c = cg.FUMI_CASES[name]
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    theta, phi = cg.make_fumi_params(seed, c["D"], c["hid"], c["Dt"], c["Ht"])
    args = (ws, c["N"], _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
            _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], c["T"], cg.ALPHA, c["tanh"])
    a = hip.fumi_step_select(*args, need_grad=True)
    b = hip.fumi_step_select(*args, need_grad=False)
    assert rel_to_max(b["logits"].cpu(), a["logits"].cpu()) <= 1e-6
    assert torch.equal(a["preds"].cpu(), b["preds"].cpu())


@pytest.mark.parametrize("name", list(cg.MAML_CASES))
def test_maml_step_matches_reference(name, dev, ws):
    from fumi_amd import hip
    c, gold = cg.MAML_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], 8)
    p = cg.make_maml_params(seed, c["D"], c["hid"], c["N"])
    out = hip.maml_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                        [_g(t, dev) for t in p], c["T"], cg.ALPHA, c["first_order"])
    assert ws.read_status() == 0
    assert rel_to_max(out["logits"].cpu(), gold["logits_q"]) <= LOGIT_TOL
    _check_preds(out["preds"], gold["logits_q"], gold["preds"])
    assert abs(float(out["loss_b"].mean()) - float(gold["loss"])) <= LOGIT_TOL * max(1.0, abs(float(gold["loss"])))
    pl = [t.clone().requires_grad_(True) for t in p]
    ref = R.maml_meta_step(pl, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["T"], cg.ALPHA, c["first_order"])
    names = [f"net.lin_{i}.{k}" for i in range(len(c["hid"] or [])) for k in ("weight", "bias")]
    names += ["net.lin_final.weight", "net.lin_final.bias"]
    _check_grads(names, out["g_params"], gold, ref["g_params"])


# ---- finer-grained ops -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 65, 50), (64, 64, 32), (160, 256, 768), (130, 67, 129), (800, 256, 2048)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fwd(M, N, K, act, dev, ws):
    from fumi_amd import hip
    g = torch.Generator().manual_seed(M * 1000 + N * 10 + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    y = hip.linear_fwd(ws, _g(x, dev), _g(W, dev), _g(b, dev), act).cpu()
    ref = x.double() @ W.double().t() + b.double()
    ref = [ref, torch.relu(ref), torch.tanh(ref)][act]
    assert rel_to_max(y, ref) <= 1e-5


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 65, 50), (160, 65, 256), (130, 67, 129), (5920, 256, 512)])
def test_linear_bwd(M, N, K, dev, ws):
    from fumi_amd import hip
    g = torch.Generator().manual_seed(M + N + K)
    x, W, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    dx = hip.linear_bwd_data(ws, _g(dy, dev), _g(W, dev)).cpu()
    dW, db = hip.linear_bwd_weight(ws, _g(dy, dev), _g(x, dev))
    assert rel_to_max(dx, dy.double() @ W.double()) <= 1e-5
    assert rel_to_max(dW.cpu(), dy.double().t() @ x.double()) <= 1e-5
    assert rel_to_max(db.cpu(), dy.double().sum(0)) <= 1e-5


def test_gemm_asymmetric_identity(dev, ws):
    """A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md section 3)."""
    from fumi_amd import hip
    n = 96
    Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 977) - 400.0
    y = hip.linear_fwd(ws, _g(torch.eye(n), dev), _g(Bm, dev), None, 0).cpu()      # I @ Bm^T
    assert torch.equal(y, Bm.t().contiguous())


def test_class_text_select(dev, ws):
    from fumi_amd import hip
    B, N, K, Dt = 3, 7, 4, 33
    ep = cg.make_episodes(5, B, N, K, 2, 8, Dt)
    out = hip.class_text_select(ws, _g(ep["text_s"], dev), _g(ep["y_s"], dev), N).cpu()
    ref = torch.stack([R.class_text_select(ep["text_s"][b], ep["y_s"][b], N) for b in range(B)])
    assert torch.equal(out, ref)
    assert ws.read_status() == 0
    y = ep["y_s"].clone()
    y[y == 2] = 3                                             # class 2 has no support sample -> reference raises
    hip.class_text_select(ws, _g(ep["text_s"], dev), _g(y, dev), N)
    with pytest.raises(IndexError):
        hip.raise_on_status(ws.read_status())


def test_glove_bag_matches_reference(dev, ws):
    from fumi_amd import hip
    gold = load_golden("wordemb")
    table, tokens = torch.from_numpy(gold["table"]).float(), torch.from_numpy(gold["tokens"])
    for mode in ("mean", "max"):
        out = hip.glove_bag(ws, _g(tokens, dev), _g(table, dev), int(gold["pad"]), mode).cpu()
        assert rel_to_max(out, gold[mode]) <= 1e-6, mode
    # GloVe-sized rows (E=300, L=128) against the oracle
    g = torch.Generator().manual_seed(0)
    V, E, L = 2000, 300, 128
    table = torch.rand(V, E, generator=g) * 2 - 1
    table[0] = 0
    tok = torch.randint(1, V, (2, 25, L), generator=g)
    for r in range(25):
        tok[:, r, 8 + r * 4:] = 0
    for mode in ("mean", "max"):
        out = hip.glove_bag(ws, _g(tok, dev), _g(table, dev), 0, mode).cpu()
        assert rel_to_max(out, R.word_embedding_pool(tok, table, 0, mode)) <= 1e-6
    with pytest.raises(NameError):
        hip.glove_bag(ws, _g(tok, dev), _g(table, dev), 0, "median")
    # fused select + bag (what FUMI.evaluate uses): first support row of each class, unsorted labels
    ep = cg.make_episodes(9, 2, 5, 5, 2, 8, 8, tokens=(V, L, 0))
    for mode in ("mean", "max"):
        out = hip.glove_bag_select(ws, _g(ep["text_s"], dev), _g(ep["y_s"], dev), 5, _g(table, dev), 0, mode).cpu()
        rows = torch.stack([R.class_text_select(ep["text_s"][b], ep["y_s"][b], 5) for b in range(2)])
        assert rel_to_max(out, R.word_embedding_pool(rows, table, 0, mode)) <= 1e-6
    assert ws.read_status() == 0


# ---- full-size property checks (BASELINE.json configs[1]) ----------------------------------------------------------
def test_fumi_full_size_against_oracle(dev, ws):
    """C2 sizes (5-way 5-shot, 32 query/class, D=2048, [256,64], GloVe-300 text, T=1) on a reduced meta-batch."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, Dt, Ht, T = 4, 5, 5, 32, 2048, [256, 64], 300, 256, 1
    ep = cg.make_episodes(77, B, N, K, Q, D, Dt)
    theta, phi = cg.make_fumi_params(77, D, hid, Dt, Ht)
    out = hip.fumi_step_select(ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, cg.ALPHA, False)
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, cg.ALPHA, False)
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    mask = safe_margin_mask(ref["logits"], MARGIN)
    assert torch.equal(out["preds"].cpu()[mask], ref["preds"][mask])
    _check_grads([str(i) for i in range(8)], out["g_theta"] + out["g_phi"], None, ref["g_theta"] + ref["g_phi"])


def test_fumi_configs1_glove_tokens_end_to_end_against_oracle(dev, ws):
    """BASELINE.json configs[1] as the bench runs it, on a reduced meta-batch: token ids [B,S,L=128] -> select + GloVe bag
    (fumi_hip_glove_bag_select) -> hypernetwork -> 1 inner step -> query loss -> all eight meta-gradients, against the oracle fed
    the same tokens (common.py:23-41 pooling of every support row, then fumi.py:207-210)."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, E, Ht, T, V, L = 4, 5, 5, 32, 2048, [256, 64], 300, 256, 1, 20000, 128
    ep = cg.make_episodes(311, B, N, K, Q, D, 1, tokens=(V, L, 0))
    theta, phi = cg.make_fumi_params(311, D, hid, E, Ht)
    table = torch.rand(V, E, generator=torch.Generator().manual_seed(311)) * 2 - 1
    table[0] = 0
    cls_text = hip.glove_bag_select(ws, _g(ep["text_s"], dev), _g(ep["y_s"], dev), N, _g(table, dev), 0, "mean")
    out = hip.fumi_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                        [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, cg.ALPHA, False, cls_text=cls_text)
    assert ws.read_status() == 0
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    text = R.word_embedding_pool(ep["text_s"], table, 0, "mean")
    ref = R.fumi_meta_step(th, ph, text, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, cg.ALPHA, False)
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    mask = safe_margin_mask(ref["logits"], MARGIN)
    assert float(mask.float().mean()) > 0.99 and torch.equal(out["preds"].cpu()[mask], ref["preds"][mask])
    _check_grads([str(i) for i in range(8)], out["g_theta"] + out["g_phi"], None, ref["g_theta"] + ref["g_phi"])


def test_fumi_configs1_full_meta_batch_against_oracle(dev, ws):
    """BASELINE.json configs[1] at the bench's own meta-batch (B = 32, 5-way 5-shot, 32 query / class, D = 2048, [256, 64],
    GloVe-300 text rows, T = 1): every logit, prediction, per-episode loss and all eight meta-gradients against the oracle
    (a fraction of a second of host time)."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, Dt, Ht, T = 32, 5, 5, 32, 2048, [256, 64], 300, 256, 1
    ep = cg.make_episodes(2024, B, N, K, Q, D, Dt)
    theta, phi = cg.make_fumi_params(2024, D, hid, Dt, Ht)
    out = hip.fumi_step_select(ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, cg.ALPHA, False)
    assert ws.read_status() == 0
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, cg.ALPHA, False)
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    mask = safe_margin_mask(ref["logits"], MARGIN)
    assert float(mask.float().mean()) > 0.99 and torch.equal(out["preds"].cpu()[mask], ref["preds"][mask])
    _check_grads([str(i) for i in range(8)], out["g_theta"] + out["g_phi"], None, ref["g_theta"] + ref["g_phi"])


def test_fumi_configs2_per_rank_meta_batch_against_oracle(dev, ws):
    """BASELINE.json configs[2]'s per-rank shape (32 episodes, 5-way 5-shot, 32 query / class, D = 2048, [256, 64], 768-d BERT text
    rows, T = 5 -- the reference's default number of training adaptation steps, fumi/utils/utils.py:171-175) against the oracle.  At
    this size the backward X-panel pass takes its two-launch form (query rows on the second stream beside the reverse sweep,
    csrc/episode.hip run_episodes): this is the in-process parity check of that form."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, Dt, Ht, T = 32, 5, 5, 32, 2048, [256, 64], 768, 256, 5
    ep = cg.make_episodes(2025, B, N, K, Q, D, Dt)
    theta, phi = cg.make_fumi_params(2025, D, hid, Dt, Ht)
    out = hip.fumi_step_select(ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, cg.ALPHA, False)
    assert ws.read_status() == 0
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, cg.ALPHA, False)
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    mask = safe_margin_mask(ref["logits"], MARGIN)
    assert float(mask.float().mean()) > 0.99 and torch.equal(out["preds"].cpu()[mask], ref["preds"][mask])
    # 32 x 185 rows x 256 units x 6 passes are ~9 M ReLU decisions: a layer-0 pre-activation within fp32 rounding of zero can fall on
    # the other side than in the host's arithmetic (the alternative forward X-panel kernels round A0 differently, tools/run_env_forms.sh:
    # FUMI_XP_PS=0 moves unit 148).  Such a flip leaves the logits alone and shifts ONE hidden unit's row of gW0, its entry of gb0 and
    # its column of gW1 by that row's adjoint: at most two such units are taken out of the comparison, everything else is held to GRAD_TOL.
    got, want = [t.cpu().clone() for t in out["g_theta"] + out["g_phi"]], [t.clone() for t in ref["g_theta"] + ref["g_phi"]]
    floor = max(0.05 * max(float(r.abs().max()) for r in want), 1e-5)
    unit_err = (got[0] - want[0]).abs().max(1)[0] / max(float(want[0].abs().max()), floor)
    flipped = (unit_err > GRAD_TOL).nonzero().flatten().tolist()
    assert len(flipped) <= 2, flipped
    for u in flipped:
        for t in (got, want):
            t[0][u] = 0; t[1][u] = 0; t[2][:, u] = 0
    _check_grads([str(i) for i in range(8)], got, None, want)
    # the same inputs again: the two streams leave no run-to-run difference (fixed summation order in every part)
    out2 = hip.fumi_step_select(ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                                _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, cg.ALPHA, False)
    for a, b in zip(out["g_theta"] + out["g_phi"], out2["g_theta"] + out2["g_phi"]):
        assert torch.equal(a, b)


def test_expired_sibling_wait_sets_the_status_bit(dev, ws):
    """The split reverse sweep waits for its sibling workgroups with a bounded spin; an expired wait must be reported
    (FUMI_ST_SYNC_TIMEOUT -> RuntimeError), not carried on from silently.  With a limit of 0 polls every wait that is not
    already satisfied expires."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, Dt, Ht, T = 16, 5, 5, 32, 2048, [256, 64], 300, 256, 1
    ep = cg.make_episodes(9, B, N, K, Q, D, Dt)
    theta, phi = cg.make_fumi_params(9, D, hid, Dt, Ht)
    args = (ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev), _g(ep["text_s"], dev),
            [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, cg.ALPHA, False)
    ws.read_status()
    old = hip.lib().fumi_hip_set_spin_limit(0)
    st = 0
    try:
        # (a wait expires only if a sibling part has not arrived yet when it is first polled -- 64 workgroups exchanging partial sums
        # make that all but certain in one step, but it is a race: up to 50 steps)
        for _ in range(50):
            hip.fumi_step_select(*args)
            st |= ws.read_status()
            if st & hip.ST_SYNC_TIMEOUT:
                break
    finally:
        hip.lib().fumi_hip_set_spin_limit(old)
    assert st & hip.ST_SYNC_TIMEOUT, "no wait expired with a limit of 0 polls (is the split sweep in use?)"
    with pytest.raises(RuntimeError):
        hip.raise_on_status(st)
    good = hip.fumi_step_select(*args)                                  # the default limit: a clean step again
    assert ws.read_status() == 0 and bool(torch.isfinite(good["g_theta"][0]).all())


@pytest.mark.parametrize("which", ["label_range", "class_missing"])
def test_invalid_episode_raises_index_error_in_the_loops(which, dev, ws):
    """The reference raises IndexError for a class without a support sample (fumi.py:209) and for a label outside [0, N)
    (cross_entropy).  The kernels flag such an episode in the status word; test_loop reads it where it synchronises anyway."""
    from types import SimpleNamespace
    from fumi_amd.models import fumi as F
    c = cg.FUMI_CASES["fumi_t1"]
    ep = cg.make_episodes(5, 2, c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    torch.manual_seed(0)
    m = F.FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_emb_dim=c["Dt"], text_hid_dim=c["Ht"],
               norm_hypernet=False).to(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=1, num_test_adapt_steps=1, step_size=cg.ALPHA, first_order=False,
                           batch_size=2, num_ways=c["N"])
    ws.read_status()
    loss, acc, _, _ = F.test_loop(args, m, [cg.to_batch(ep)], 1)                       # a valid batch passes
    assert np.isfinite(float(loss))
    bad = {k: v.clone() for k, v in ep.items()}
    if which == "label_range":
        bad["y_q"][1, 3] = c["N"]
    else:
        bad["y_s"][bad["y_s"] == 2] = 3
    with pytest.raises(IndexError):
        F.test_loop(args, m, [cg.to_batch(bad)], 1)
    assert ws.read_status() == 0                                                     # the word was cleared by the check


def test_fumi_linearity_in_grad_scale_and_episode_sum(dev, ws):
    """Size-independent properties at the bench size (B=32): (1) gradients are linear in grad_scale; (2) the
    gradient of a meta-batch is the sum of the gradients of its two halves (what the multi-GPU sharding relies on);
    (3) per-episode outputs do not depend on which other episodes share the batch."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, Dt, Ht, T = 32, 5, 5, 32, 2048, [256, 64], 300, 256, 1
    ep = cg.make_episodes(123, B, N, K, Q, D, Dt)
    theta, phi = cg.make_fumi_params(123, D, hid, Dt, Ht)
    th, ph = [_g(t, dev) for t in theta], [_g(t, dev) for t in phi]

    def run(sl, scale):
        return hip.fumi_step_select(ws, N, _g(ep["x_s"][sl], dev), _g(ep["y_s"][sl], dev), _g(ep["x_q"][sl], dev),
                                    _g(ep["y_q"][sl], dev), _g(ep["text_s"][sl], dev), th, ph, T, cg.ALPHA, True,
                                    grad_scale=scale)
    full = run(slice(0, B), 1.0 / B)
    keep = lambda d: {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in d.items() if v is not None}
    full = keep(full)
    a = run(slice(0, B // 2), 1.0 / B)
    a = keep(a)
    b = run(slice(B // 2, B), 1.0 / B)
    assert torch.equal(torch.cat([a["preds"], b["preds"]]), full["preds"])
    assert rel_to_max(torch.cat([a["logits"], b["logits"]]).cpu(), full["logits"].cpu()) <= 1e-6
    for x, y, z in zip(a["g_theta"] + a["g_phi"], b["g_theta"] + b["g_phi"], full["g_theta"] + full["g_phi"]):
        assert rel_to_max((x + y).cpu(), z.cpu(), floor=1e-6) <= 1e-4
    dbl = run(slice(0, B), 2.0 / B)
    for x, z in zip(dbl["g_theta"] + dbl["g_phi"], full["g_theta"] + full["g_phi"]):
        assert rel_to_max(x.cpu(), 2 * z.cpu(), floor=1e-7) <= 1e-6


def test_engine_rejects_cpu_tensors(dev, ws):
    from fumi_amd import hip
    with pytest.raises(hip.FumiHipError):
        hip.linear_fwd(ws, torch.zeros(2, 2), torch.zeros(2, 2))


def test_engine_rejects_mismatched_shapes(dev, ws):
    """A width that does not match (an embedding file other than --im_emb_dim, a wrong --text_emb_dim) must raise on the host:
    the kernels would read past the end of the buffers (the reference raises a matmul shape error)."""
    from fumi_amd import hip
    c = cg.FUMI_CASES["fumi_t1"]
    ep = cg.make_episodes(1, 2, c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    theta, phi = cg.make_fumi_params(1, c["D"], c["hid"], c["Dt"], c["Ht"])
    th, ph = [_g(t, dev) for t in theta], [_g(t, dev) for t in phi]
    a = dict(x_s=_g(ep["x_s"], dev), y_s=_g(ep["y_s"], dev), x_q=_g(ep["x_q"], dev), y_q=_g(ep["y_q"], dev), text_s=_g(ep["text_s"], dev))
    run = lambda **k: hip.fumi_step_select(ws, c["N"], k.get("x_s", a["x_s"]), k.get("y_s", a["y_s"]), k.get("x_q", a["x_q"]),
                                           k.get("y_q", a["y_q"]), k.get("text_s", a["text_s"]), k.get("th", th), k.get("ph", ph),
                                           1, cg.ALPHA, False)
    run()
    for bad in (dict(x_q=a["x_q"][..., :-1].contiguous()), dict(text_s=a["text_s"][..., :-2].contiguous()),
                dict(y_q=a["y_q"][:, :-1].contiguous()), dict(th=[th[0], th[1], th[2][:, :-1].contiguous(), th[3]]),
                dict(ph=[ph[0], ph[1], ph[2][:-1].contiguous(), ph[3]])):
        with pytest.raises(hip.FumiHipError, match="expected shape"):
            run(**bad)
    p = [_g(t, dev) for t in cg.make_maml_params(1, c["D"], c["hid"], c["N"])]
    with pytest.raises(hip.FumiHipError, match="expected shape"):
        hip.maml_step(ws, a["x_s"], a["y_s"], a["x_q"][..., :-1].contiguous(), a["y_q"], p, 1, cg.ALPHA)


@pytest.mark.parametrize("name", list(cg.AM3_CASES))
def test_am3_step_matches_reference(name, dev, ws):
    from fumi_amd import hip
    c, gold = cg.AM3_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    w = cg.make_am3_params(seed, c["D"], c["Dt"], c["Ht"], c["P"])
    wl = [w[k] for k in hip.AM3_KEYS]
    out = hip.am3_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                       _g(ep["text_s"], dev), [_g(t, dev) for t in wl], c["N"], c["lamda_fixed"])
    assert ws.read_status() == 0
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    ref = R.am3_step(wr, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], c["lamda_fixed"])
    assert abs(float(out["loss"]) - float(gold["loss"])) <= LOGIT_TOL * max(1.0, abs(float(gold["loss"])))
    nq = c["B"] * c["N"] * c["Q"]
    assert abs(float(out["correct"]) / nq - float(gold["acc"])) < 1e-6
    # integer predictions: bit-exact where the two nearest prototypes are not within round-off of each other
    d2 = ref["dist"].transpose(1, 2).topk(2, dim=-1, largest=False)[0]
    mask = (d2[..., 1] - d2[..., 0]) > 1e-4 * d2[..., 1].abs().clamp_min(1.0)
    assert float(mask.float().mean()) > 0.95
    assert torch.equal(out["preds"].cpu()[mask], ref["preds"][mask])
    assert rel_to_max(out["lamda_s"].cpu(), ref["lamda_s"]) <= 1e-5
    assert abs(float(out["lamda_s"].mean()) - float(gold["avg_lamda"])) < 1e-5
    sd = {"Wi": "image_encoder.weight", "bi": "image_encoder.bias", "G0": "g.0.weight", "g0": "g.0.bias",
          "G1": "g.3.weight", "g1": "g.3.bias", "H0": "h.0.weight", "h0": "h.0.bias", "H1": "h.3.weight", "h1": "h.3.bias"}
    _check_grads([sd[k] for k in hip.AM3_KEYS], out["grads"], gold, [ref["grads"][k] for k in hip.AM3_KEYS])


def test_am3_eval_mode_matches_train_forward(dev, ws):
    from fumi_amd import hip
    c = cg.AM3_CASES["am3_lam"]
    ep = cg.make_episodes(3, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    wl = [_g(cg.make_am3_params(3, c["D"], c["Dt"], c["Ht"], c["P"])[k], dev) for k in hip.AM3_KEYS]
    args = (ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev), _g(ep["text_s"], dev), wl, c["N"], None)
    a = hip.am3_step(*args, need_grad=True)
    la, pa = float(a["loss"]), a["preds"].clone()
    b = hip.am3_step(*args, need_grad=False)
    assert abs(la - float(b["loss"])) < 1e-6 and torch.equal(pa, b["preds"])


def test_am3_device_metrics_match_host_metrics(dev, ws):
    """AM3.evaluate (train / val) takes accuracy and macro precision / recall / F1 from the confusion counts the step leaves on
    the device (fumi_hip_am3_metrics); the host form (what sklearn returns, utils.py:319-326) is the checker, including classes
    that never occur.  Then end to end: AM3.evaluate('train') on the GPU vs the host metrics of its own predictions."""
    from fumi_amd import hip
    from fumi_amd.utils.utils import macro_metrics
    rs = np.random.RandomState(0)
    for N, n in [(5, 800), (5, 7), (20, 300), (3, 50), (64, 5000)]:
        t = rs.randint(0, N, size=n); p = rs.randint(0, N, size=n)
        if N == 20:
            p[p == 3] = 4; t[t == 7] = 8; p[p == 7] = 8                 # a class never predicted, a class absent everywhere
        conf = np.zeros((N, N), dtype=np.float32)
        np.add.at(conf, (t, p), 1.0)
        stats = torch.from_numpy(np.concatenate([[1.25, 0.0, 0.375], conf.ravel()]).astype(np.float32)).to(dev)
        got = hip.am3_metrics(ws, N, stats).cpu().numpy()
        np.testing.assert_allclose(got[1:5], np.array(macro_metrics(t, p)), rtol=3e-6, atol=1e-7)
        assert got[0] == 1.25 and got[5] == 0.375
    # end to end on a golden AM3 case
    from fumi_amd.models.am3 import AM3
    from fumi_amd.optim import Adam
    name = "am3_default"
    c = cg.AM3_CASES[name]
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    torch.manual_seed(3)
    m = AM3(im_encoder="precomputed", im_emb_dim=c["D"], text_encoder="BERT", text_emb_dim=c["Dt"], text_hid_dim=c["Ht"],
            prototype_dim=c["P"], dropout=0.0, fine_tune=False, dictionary=None, pooling_strat="mean",
            lamda_fixed=c.get("lamda_fixed")).to(dev)
    opt = Adam(m.parameters(), lr=1e-3)
    batch = cg.to_batch(ep)
    with torch.no_grad():
        ref = m.evaluate(batch, None, None, c["N"], dev, "test")                      # host metrics of the same predictions
    got = m.evaluate(batch, opt, None, c["N"], dev, "train")
    assert len(got) == 6
    np.testing.assert_allclose([float(x) for x in got], [float(x) for x in ref[:6]], rtol=2e-5, atol=1e-6)


def test_fused_adam_matches_torch_adam(dev, ws):
    """fumi_hip_adam_step == torch.optim.Adam(lr, weight_decay) (utils.py:280-283) over several steps, odd sizes included."""
    from fumi_amd.optim import Adam
    g = torch.Generator().manual_seed(0)
    shapes = [(256, 2048), (256,), (64, 256), (65,), (7, 3), (1,)]
    pa = [torch.randn(*s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa, ob = Adam(pa, lr=3e-3, weight_decay=5e-4), torch.optim.Adam(pb, lr=3e-3, weight_decay=5e-4)
    for it in range(4):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(dev)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert rel_to_max(x.detach().cpu(), y.detach().cpu()) <= 2e-7
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys() and set(sa["state"][0]) == set(sb["state"][0])
    assert float(sa["state"][0]["step"]) == float(sb["state"][0]["step"]) == 4.0
    ob.load_state_dict(sa)                                     # checkpoints interchange with torch.optim.Adam


def test_step_stats_output(dev, ws):
    """stats = grad_scale * (sum loss_b, sum acc_b), what FUMI.evaluate puts in the all-reduce buffer."""
    from fumi_amd import hip
    c = cg.FUMI_CASES["fumi_t1"]
    ep = cg.make_episodes(2, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    theta, phi = cg.make_fumi_params(2, c["D"], c["hid"], c["Dt"], c["Ht"])
    for need_grad in (True, False):
        stats = torch.zeros(2, device=dev)
        out = hip.fumi_step_select(ws, c["N"], _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                                   _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], c["T"],
                                   cg.ALPHA, False, need_grad=need_grad, grad_scale=0.25, stats=stats)
        assert abs(float(stats[0]) - 0.25 * float(out["loss_b"].sum())) < 1e-6
        assert abs(float(stats[1]) - 0.25 * float(out["acc_b"].sum())) < 1e-6


def test_fumi_evaluate_on_gpu_lazy_scalars(dev):
    """FUMI.evaluate end to end on the GPU: returned scalars behave like the reference's 0-d arrays."""
    from types import SimpleNamespace
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.optim import Adam
    name = "fumi_t5_tanh"
    c, gold = cg.FUMI_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    theta, phi = cg.make_fumi_params(seed, c["D"], c["hid"], c["Dt"], c["Ht"])
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="BERT", text_emb_dim=c["Dt"],
             text_hid_dim=c["Ht"], norm_hypernet=c["tanh"])
    m.load_state_dict(cg.fumi_state_dict(theta, phi))
    m.to(dev)
    opt = Adam(m.parameters(), lr=3e-5, weight_decay=5e-4)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=cg.ALPHA,
                           first_order=False, num_ways=c["N"], batch_size=c["B"])
    loss, acc, preds, tgt = m.evaluate(args, cg.to_batch(ep), opt, "train")
    assert abs(float(loss) - float(gold["loss"])) < 1e-4 and abs(float(acc) - float(gold["acc"])) < 1e-6
    assert np.isfinite(np.asarray(loss)) and (loss < 10.0) and f"{loss:.3f}" == f"{float(loss):.3f}"
    assert np.array_equal(preds.cpu().numpy().astype(np.int64), gold["preds"])
    for n, p in m.named_parameters():                        # fused Adam step == the reference's torch.optim.Adam step
        np.testing.assert_allclose(cg.digest(p.detach().cpu())[3:], gold[f"post.{n}.digest"][3:], rtol=0, atol=3e-7)


def test_lazy_scalars_arrive_without_events(dev, ws):
    """fumi_hip_publish_scalars: the values and then the sequence word land in pinned host memory by system-scope stores;
    lazy.scalars polls the word.  40 outstanding reads (> the ring of 16) read back in order, late, and out of order."""
    from fumi_amd import hip, lazy
    assert not lazy.USE_EVENT and not lazy.SYNC
    host = torch.zeros(16, dtype=torch.float32, pin_memory=True)
    src = torch.tensor([1.5, -2.25, 3.0], device=dev)
    hip.publish_scalars(ws, src, 3, host, 0xABCDEF0123456789)
    torch.cuda.synchronize()
    assert host[:3].tolist() == [1.5, -2.25, 3.0] and int(host.numpy().view(np.uint64)[7]) == 0xABCDEF0123456789
    assert host[3:14].abs().sum() == 0
    with pytest.raises(hip.FumiHipError):
        hip.publish_scalars(ws, src, 3, torch.zeros(16), 1)                       # not pinned
    vals = [torch.tensor([float(i), i + 0.5], device=dev) for i in range(40)]
    outs = [lazy.scalars(v, 2) for v in vals]
    for i in (39, 0, 17, 5):
        assert float(outs[i][0]) == float(i) and float(outs[i][1]) == i + 0.5
    assert [float(a) for a, _ in outs] == [float(i) for i in range(40)]
    # deferred form: nothing is issued until an optimizer launch carries it or flush() does
    d = torch.tensor([7.0, 8.0], device=dev)
    a, b = lazy.scalars(d, 2, defer=True)
    torch.cuda.synchronize()
    assert a._s._flag[0] != a._s._seq
    lazy.flush(dev)
    assert float(a) == 7.0 and float(b) == 8.0
    from fumi_amd.optim import Adam
    w = torch.nn.Parameter(torch.ones(300, device=dev)); w.grad = torch.full((300,), 0.5, device=dev)
    opt = Adam([w], lr=1e-2)
    a, b = lazy.scalars(d, 2, defer=True)
    opt.step()                                                                     # the fused Adam launch carries the stores
    torch.cuda.synchronize()
    assert a._s._flag[0] == a._s._seq and float(b) == 8.0
    lazy.flush(dev)                                                                 # nothing pending: no-op
    big = torch.arange(20, dtype=torch.float32, device=dev)                         # more than 14 values: copy + event form
    assert [float(x) for x in lazy.scalars(big, 20)] == list(range(20))


@pytest.mark.parametrize("name,p", [("fumi_t5", 0.25), ("fumi_3layer", 0.5), ("fumi_1shot", 0.1)])
def test_fumi_inner_loop_dropout_matches_oracle_with_same_masks(name, p, dev, ws):
    """Train-mode Dropout after every ReLU of im_net (fumi.py:93-99; CLI default 0.25): the engine's counter-based masks are
    regenerated on the host and fed to the oracle -- logits, loss and all second-order meta-gradients must agree."""
    from fumi_amd import hip
    from helpers import dropout_mask
    c = cg.FUMI_CASES[name]
    seed_ep = case_seed(name)
    ep = cg.make_episodes(seed_ep, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    theta, phi = cg.make_fumi_params(seed_ep, c["D"], c["hid"], c["Dt"], c["Ht"])
    seed = 0x1234_5678_9ABC_DEF1
    out = hip.fumi_step_select(ws, c["N"], _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi],
                               c["T"], cg.ALPHA, c["tanh"], dropout_p=p, seed=seed)
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], c["T"], cg.ALPHA, c["tanh"],
                           dropout=lambda b, call, layer, rows, width: dropout_mask(seed, p, b, call, layer, rows, width))
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    _check_grads([str(i) for i in range(len(theta) + 4)], out["g_theta"] + out["g_phi"], None, ref["g_theta"] + ref["g_phi"])
    # the masks really drop ~p of the active units: compare with the no-dropout run
    base = hip.fumi_step_select(ws, c["N"], _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                                _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi],
                                c["T"], cg.ALPHA, c["tanh"])
    assert rel_to_max(out["logits"].cpu(), base["logits"].cpu()) > 1e-3


def test_am3_dropout_matches_oracle_with_same_masks(dev, ws):
    """AM3 train-mode Dropout inside g and h (am3.py:82,88; reference default 0.7/CLI 0.25) with the engine's masks."""
    from fumi_amd import hip
    from helpers import dropout_mask_flat
    c = cg.AM3_CASES["am3_lam"]
    ep = cg.make_episodes(6, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    w = cg.make_am3_params(6, c["D"], c["Dt"], c["Ht"], c["P"])
    p, seed, Rs = 0.3, 0xABCDEF0123456789, c["B"] * c["N"] * c["K"]
    out = hip.am3_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev), _g(ep["text_s"], dev),
                       [_g(w[k], dev) for k in hip.AM3_KEYS], c["N"], None, dropout_p=p, seed=seed)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    masks = (dropout_mask_flat(seed, p, 1, Rs, c["Ht"]), dropout_mask_flat(seed, p, 2, Rs, c["Ht"]))
    ref = R.am3_step(wr, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], None, masks=masks)
    assert abs(float(out["loss"]) - float(ref["loss"])) <= LOGIT_TOL * max(1.0, abs(float(ref["loss"])))
    assert rel_to_max(out["lamda_s"].cpu(), ref["lamda_s"]) <= 1e-5
    _check_grads(list(hip.AM3_KEYS), out["grads"], None, [ref["grads"][k] for k in hip.AM3_KEYS])


def test_generic_episode_kernels_in_subprocess(dev):
    """The LDS-resident adapt / query / reverse kernels are taken whenever an episode fits the CU's LDS; larger problems use
    the generic global-memory kernels.  Re-run the reference-parity cases with FUMI_EPI_GLOBAL=1 (read once per process,
    hence the child process) so both forms stay covered."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, FUMI_EPI_GLOBAL="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_parity.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k",
                        "fumi_step_matches_reference or maml_step_matches_reference or dropout_matches_oracle or eval_mode_no_grad"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_two_launch_backward_xpanel_pass_in_subprocess(dev):
    """With two or more inner steps and a meta-batch of at least 2048 query rows the backward X-panel pass runs as two launches (its
    query-row part on a second stream beside the reverse sweep, FUMI_EPI_OVERLAP).  The reference-generated goldens are small
    meta-batches and take the one-launch form in this process; re-run them with FUMI_EPI_OVERLAP=2 (the two-launch form whatever
    the size; read once per process, hence the child process) so that every T >= 2 golden pins both forms."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, FUMI_EPI_OVERLAP="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_parity.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "fumi_step_matches_reference or maml_step_matches_reference"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


_SB_SCRIPT = r"""
import sys, json, torch
sys.path.insert(0, %r)
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
g = torch.Generator().manual_seed(5)
B, S, Qn, D, h0 = 3, 25, 43, 256, 96
x_s = torch.randn(B, S, D, generator=g).abs(); x_q = torch.randn(B, Qn, D, generator=g) * 3.0      # post-ReLU-like support rows
W0 = torch.randn(h0, D, generator=g) * 0.05
A0, G = hip.xpanel_fwd(ws, x_s.to(dev), x_q.to(dev), W0.to(dev))
X = torch.cat([x_s, x_q], 1).double()
A0r = X @ W0.double().T; Gr = X @ x_s.double().transpose(1, 2)
ea = float((A0.cpu().double() - A0r).abs().max() / A0r.abs().max())
eg = float((G.cpu().double() - Gr).abs().max() / Gr.abs().max())
print(json.dumps({"ea": ea, "eg": eg}))
"""


def test_xpanel_fwd_split_bf16_has_fp32_accuracy(dev):
    """Layer 0 / the Gram matrix run on the bf16 matrix pipe from exact three-way bf16 splits of the fp32 operands (default;
    FUMI_XP_SB=0 selects the fp32 MFMA kernel, xpanel.hip).  Its error against fp64 must not exceed the fp32 MFMA kernel's
    (both ~1e-7..1e-6 of the largest entry, far inside the 1e-4 parity tolerance).  The knob is read once per process, hence
    the child processes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    errs = {}
    for sb in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", _SB_SCRIPT % root], env=dict(os.environ, FUMI_XP_SB=sb), cwd=root,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        errs[sb] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("ea", "eg"):
        assert errs["0"][k] < 2e-6, errs
        assert errs["1"][k] < 2e-6, errs
        assert errs["1"][k] < 1.5 * errs["0"][k] + 5e-8, errs


@pytest.mark.parametrize("B,S,Qn,D,h0", [(3, 25, 43, 320, 128),      # 10 slabs (the pipelined loop's minimum + 1), ragged row tiles
                                         (9, 40, 100, 512, 256),     # two Gram column blocks (S > 32), a partly filled XCD group
                                         (2, 5, 15, 2048, 256)])     # fewer rows than one tile
def test_xpanel_fwd_presplit_column_operand_has_fp32_accuracy(dev, ws, B, S, Qn, D, h0):
    """h0 % 128 == 0 and >= 9 slabs of 32: W0 and the support rows are split once into bf16 planes in MFMA fragment order
    (xpanel_presplit_kernel) and the tiles only split X (xpanel_fwd_ps_kernel, xpanel.hip).  Same six piece products per fp32
    product as the per-tile split: the error against fp64 stays at the fp32 level, far inside the 1e-4 parity tolerance."""
    from fumi_amd import hip
    g = torch.Generator().manual_seed(B * 1000 + D)
    x_s = torch.randn(B, S, D, generator=g).abs(); x_q = torch.randn(B, Qn, D, generator=g) * 3.0
    W0 = torch.randn(h0, D, generator=g) * 0.05
    A0, G = hip.xpanel_fwd(ws, x_s.to(dev), x_q.to(dev), W0.to(dev))
    assert ws.read_status() == 0
    X = torch.cat([x_s, x_q], 1).double()
    A0r = X @ W0.double().T; Gr = X @ x_s.double().transpose(1, 2)
    assert float((A0.cpu().double() - A0r).abs().max() / A0r.abs().max()) < 2e-6
    assert float((G.cpu().double() - Gr).abs().max() / Gr.abs().max()) < 4e-6          # (all-positive support rows: truncated tails add up)
    # the planes are rebuilt every call: a changed weight must show
    A1, _ = hip.xpanel_fwd(ws, x_s.to(dev), x_q.to(dev), (2.0 * W0).to(dev))
    assert float((A1.cpu().double() - 2.0 * A0r).abs().max() / A0r.abs().max()) < 4e-6


def test_meta_batch_larger_than_the_chip_matches_chunks(dev, ws):
    """72 episodes at the reference sizes: the split reverse sweep launches 8*9*4 = 288 workgroups of one CU each (> 256
    CUs), so parts of an episode wait for partners that are dispatched later; T = 2 gives two exchange rounds.  Gradients
    must equal the sum over chunks of 24 episodes (each chunk fits the chip), and the call must not hang."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, Dt, Ht, T = 72, 5, 5, 32, 2048, [256, 64], 300, 256, 2
    ep = cg.make_episodes(11, B, N, K, Q, D, Dt)
    theta, phi = cg.make_fumi_params(11, D, hid, Dt, Ht)
    g = lambda t: t.to(dev).contiguous()
    th, ph = [g(t) for t in theta], [g(t) for t in phi]

    def run(lo, hi):
        o = hip.fumi_step_select(ws, N, g(ep["x_s"][lo:hi]), g(ep["y_s"][lo:hi]), g(ep["x_q"][lo:hi]), g(ep["y_q"][lo:hi]),
                                 g(ep["text_s"][lo:hi]), th, ph, T, cg.ALPHA, False, grad_scale=1.0)
        return [x.clone() for x in o["g_theta"] + o["g_phi"]], o["loss_b"].clone()

    full, loss_full = run(0, B)
    acc, losses = None, []
    for lo in range(0, B, 24):
        gs, lb = run(lo, lo + 24)
        losses.append(lb)
        acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
    assert torch.allclose(loss_full, torch.cat(losses), rtol=1e-5, atol=1e-6)
    floor = 0.05 * max(float(b.abs().max()) for b in acc)
    for a, b in zip(full, acc):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), floor)


def test_side_stream_overlap_in_subprocess(dev):
    """FUMI_OVERLAP=3 runs the hypernetwork forward / backward on the workspace's side stream beside the two X-panel passes
    (fork / join with events).  Same results required; the knob is read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_parity.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "fumi_step_matches_reference or larger_than_the_chip"],
                       env=dict(os.environ, FUMI_OVERLAP="3"), cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_adapt_split_over_column_parts_in_subprocess(dev):
    """FUMI_ADAPT_P=4: the inner loop's layer-0 columns split over four workgroups per episode that exchange the layer-1
    partial sums once per inner step (off by default).  Same parity required, including more episodes than the chip holds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_parity.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k",
                        "fumi_step_matches_reference or maml_step_matches_reference or larger_than_the_chip or full_size or dropout"],
                       env=dict(os.environ, FUMI_ADAPT_P="4"), cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("form", ["0", "1"])
def test_other_hypernet_forward_forms_in_subprocess(form, dev):
    """FUMI_HYPER_FWD=0: one launch per hypernetwork layer (what hypernetworks wider than 256 use); =1: one launch with a
    workgroup per row block; the default (2) splits layer 0's columns over workgroups.  Same golden / oracle parity required."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_parity.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "fumi_step_matches_reference or odd_shapes"],
                       env=dict(os.environ, FUMI_HYPER_FWD=form), cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_cli_fumi_synthetic_end_to_end_on_gpu(dev, tmp_path, monkeypatch):
    """`python -m fumi_amd.main --model fumi --dataset synthetic` (BASELINE.json configs[1] wording, shortened): parse -> loaders
    -> initial validation -> meta-training on the HIP engine -> checkpoint -> test.  The task is learnable: the loss must drop."""
    from fumi_amd import main as cli
    from fumi_amd.models import common
    monkeypatch.chdir(tmp_path)
    rs = np.random.RandomState(3)            # stands in for the GloVe download (no network): tokens tok1..tok499, 300-d
    words = [f"tok{i}" for i in range(1, 500)]
    common.register_word_vectors("glove", common.ArrayKeyedVectors(words, rs.standard_normal((499, 300)).astype(np.float32)))
    argv = ["--model", "fumi", "--dataset", "synthetic", "--text_encoder", "glove", "--text_emb_dim", "300", "--batch_size", "8",
            "--num_shots", "5", "--num_ways", "5", "--num_shots_test", "8", "--epochs", "40", "--eval_freq", "20",
            "--num_ep_test", "16", "--num_train_adapt_steps", "2", "--num_test_adapt_steps", "2", "--lr", "1e-3",
            "--dropout", "0.25", "--log_dir", str(tmp_path / "res"), "--synthetic_classes", "20", "--synthetic_vocab", "500",
            "--synthetic_seq_len", "16", "--wandb_offline"]
    args = cli.parse_args(argv)
    assert args.device.type == "cuda"
    res = cli.main(args)
    assert np.isfinite(res["test_loss"]) and 0.0 <= res["test_acc"] <= 1.0
    assert res["test_loss"] < 1.55          # below ln(5) = 1.609: the engine's gradients train the model


def test_fumi_eval_with_100_adapt_steps_matches_oracle(dev, ws):
    """--num_test_adapt_steps defaults to 100 (fumi/utils/utils.py:172-175): evaluation runs a long untaped inner loop.
    Checked against the oracle in float64.  (A long loop can walk a pre-activation through zero: with step size 0.05 this
    case has |z| = 1.8e-8 at step 85 of episode 3, where fp32 summation order decides the ReLU's side and implementations
    legitimately part ways by 1e-3; the reference's default step size keeps every |z| well away from rounding noise.)"""
    from fumi_amd import hip
    c = cg.FUMI_CASES["fumi_t1"]
    ep = cg.make_episodes(77, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    theta, phi = cg.make_fumi_params(77, c["D"], c["hid"], c["Dt"], c["Ht"])
    alpha, T = cg.ALPHA, 100
    out = hip.fumi_step_select(ws, c["N"], _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, alpha, False,
                               need_grad=False)
    d = torch.float64
    th = [t.clone().to(d).requires_grad_(True) for t in theta]; ph = [t.clone().to(d).requires_grad_(True) for t in phi]
    a = (th, ph, ep["text_s"].to(d), ep["x_s"].to(d), ep["y_s"], ep["x_q"].to(d), ep["y_q"], c["N"])
    ref = R.fumi_meta_step(*a, T, alpha, False, need_grad=False)
    ref0 = R.fumi_meta_step(*a, 0, alpha, False, need_grad=False)
    assert rel_to_max(ref["logits"], ref0["logits"]) > 0.02            # the adaptation is not a no-op
    assert rel_to_max(out["logits"].cpu().double(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu().double(), ref["loss_b"]) <= LOGIT_TOL


_FUZZ = [   # B  N  K  Q   D    hid              Dt  Ht  T  tanh   -- odd sizes on purpose: ragged tiles, 1-4 hidden layers, splits
    dict(B=5, N=3, K=2, Q=7, D=72, hid=[40, 12], Dt=20, Ht=24, T=3, tanh=False),
    dict(B=2, N=7, K=3, Q=5, D=130, hid=[36], Dt=12, Ht=16, T=2, tanh=True),
    dict(B=9, N=2, K=4, Q=33, D=256, hid=[128, 64, 32], Dt=64, Ht=64, T=2, tanh=False),
    dict(B=3, N=5, K=1, Q=3, D=64, hid=[16, 16, 16, 8], Dt=8, Ht=8, T=4, tanh=True),
    dict(B=4, N=4, K=6, Q=9, D=512, hid=[256, 32], Dt=100, Ht=128, T=1, tanh=False),
    dict(B=1, N=10, K=2, Q=4, D=96, hid=[64, 64], Dt=32, Ht=64, T=2, tanh=False),
    dict(B=17, N=5, K=5, Q=8, D=320, hid=[192, 48], Dt=300, Ht=256, T=1, tanh=False),
    # hypernetwork shapes: every (tiles per wave, contraction chunks) form of the fused forward kernel, and the per-layer one
    dict(B=3, N=5, K=2, Q=4, D=128, hid=[64, 32], Dt=768, Ht=256, T=1, tanh=True),
    dict(B=7, N=3, K=2, Q=5, D=96, hid=[32], Dt=400, Ht=192, T=2, tanh=False),
    dict(B=2, N=6, K=1, Q=3, D=64, hid=[48, 16], Dt=500, Ht=128, T=1, tanh=True),
    dict(B=4, N=5, K=1, Q=2, D=64, hid=[32], Dt=768, Ht=64, T=1, tanh=False),
    dict(B=3, N=4, K=2, Q=3, D=64, hid=[32], Dt=52, Ht=320, T=1, tanh=False),
]


@pytest.mark.parametrize("i", range(len(_FUZZ)))
def test_fumi_step_odd_shapes_match_oracle(i, dev, ws):
    """Shapes the golden cases do not have (ragged row / column tiles, 1-4 hidden layers, every split factor of the reverse
    sweep, LDS-resident and generic kernels as the sizes dictate): logits, losses, predictions and all gradients vs the oracle."""
    from fumi_amd import hip
    c = _FUZZ[i]
    ep = cg.make_episodes(1000 + i, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=bool(i & 1))
    theta, phi = cg.make_fumi_params(1000 + i, c["D"], c["hid"], c["Dt"], c["Ht"])
    out = hip.fumi_step_select(ws, c["N"], _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi],
                               c["T"], cg.ALPHA, c["tanh"])
    assert ws.read_status() == 0
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], c["T"], cg.ALPHA, c["tanh"])
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    safe = safe_margin_mask(ref["logits"], 1e-4 * float(ref["logits"].abs().max()))
    assert torch.equal(out["preds"].cpu()[safe], ref["preds"][safe])
    names = [f"im_net.linear{k}.{w}" for k in range(len(c["hid"])) for w in ("weight", "bias")]
    names += ["hyper_net.0.weight", "hyper_net.0.bias", "hyper_net.2.weight", "hyper_net.2.bias"]
    _check_grads(names, out["g_theta"] + out["g_phi"], None, ref["g_theta"] + ref["g_phi"])


@pytest.mark.parametrize("c", [
    dict(B=6, N=7, K=3, Q=5, D=130, hid=None, T=4, first_order=False),          # bare linear head, unaligned D, second order
    dict(B=33, N=5, K=5, Q=32, D=2048, hid=None, T=2, first_order=False),        # reference sizes, more episodes than 32
    dict(B=4, N=3, K=2, Q=9, D=64, hid=None, T=3, first_order=True),
    dict(B=5, N=6, K=2, Q=4, D=200, hid=[24, 24, 12], T=3, first_order=False),   # three hidden layers
    dict(B=3, N=4, K=3, Q=6, D=96, hid=[20], T=2, first_order=True),             # one hidden layer, first order
], ids=lambda c: f"hid{c['hid']}_T{c['T']}_{'fo' if c['first_order'] else 'so'}")
def test_maml_step_odd_shapes_match_oracle(c, dev, ws):
    from fumi_amd import hip
    ep = cg.make_episodes(2000 + c["D"], c["B"], c["N"], c["K"], c["Q"], c["D"], 8)
    p = cg.make_maml_params(2000 + c["D"], c["D"], c["hid"], c["N"])
    out = hip.maml_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                        [_g(t, dev) for t in p], c["T"], cg.ALPHA, c["first_order"])
    assert ws.read_status() == 0
    pl = [t.clone().requires_grad_(True) for t in p]
    ref = R.maml_meta_step(pl, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["T"], cg.ALPHA, c["first_order"])
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    assert rel_to_max(out["loss_b"].cpu(), ref["loss_b"]) <= LOGIT_TOL
    names = [f"net.lin_{i}.{k}" for i in range(len(c["hid"] or [])) for k in ("weight", "bias")] + ["net.lin_final.weight", "net.lin_final.bias"]
    _check_grads(names, out["g_params"], None, ref["g_params"])


# ---- beside the episodic path: CLIP baseline and the bi-LSTM text encoders --------------------------------
def test_clip_step_matches_reference(dev, ws):
    """fumi_hip_clip_step against the reference's own CLIP outputs (golden) and the oracle at the CLI's default sizes."""
    from fumi_amd import hip
    gold = load_golden("clip")
    w = [torch.from_numpy(gold[k]) for k in R.CLIP_KEYS]
    out = hip.clip_step(ws, _g(torch.from_numpy(gold["text"]), dev), _g(torch.from_numpy(gold["image"]), dev), [_g(t, dev) for t in w])
    assert rel_to_max(out["sim"].cpu(), gold["sim"]) <= 1e-5
    assert abs(float(out["loss"]) - float(gold["loss"])) <= 1e-5
    for k, g in zip(R.CLIP_KEYS, out["grads"]):
        assert rel_to_max(g.cpu(), gold["grad." + k], 1e-6) <= 1e-4, k
    zs = hip.clip_step(ws, _g(torch.from_numpy(gold["text"][:1]), dev), _g(torch.from_numpy(gold["image"][:5]), dev), [_g(t, dev) for t in w],
                       need_loss=False, need_grad=False)
    assert rel_to_max(zs["sim"].cpu(), gold["zero_shot"]) <= 1e-5
    g = torch.Generator().manual_seed(4)
    n, Dt, D, P = 61, 768, 2048, 512                                    # --text_emb_dim / --im_emb_dim / --clip_latent_dim defaults
    w = []
    for o, i in ((P, Dt), (P, P), (P, D), (P, P)):
        w += [torch.randn(o, i, generator=g) / i ** 0.5, torch.randn(o, generator=g) * 0.1]
    text, image = torch.randn(n, Dt, generator=g), torch.randn(n, D, generator=g)
    out = hip.clip_step(ws, _g(text, dev), _g(image, dev), [_g(t, dev) for t in w])
    ref = R.clip_step([t.clone().requires_grad_(True) for t in w], text, image)
    assert rel_to_max(out["sim"].cpu(), ref["sim"]) <= 1e-5 and abs(float(out["loss"]) - float(ref["loss"])) <= 1e-5
    for k, a, b in zip(R.CLIP_KEYS, out["grads"], ref["grads"]):
        assert rel_to_max(a.cpu(), b, 1e-7) <= 1e-3, k


def test_lstm_bidir_matches_reference(dev, ws):
    """fumi_hip_lstm_bidir against the reference's RNN / RnnHid outputs (golden) and the oracle at GloVe-sized rows."""
    from fumi_amd import hip
    gold = load_golden("rnn")
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [torch.from_numpy(gold["rnn." + n + sfx]) for sfx in ("", "_reverse") for n in names]
    tokens, table = torch.from_numpy(gold["tokens"]), torch.from_numpy(gold["embed.weight"])
    for use_cell, key in ((False, "rnn"), (True, "rnnhid")):
        out = hip.lstm_bidir(ws, _g(tokens, dev), _g(table, dev), [_g(t, dev) for t in w], 0, use_cell).cpu()
        assert rel_to_max(out, gold[key]) <= 1e-5, key
    g = torch.Generator().manual_seed(6)
    V, E, H, L = 500, 300, 150, 40
    table = torch.rand(V, E, generator=g) * 2 - 1
    table[0] = 0
    w = []
    for _ in range(2):
        w += [torch.randn(4 * H, E, generator=g) / E ** 0.5, torch.randn(4 * H, H, generator=g) / H ** 0.5,
              torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1]
    tok = torch.randint(1, V, (3, 7, L), generator=g)
    for r in range(7):
        tok[:, r, 1 + 5 * r:] = 0
    for use_cell in (False, True):
        out = hip.lstm_bidir(ws, _g(tok, dev), _g(table, dev), [_g(t, dev) for t in w], 0, use_cell).cpu()
        assert rel_to_max(out, R.lstm_encode(tok, table, w, 0, use_cell)) <= 2e-5
    assert ws.read_status() == 0


def test_fumi_20way_episode_shape_matches_oracle(dev, ws):
    """BASELINE.json configs[4]'s EPISODE shape on embeddings (20-way 5-shot, 15 query / class, T = 5; its ResNet-12 / bf16 encoder
    is not built): S = 100 support rows do not fit the LDS-resident phase kernels, so this is the parity test of the
    global-memory forms at the shape that takes them by itself (tests/dev/probe_20way.py also times it)."""
    from fumi_amd import hip
    B, N, K, Q, D, hid, Dt, Ht, T = 3, 20, 5, 15, 2048, [256, 64], 768, 256, 5
    ep = cg.make_episodes(7, B, N, K, Q, D, Dt)
    theta, phi = cg.make_fumi_params(7, D, hid, Dt, Ht)
    out = hip.fumi_step_select(ws, N, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                               _g(ep["text_s"], dev), [_g(t, dev) for t in theta], [_g(t, dev) for t in phi], T, 0.01, False)
    assert ws.read_status() == 0
    th = [t.clone().requires_grad_(True) for t in theta]
    ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, 0.01, False)
    assert rel_to_max(out["logits"].cpu(), ref["logits"]) <= LOGIT_TOL
    per = [(float((a.cpu() - b).abs().max()), float(b.abs().max())) for a, b in zip(out["g_theta"] + out["g_phi"], ref["g_theta"] + ref["g_phi"])]
    floor = 1e-3 * max(m for _, m in per)                 # (a gradient that is analytically zero has no relative error)
    assert max(e / max(m, floor) for e, m in per) <= 1e-3, per


_SBB_SCRIPT = r"""
import sys, json, torch
sys.path.insert(0, %r)
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
g = torch.Generator().manual_seed(9)
worst = 0.0
for B, S, Qn, D, h0 in [(5, 25, 43, 256, 256),            # 340 contraction rows: a ragged last slab
                        (3, 10, 21, 128, 512),            # two 256-row tiles of gW0, two 64-column tiles
                        (40, 25, 160, 192, 256)]:         # 7400 rows: many slabs, episodes straddling slab borders
    x_s = torch.randn(B, S, D, generator=g).abs(); x_q = torch.randn(B, Qn, D, generator=g) * 3.0
    Ab = torch.randn(B, S + Qn, h0, generator=g) * torch.rand(B, S + Qn, 1, generator=g)      # rows of very different scale
    gW = hip.xpanel_bwd(ws, x_s.to(dev), x_q.to(dev), Ab.to(dev))
    X = torch.cat([x_s, x_q], 1).double().reshape(-1, D)
    ref = Ab.double().reshape(-1, h0).T @ X
    worst = max(worst, float((gW.cpu().double() - ref).abs().max() / ref.abs().max()))
print(json.dumps({"e": worst}))
"""


def test_xpanel_bwd_split_bf16_has_fp32_accuracy(dev):
    """gW0 = sum_b Abar0_b^T [Xs_b;Xq_b] runs on the bf16 matrix pipe from exact three-way bf16 splits read back with the transposing
    LDS loads (default; FUMI_XPB_SB=0 selects the fp32 MFMA kernel, xpanel.hip).  Its error against fp64 must not exceed the fp32
    MFMA kernel's.  The knob is read once per process, hence the child processes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    errs = {}
    for sb in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", _SBB_SCRIPT % root], env=dict(os.environ, FUMI_XPB_SB=sb), cwd=root,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        errs[sb] = json.loads(r.stdout.strip().splitlines()[-1])["e"]
    assert errs["0"] < 2e-6 and errs["1"] < 2e-6, errs
    assert errs["1"] < 1.5 * errs["0"] + 5e-8, errs


def test_gradient_update_parameters_reference_style_loop_matches_fused_step(dev, ws):
    """a MAML loop written the reference's way (maml.py:156-191: model(x, params=p), F.cross_entropy,
    gradient_update_parameters, outer backward) through fumi_amd.meta on the engine's exported ops gives the fused step's
    logits and second-order meta-gradients."""
    import torch.nn.functional as Fn
    from fumi_amd import hip
    from fumi_amd.meta import gradient_update_parameters
    from fumi_amd.models.maml import PureImageNetwork
    B, N, K, Q, D, hid, T = 3, 5, 2, 4, 64, [32, 16], 2
    ep = cg.make_episodes(17, B, N, K, Q, D, 8)
    p = cg.make_maml_params(17, D, hid, N)
    net = PureImageNetwork(D, N, hid).to(dev)
    with torch.no_grad():
        for dst, src in zip(net.parameters(), p):
            dst.copy_(src)
    outer = 0.0
    logits = []
    for b in range(B):
        xs, ys, xq, yq = (_g(ep[k][b], dev) for k in ("x_s", "y_s", "x_q", "y_q"))
        params = None
        for _ in range(T):
            inner = Fn.cross_entropy(net(xs, params=params), ys)
            params = gradient_update_parameters(net, inner, params=params, step_size=cg.ALPHA, first_order=False)
        lq = net(xq, params=params)
        logits.append(lq.detach())
        outer = outer + Fn.cross_entropy(lq, yq)
    (outer / B).backward()
    out = hip.maml_step(ws, _g(ep["x_s"], dev), _g(ep["y_s"], dev), _g(ep["x_q"], dev), _g(ep["y_q"], dev),
                        [_g(t, dev) for t in p], T, cg.ALPHA, False)
    assert rel_to_max(torch.stack(logits).cpu(), out["logits"].cpu()) <= LOGIT_TOL
    floor = 0.05 * max(float(g.abs().max()) for g in out["g_params"])
    for prm, g in zip(net.parameters(), out["g_params"]):
        assert rel_to_max(prm.grad.cpu(), g.cpu(), floor) <= GRAD_TOL
