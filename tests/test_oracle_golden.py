"""CPU suite: the oracle restatement (oracle/fumi_ref.py) against golden vectors produced by the REAL
reference (oracle/refharness/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import casegen as cg
from oracle import fumi_ref as R
from helpers import load_golden, case_seed, assert_close_max, check_grad

TOL = 2e-5          # fp32 round-off between two orderings of the same math, relative to max|.|


def _leaf(ts):
    return [t.clone().requires_grad_(True) for t in ts]


@pytest.mark.parametrize("name", list(cg.FUMI_CASES))
def test_fumi_oracle_matches_reference(name):
    c, gold = cg.FUMI_CASES[name], load_golden(name)
    seed = case_seed(name)
    assert int(gold["seed"]) == seed
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    d = cg.digest(torch.cat([ep["x_s"].reshape(-1), ep["x_q"].reshape(-1), ep["text_s"].reshape(-1)]))
    np.testing.assert_array_equal(d[3:], gold["in_digest"][3:])        # regenerated inputs == generator's inputs
    np.testing.assert_allclose(d[:3], gold["in_digest"][:3], rtol=1e-12)
    theta, phi = cg.make_fumi_params(seed, c["D"], c["hid"], c["Dt"], c["Ht"])
    if c["init_bias"]:
        W, b = torch.from_numpy(gold["init_head_weight"]), torch.from_numpy(gold["init_head_bias"])
        assert float(W.abs().max()) == 0.0                       # hypernet_init.py:150 (adjust_weights=False)
        assert abs(float(b.norm()) - 2 ** 0.5) < 1e-5            # normc row of norm gain('relu')
        phi[2], phi[3] = W, b
    theta, phi = _leaf(theta), _leaf(phi)
    out = R.fumi_meta_step(theta, phi, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"],
                           c["N"], c["T"], cg.ALPHA, c["tanh"])
    assert_close_max(out["logits"], gold["logits_q"], TOL, "logits")
    assert abs(float(out["loss"]) - float(gold["loss"])) <= TOL * max(1.0, abs(float(gold["loss"])))
    assert np.array_equal(out["preds"].numpy(), gold["preds"])   # integer predictions: bit-exact
    assert abs(float(out["acc"]) - float(gold["acc"])) < 1e-6
    names = [f"im_net.linear{i}.{k}" for i in range(len(c["hid"] or [])) for k in ("weight", "bias")]
    for n, g in zip(names, out["g_theta"]):
        check_grad(gold, "grad." + n, g, 5e-5)
    for n, g in zip(["hyper_net.0.weight", "hyper_net.0.bias", "hyper_net.2.weight", "hyper_net.2.bias"], out["g_phi"]):
        check_grad(gold, "grad." + n, g, 5e-5)


@pytest.mark.parametrize("name", list(cg.MAML_CASES))
def test_maml_oracle_matches_reference(name):
    c, gold = cg.MAML_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], 8)
    p = _leaf(cg.make_maml_params(seed, c["D"], c["hid"], c["N"]))
    out = R.maml_meta_step(p, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["T"], cg.ALPHA, c["first_order"])
    assert_close_max(out["logits"], gold["logits_q"], TOL, "logits")
    assert abs(float(out["loss"]) - float(gold["loss"])) <= TOL * max(1.0, abs(float(gold["loss"])))
    assert np.array_equal(out["preds"].numpy(), gold["preds"])
    names = [f"net.lin_{i}.{k}" for i in range(len(c["hid"] or [])) for k in ("weight", "bias")]
    names += ["net.lin_final.weight", "net.lin_final.bias"]
    for n, g in zip(names, out["g_params"]):
        check_grad(gold, "grad." + n, g, 5e-5)


@pytest.mark.parametrize("name", list(cg.AM3_CASES))
def test_am3_oracle_matches_reference(name):
    c, gold = cg.AM3_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    w = {k: v.clone().requires_grad_(True) for k, v in cg.make_am3_params(seed, c["D"], c["Dt"], c["Ht"], c["P"]).items()}
    out = R.am3_step(w, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], c["lamda_fixed"])
    assert abs(float(out["loss"]) - float(gold["loss"])) <= TOL * max(1.0, abs(float(gold["loss"])))
    assert abs(float(out["acc"]) - float(gold["acc"])) < 1e-6
    assert abs(float(out["avg_lamda"]) - float(gold["avg_lamda"])) < 1e-5
    sd_names = {"Wi": "image_encoder.weight", "bi": "image_encoder.bias", "G0": "g.0.weight", "g0": "g.0.bias",
                "G1": "g.3.weight", "g1": "g.3.bias", "H0": "h.0.weight", "h0": "h.0.bias",
                "H1": "h.3.weight", "h1": "h.3.bias"}
    for k, g in out["grads"].items():
        if float(np.abs(gold["grad." + sd_names[k] + ".digest"][:3]).max()) == 0.0:
            assert float(g.abs().max()) == 0.0
            continue
        check_grad(gold, "grad." + sd_names[k], g, 5e-5)


def test_word_embedding_oracle_matches_reference():
    gold = load_golden("wordemb")
    table = torch.from_numpy(gold["table"]).float()
    tokens = torch.from_numpy(gold["tokens"])
    for mode in ("mean", "max"):
        out = R.word_embedding_pool(tokens, table, int(gold["pad"]), mode)
        assert_close_max(out, gold[mode], 1e-6, mode)


def test_clip_restatement_matches_reference():
    """oracle/fumi_ref.py clip_forward / clip_step against the reference's own CLIP (clip.py:11-41,96-108) outputs."""
    gold = load_golden("clip")
    w = [torch.from_numpy(gold[k]).requires_grad_(True) for k in R.CLIP_KEYS]
    out = R.clip_step(w, torch.from_numpy(gold["text"]), torch.from_numpy(gold["image"]))
    assert_close_max(out["sim"], gold["sim"], 1e-6, "sim")
    assert abs(float(out["loss"]) - float(gold["loss"])) < 1e-6
    for k, g in zip(R.CLIP_KEYS, out["grads"]):
        assert_close_max(g, gold["grad." + k], 1e-5, k)
    zs = R.clip_forward([t.detach() for t in w], torch.from_numpy(gold["text"][:1]), torch.from_numpy(gold["image"][:5]))
    assert_close_max(zs, gold["zero_shot"], 1e-6, "zero-shot call")


@pytest.mark.parametrize("enc", ["RNN", "RNNhid"])
def test_fumi_with_trainable_lstm_oracle_matches_reference(enc):
    """--fine_tune with RNN / RNNhid (fumi.py:46-67): the loss reaches the bi-LSTM through get_hyper_params (fumi.py:196-212).
    The oracle's lstm_encode inside fumi_meta_step's graph against the reference's own .grad of rnn.* (and of every other
    parameter) after one unmodified evaluate(train)."""
    from helpers import rnn_finetune_case, RNN_KEYS
    gold, c, ep, theta, phi, table, lstm_w = rnn_finetune_case()
    theta, phi, lstm_w = _leaf(theta), _leaf(phi), _leaf(lstm_w)
    text = R.lstm_encode(ep["text_s"], table, lstm_w, 0, enc == "RNNhid")
    out = R.fumi_meta_step(theta, phi, text, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], c["T"], cg.ALPHA, True,
                           extra=lstm_w)
    assert abs(float(out["loss"]) - float(gold[f"{enc}.loss"])) <= TOL and abs(float(out["acc"]) - float(gold[f"{enc}.acc"])) < 1e-6
    assert np.array_equal(out["preds"].numpy(), gold[f"{enc}.preds"])
    for k, g in zip(RNN_KEYS, out["g_extra"]):
        assert_close_max(g, gold[f"{enc}.grad.text_encoder.rnn.{k}"], 5e-5, k)
        assert float(np.abs(gold[f"{enc}.grad.text_encoder.rnn.{k}"]).max()) > 1e-6          # the fixture is not a zero gradient
    names = ["im_net.linear0.weight", "im_net.linear0.bias", "hyper_net.0.weight", "hyper_net.0.bias", "hyper_net.2.weight",
             "hyper_net.2.bias"]
    for n, g in zip(names, out["g_theta"] + out["g_phi"]):
        assert_close_max(g, gold[f"{enc}.grad.{n}"], 5e-5, n)


@pytest.mark.parametrize("enc", ["RNN", "RNNhid"])
def test_am3_with_trainable_lstm_oracle_matches_reference(enc):
    """AM3(text_encoder=RNN / RNNhid, fine_tune=True) (am3.py:61-76,113-126): every support row's encoding feeds its class prototype.
    The oracle's lstm_encode inside am3_step's graph against the reference's own .grad of rnn.* and of the ten AM3 tensors."""
    from helpers import rnn_finetune_case, RNN_KEYS
    gold, c, ep, _, _, table, lstm_w = rnn_finetune_case()
    w = {k: v.clone().requires_grad_(True) for k, v in cg.make_am3_params(int(gold["seed"]), c["D"], c["Dt"], c["Ht"], int(gold["am3_P"])).items()}
    lstm_w = _leaf(lstm_w)
    text = R.lstm_encode(ep["text_s"], table, lstm_w, 0, enc == "RNNhid")
    out = R.am3_step(w, text, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], extra=lstm_w)
    assert abs(float(out["loss"]) - float(gold[f"am3.{enc}.loss"])) <= TOL * 3 and abs(float(out["acc"]) - float(gold[f"am3.{enc}.acc"])) < 1e-6
    assert abs(float(out["avg_lamda"]) - float(gold[f"am3.{enc}.avg_lamda"])) < 1e-6
    for k, g in zip(RNN_KEYS, out["g_extra"]):
        assert_close_max(g, gold[f"am3.{enc}.grad.text_encoder.rnn.{k}"], 5e-5, k)
        assert float(np.abs(gold[f"am3.{enc}.grad.text_encoder.rnn.{k}"]).max()) > 1e-6
    for n, g in zip(cg.am3_state_dict(w), out["grads"].values()):
        assert_close_max(g, gold[f"am3.{enc}.grad.{n}"], 5e-5, n)


def test_lstm_encoder_restatement_matches_reference():
    """oracle/fumi_ref.py lstm_encode against the reference's RNN (output states) and RnnHid (cell states), common.py:44-161."""
    gold = load_golden("rnn")
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [torch.from_numpy(gold["rnn." + n + suffix]) for suffix in ("", "_reverse") for n in names]
    tokens, table = torch.from_numpy(gold["tokens"]), torch.from_numpy(gold["embed.weight"])
    assert_close_max(R.lstm_encode(tokens, table, w, 0, False), gold["rnn"], 1e-6, "RNN")
    assert_close_max(R.lstm_encode(tokens, table, w, 0, True), gold["rnnhid"], 1e-6, "RnnHid")
