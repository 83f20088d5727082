"""CPU suite: the host-side mirror of the reference surface (flags, state_dict keys, evaluate plumbing, loops,
optimizer step, checkpoints, CLI) with the oracle standing in for the GPU engine (tests/oracle_engine.py)."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import casegen as cg
from helpers import GOLDEN, load_golden, case_seed
from oracle_engine import OracleEngine


@pytest.fixture()
def oracle_engine():
    from fumi_amd import engine
    old = engine.set_engine(OracleEngine())
    yield
    engine.set_engine(old)


def _surface():
    return json.load(open(os.path.join(GOLDEN, "surface.json")))


def test_every_reference_flag_same_name_default_type():
    from fumi_amd.utils import utils
    ref = _surface()["flags"]
    p = utils.parser()
    mine = {a.dest: a for a in p._actions if a.option_strings and a.dest != "help"}
    assert len(ref) == 50
    for dest, r in ref.items():
        assert dest in mine, f"flag {r['flag']} missing"
        a = mine[dest]
        assert a.option_strings[0] == r["flag"]
        assert a.default == r["default"], dest
        assert (a.type.__name__ if a.type else None) == r["type"], dest
        assert a.nargs == r["nargs"], dest
        assert (list(a.choices) if a.choices else None) == r["choices"], dest
        assert (type(a).__name__ == "_StoreTrueAction") == r["store_true"], dest


def test_state_dict_keys_match_reference(oracle_engine):
    from fumi_amd.utils import utils
    from fumi_amd.models import fumi, maml, am3
    ref = _surface()
    d = utils.parser().parse_args([])
    f = fumi.FUMI(n_way=d.num_ways, im_emb_dim=d.im_emb_dim, im_hid_dim=d.im_hid_dim, text_encoder="BERT",
                  text_emb_dim=d.text_emb_dim, text_hid_dim=d.text_hid_dim, dropout_rate=d.dropout,
                  norm_hypernet=d.norm_hypernet)
    m = maml.PureImageNetwork(im_embed_dim=d.im_emb_dim, n_way=d.num_ways, hidden_dims=d.im_hid_dim)
    a = am3.AM3(im_encoder=d.im_encoder, im_emb_dim=d.im_emb_dim, text_encoder="BERT", text_emb_dim=d.text_emb_dim,
                text_hid_dim=d.text_hid_dim, prototype_dim=d.prototype_dim, dropout=d.dropout)
    for mod, key in ((f, "fumi"), (m, "maml"), (a, "am3")):
        assert {k: list(v.shape) for k, v in mod.state_dict().items()} == ref[key], key
    assert [n for n, _ in f.im_net.meta_named_parameters()] == ["linear0.weight", "linear0.bias", "linear1.weight", "linear1.bias"]
    assert [n for n, _ in m.meta_named_parameters()][:2] == ["net.lin_0.weight", "net.lin_0.bias"]


def test_error_types_match_reference():
    from fumi_amd.models import fumi, am3
    with pytest.raises(NameError):
        fumi.FUMI(text_encoder="nope")
    with pytest.raises(NotImplementedError):
        fumi.FUMI(init_all_layers=True)
    with pytest.raises(NameError):
        am3.AM3("nope", 8, "BERT")
    with pytest.raises(NameError):
        am3.AM3("precomputed", 8, "nope")


def _args(T, first_order=False):
    return SimpleNamespace(device=torch.device("cpu"), num_train_adapt_steps=T, num_test_adapt_steps=T,
                           step_size=cg.ALPHA, first_order=first_order, num_ways=5, batch_size=4)


@pytest.mark.parametrize("name", ["fumi_t1", "fumi_t5_tanh"])
def test_fumi_evaluate_train_step_equals_reference_post_step_params(name, oracle_engine):
    """evaluate(train) = meta-grads + Adam(lr 3e-5, wd 5e-4) step: parameters afterwards match the reference's."""
    from fumi_amd.models.fumi import FUMI
    c, gold = cg.FUMI_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    theta, phi = cg.make_fumi_params(seed, c["D"], c["hid"], c["Dt"], c["Ht"])
    model = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="BERT", text_emb_dim=c["Dt"],
                 text_hid_dim=c["Ht"], dropout_rate=0.0, norm_hypernet=c["tanh"])
    model.load_state_dict(cg.fumi_state_dict(theta, phi))
    opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)
    loss, acc, preds, tgt = model.evaluate(_args(c["T"]), cg.to_batch(ep), opt, "train")
    assert abs(float(loss) - float(gold["loss"])) < 2e-5 and abs(float(acc) - float(gold["acc"])) < 1e-6
    assert preds.dtype == torch.float32 and np.array_equal(preds.numpy().astype(np.int64), gold["preds"])   # fumi.py:140
    for n, p in model.named_parameters():
        d = cg.digest(p)
        np.testing.assert_allclose(d[3:], gold[f"post.{n}.digest"][3:], rtol=0, atol=2e-7)
    l2, a2, p2, _ = model.evaluate(_args(c["T"]), cg.to_batch(ep), None, "test")
    assert abs(float(l2) - float(gold["test_loss"])) < 2e-5 and np.array_equal(p2.numpy().astype(np.int64), gold["test_preds"])
    assert not model.training


def test_maml_evaluate_matches_reference(oracle_engine):
    from fumi_amd.models import maml
    name = "maml_2nd"
    c, gold = cg.MAML_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], 8)
    model = maml.PureImageNetwork(im_embed_dim=c["D"], n_way=c["N"], hidden_dims=c["hid"])
    model.load_state_dict(cg.maml_state_dict(cg.make_maml_params(seed, c["D"], c["hid"], c["N"])))
    opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)
    loss, acc = maml.evaluate(_args(c["T"]), model, cg.to_batch(ep), opt, "train")
    assert abs(float(loss) - float(gold["loss"])) < 2e-5 and abs(float(acc) - float(gold["acc"])) < 1e-6
    assert model.training                                                    # maml.py:143 forces train mode


@pytest.mark.parametrize("name", ["am3_lam", "am3_lam0"])
def test_am3_evaluate_matches_reference(name, oracle_engine):
    from fumi_amd.models.am3 import AM3
    c, gold = cg.AM3_CASES[name], load_golden(name)
    seed = case_seed(name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    model = AM3("precomputed", c["D"], "BERT", text_emb_dim=c["Dt"], text_hid_dim=c["Ht"], prototype_dim=c["P"],
                dropout=0.0, lamda_fixed=c["lamda_fixed"])
    model.load_state_dict(cg.am3_state_dict(cg.make_am3_params(seed, c["D"], c["Dt"], c["Ht"], c["P"])))
    opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)
    loss, acc, f1, prec, rec, lam = model.evaluate(cg.to_batch(ep), opt, None, c["N"], torch.device("cpu"), "train")
    for got, key in ((loss, "loss"), (acc, "acc"), (f1, "f1"), (prec, "prec"), (rec, "rec"), (lam, "avg_lamda")):
        assert abs(float(got) - float(gold[key])) < 2e-5, key
    for n, p in model.named_parameters():
        np.testing.assert_allclose(cg.digest(p)[3:], gold[f"post.{n}.digest"][3:], rtol=0, atol=2e-7)
    with torch.no_grad():
        r = model.evaluate(cg.to_batch(ep), None, None, c["N"], torch.device("cpu"), "test")
    assert len(r) == 11 and abs(float(r[0]) - float(gold["test_loss"])) < 2e-5
    assert np.array_equal(np.asarray(r[6]), gold["test_preds"])
    np.testing.assert_allclose(np.asarray(r[10]), gold["test_lamda_s"], atol=1e-6)


def test_am3_rand_text_encoder_bypasses_g(oracle_engine):
    """text_encoder='rand' (am3.py:68-69,118-121): prototype-space text rows are drawn uniformly in [-1, 1), g is not part of
    the graph (its .grad stays None), h and the image encoder train.  The step receives an exact identity in g's place:
    with lamda_fixed = 0 the prototypes are the drawn rows themselves, so the loss equals the oracle's on those rows."""
    from fumi_amd.models.am3 import AM3
    from oracle import fumi_ref as R
    c = cg.AM3_CASES["am3_lam"]
    P, Ht = 8, 20
    ep = cg.make_episodes(4, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    model = AM3("precomputed", c["D"], "rand", text_emb_dim=c["Dt"], text_hid_dim=Ht, prototype_dim=P, dropout=0.0)
    g_before = [p.detach().clone() for p in model.g.parameters()]
    h_before = [p.detach().clone() for p in model.h.parameters()]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    torch.manual_seed(5)
    loss = model.evaluate(cg.to_batch(ep), opt, None, c["N"], torch.device("cpu"), "train")[0]
    torch.manual_seed(5)
    drawn = 2 * torch.rand(c["B"], c["N"] * c["K"], P) - 1
    w = {k: v.detach().clone() for k, v in zip(
        ["Wi", "bi", "G0", "g0", "G1", "g1", "H0", "h0", "H1", "h1"],
        [model.image_encoder.weight, model.image_encoder.bias] + model._rand_g(torch.device("cpu")) + h_before)}
    w["Wi"], w["bi"] = w["Wi"] + 0, w["bi"] + 0
    assert all(p.grad is None for p in model.g.parameters())
    assert all(torch.equal(a, b.detach()) for a, b in zip(g_before, model.g.parameters()))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(h_before, model.h.parameters()))
    # the identity in g's form is exact
    G0, g0, G1, g1 = model._rand_g(torch.device("cpu"))
    assert torch.equal(torch.relu(drawn @ G0.t() + g0) @ G1.t() + g1, drawn)
    assert np.isfinite(float(loss))
    with pytest.raises(NotImplementedError):
        AM3("precomputed", c["D"], "rand", text_emb_dim=c["Dt"], text_hid_dim=Ht, prototype_dim=P, dropout=0.5).evaluate(
            cg.to_batch(ep), opt, None, c["N"], torch.device("cpu"), "train")


def test_gradient_update_parameters_matches_torchmeta_contract(oracle_engine):
    """torchmeta's gradient_update_parameters (fumi.py:172-176, maml.py:173-177): p - step_size * dL/dp for
    every meta-parameter, graph kept unless first_order; MetaLinear.forward is differentiable twice."""
    import torch.nn.functional as Fn
    from fumi_amd.meta import gradient_update_parameters
    from fumi_amd.models.maml import PureImageNetwork
    torch.manual_seed(0)
    net = PureImageNetwork(12, 3, [8])
    x, y = torch.randn(6, 12), torch.tensor([0, 1, 2, 0, 1, 2])
    loss = Fn.cross_entropy(net(x), y)
    ref_g = torch.autograd.grad(loss, list(net.parameters()), create_graph=True)
    upd = gradient_update_parameters(net, loss, step_size=0.3)
    assert list(upd.keys()) == [n for n, _ in net.meta_named_parameters()]
    for (n, p), g in zip(net.meta_named_parameters(), ref_g):
        assert torch.allclose(upd[n], p - 0.3 * g, atol=1e-6)
    # second order: d/dp of the post-update loss flows through the update
    outer = Fn.cross_entropy(net(x, params=upd), y)
    g2 = torch.autograd.grad(outer, list(net.parameters()))
    # plain torch restatement
    ps = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
    f = lambda q, inp: Fn.linear(torch.relu(Fn.linear(inp, q[0], q[1])), q[2], q[3])
    l1 = Fn.cross_entropy(f(ps, x), y)
    gs = torch.autograd.grad(l1, ps, create_graph=True)
    l2 = Fn.cross_entropy(f([p - 0.3 * g for p, g in zip(ps, gs)], x), y)
    for a, b in zip(g2, torch.autograd.grad(l2, ps)):
        assert torch.allclose(a, b, atol=1e-6)
    fo = gradient_update_parameters(net, Fn.cross_entropy(net(x), y), step_size=0.3, first_order=True)
    assert all(not v.requires_grad or v.grad_fn is not None for v in fo.values())
    with pytest.raises(ValueError):
        gradient_update_parameters(torch.nn.Linear(2, 2), loss)


def test_test_loop_consumes_max_plus_one_batches(oracle_engine):
    """The reference's break is tested after the batch is processed (fumi.py:324)."""
    from fumi_amd.models import fumi
    c = cg.FUMI_CASES["fumi_t1"]
    model = fumi.FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="BERT", text_emb_dim=c["Dt"],
                      text_hid_dim=c["Ht"], norm_hypernet=False)
    seen = []

    def loader():
        for i in range(10):
            seen.append(i)
            yield cg.to_batch(cg.make_episodes(i, 2, c["N"], c["K"], c["Q"], c["D"], c["Dt"]))
    fumi.test_loop(_args(1), model, loader(), 3)
    assert len(seen) == 4


def test_hypernet_bias_init():
    from fumi_amd.models.fumi import FUMI
    m = FUMI(im_hid_dim=[32, 16], text_emb_dim=12, text_hid_dim=10, init_bias=True)
    assert float(m.hyper_net[2].weight.abs().max()) == 0.0
    assert abs(float(m.hyper_net[2].bias.norm()) - 2 ** 0.5) < 1e-5


def test_macro_metrics_equal_sklearn():
    from sklearn.metrics import precision_recall_fscore_support, accuracy_score
    from fumi_amd.utils.utils import macro_metrics
    rs = np.random.RandomState(0)
    for _ in range(5):
        t, p = rs.randint(0, 5, 200), rs.randint(0, 6, 200)
        acc, f1, prec, rec = macro_metrics(t, p)
        pr, rc, f, _ = precision_recall_fscore_support(t, p, average="macro", zero_division=0)
        assert abs(acc - accuracy_score(t, p)) < 1e-12 and abs(f1 - f) < 1e-12 and abs(prec - pr) < 1e-12 and abs(rec - rc) < 1e-12


def test_fumi_dropout_train_vs_eval(oracle_engine):
    """--dropout > 0 (CLI default 0.25): active in train mode with a fresh seed per step, off in eval mode (fumi.py:93-99,123-127)."""
    from fumi_amd.models.fumi import FUMI
    c = cg.FUMI_CASES["fumi_t1"]
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_emb_dim=c["Dt"], text_hid_dim=c["Ht"], dropout_rate=0.25,
             text_encoder="BERT")
    assert "im_net.linear0.weight" in m.state_dict() and hasattr(m.im_net, "dropout0")
    ep = cg.make_episodes(1, 2, c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    torch.manual_seed(0)
    l1 = float(m.evaluate(_args(1), cg.to_batch(ep), opt, "train")[0])
    l2 = float(m.evaluate(_args(1), cg.to_batch(ep), opt, "train")[0])
    torch.manual_seed(0)
    l3 = float(m.evaluate(_args(1), cg.to_batch(ep), opt, "train")[0])
    assert l1 != l2 and l1 == l3                               # new mask every step, reproducible from torch.manual_seed
    e1 = float(m.evaluate(_args(1), cg.to_batch(ep), None, "test")[0])
    e2 = float(m.evaluate(_args(1), cg.to_batch(ep), None, "test")[0])
    assert e1 == e2


def test_cli_maml_synthetic_cpu_plumbing(oracle_engine, tmp_path, monkeypatch):
    """BASELINE.json configs[0]: `main.py --model maml` on CPU (plumbing): parse -> loaders -> train -> checkpoint -> test."""
    from fumi_amd import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ["--model", "maml", "--dataset", "synthetic", "--disable_cuda", "--num_shots", "1", "--batch_size", "4",
            "--im_emb_dim", "512", "--image_embedding_model", "resnet-34", "--im_hid_dim", "32", "16",
            "--epochs", "2", "--eval_freq", "2", "--num_ep_test", "8", "--num_train_adapt_steps", "2",
            "--num_test_adapt_steps", "2", "--log_dir", str(tmp_path / "res"), "--synthetic_classes", "12"]
    args = cli.parse_args(argv)
    assert args.device.type == "cpu"
    res = cli.main(args)
    assert np.isfinite(res["test_loss"]) and 0.0 <= res["test_acc"] <= 1.0
    runs = os.listdir(tmp_path / "res" / "runs")
    assert runs and os.path.exists(tmp_path / "res" / "runs" / runs[0] / "ckpt.pth.tar")


def test_disable_cuda_is_refused_where_the_engine_is_first_needed(tmp_path):
    """The reference falls back to the CPU (fumi/main.py:145-146); this engine has no CPU path and says so at the top of `main`,
    before anything is built -- `parse_args` itself stays pure (host-only users of the parser get the reference's namespace)."""
    from fumi_amd import engine, hip, main as cli
    old = engine.set_engine(None)
    try:
        args = cli.parse_args(["--model", "maml", "--disable_cuda", "--log_dir", str(tmp_path)])
        assert args.device.type == "cpu"
        with pytest.raises(hip.FumiHipError, match="no CPU execution path"):
            cli.main(args)
        assert not os.listdir(tmp_path)                                  # refused before any side effect
    finally:
        engine.set_engine(old)


def test_unsupported_flag_combinations_are_refused_up_front(oracle_engine):
    """`--model am3 --text_encoder rand` trains only with --dropout 0 (am3.py:118-126 applies dropout inside h only): refused by
    `check_supported` for every model name that builds AM3 (unknown names are AM3, like utils.init_model).  --fine_tune with RNN /
    RNNhid trains the bi-LSTM like the reference (fumi.py:65-67) for every model and image encoder: accepted."""
    from fumi_amd import main as cli
    for enc in ("RNN", "RNNhid"):
        for model in ("fumi", "am3", "some-unknown-name", "maml"):
            cli.check_supported(cli.parse_args(["--model", model, "--disable_cuda", "--text_encoder", enc, "--fine_tune"]))
        for im in ("conv4", "resnet12"):
            cli.check_supported(cli.parse_args(["--model", "fumi", "--disable_cuda", "--text_encoder", enc, "--fine_tune",
                                                "--im_encoder", im]))
    with pytest.raises(NotImplementedError, match="--dropout 0"):
        cli.check_supported(cli.parse_args(["--model", "am3", "--disable_cuda", "--text_encoder", "rand"]))      # CLI default 0.25
    cli.check_supported(cli.parse_args(["--model", "am3", "--disable_cuda", "--text_encoder", "rand", "--dropout", "0"]))
    cli.check_supported(cli.parse_args(["--model", "am3", "--disable_cuda", "--text_encoder", "rand", "--evaluate"]))


def test_cli_flag_validation_raises_value_error(oracle_engine, tmp_path, monkeypatch):
    from fumi_amd import main as cli
    monkeypatch.chdir(tmp_path)
    args = cli.parse_args(["--disable_cuda", "--image_embedding_model", "resnet-34", "--log_dir", str(tmp_path)])
    with pytest.raises(ValueError):
        cli.main(args)


def test_product_engine_refuses_cpu():
    """Without the test hook the product engine is the HIP library and it refuses CPU tensors loudly."""
    from fumi_amd import engine, hip
    from fumi_amd.models.fumi import FUMI
    engine.set_engine(None)
    c = cg.FUMI_CASES["fumi_t1"]
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_emb_dim=c["Dt"], text_hid_dim=c["Ht"], norm_hypernet=False)
    ep = cg.make_episodes(1, 2, c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    with pytest.raises(hip.FumiHipError):
        m.evaluate(_args(1), cg.to_batch(ep), None, "test")


def test_fumi_glove_path_uses_class_rows_only(oracle_engine):
    """text_encoder='glove': evaluate pools only the N class rows (select + bag) and matches the all-rows reference order."""
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.models import common
    from oracle import fumi_ref as R
    V, L, E = 50, 7, 12
    rs = np.random.RandomState(0)
    words = [f"w{i}" for i in range(V)]
    common.register_word_vectors("glove", common.ArrayKeyedVectors(words, rs.standard_normal((V, E)).astype(np.float32)))
    dictionary = {"PAD": 0, **{w: i for i, w in enumerate(words) if i > 0}}
    c = dict(B=3, N=4, K=2, Q=3, D=32, hid=[16, 8], Ht=10)
    ep = cg.make_episodes(4, c["B"], c["N"], c["K"], c["Q"], c["D"], 1, tokens=(V, L, 0))
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="glove", text_hid_dim=c["Ht"],
             dictionary=dictionary, norm_hypernet=False)
    assert m.text_emb_dim == E and "text_encoder.embed.weight" in m.state_dict()
    args = SimpleNamespace(device=torch.device("cpu"), num_train_adapt_steps=2, num_test_adapt_steps=2, step_size=cg.ALPHA,
                           first_order=False, num_ways=c["N"], batch_size=c["B"])
    loss, acc, preds, _ = m.evaluate(args, cg.to_batch(ep), None, "test")
    text = R.word_embedding_pool(ep["text_s"], m.text_encoder.embed.weight.detach(), 0, "mean")
    th = [p.detach().clone().requires_grad_(True) for p in m._theta()]
    ph = [p.detach().clone().requires_grad_(True) for p in m._phi()]
    ref = R.fumi_meta_step(th, ph, text, ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], c["N"], 2, cg.ALPHA, False, need_grad=False)
    assert abs(float(loss) - float(ref["loss"])) < 1e-5 and np.array_equal(preds.numpy().astype(np.int64), ref["preds"].numpy())


@pytest.mark.parametrize("enc", ["RNN", "RNNhid"])
def test_fumi_trains_the_bilstm_under_fine_tune(enc, oracle_engine):
    """FUMI(text_encoder=RNN / RNNhid, fine_tune=True).evaluate(train) (fumi.py:46-67,115-196): class token rows -> taped LSTM
    forward -> meta-step with the text adjoint -> LSTM backward -> Adam over every trainable tensor.  Gradients and post-step
    parameters against the reference's own (host plumbing on the checker engine; the kernels: test_hip_parity.py)."""
    from helpers import rnn_finetune_case, RNN_KEYS
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.models import common
    gold, c, ep, theta, phi, table, lstm_w = rnn_finetune_case()
    words = [f"w{i}" for i in range(30)]
    common.register_word_vectors("glove", common.ArrayKeyedVectors(words, table[1:].numpy()))
    dictionary = {"PAD": 0, **{w: i + 1 for i, w in enumerate(words)}}
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder=enc, text_emb_dim=c["Dt"], text_hid_dim=c["Ht"],
             dropout_rate=0.0, dictionary=dictionary, fine_tune=True)
    sd = cg.fumi_state_dict(theta, phi)
    sd.update({k: torch.from_numpy(gold[k]) for k in gold if k.startswith("text_encoder.")})
    m.load_state_dict(sd)
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    assert [n for n in trainable if n.startswith("text_encoder.")] == ["text_encoder.rnn." + k for k in RNN_KEYS]
    opt = torch.optim.Adam(m.parameters(), lr=3e-5, weight_decay=5e-4)
    loss, acc, preds, _ = m.evaluate(_args(c["T"]), cg.to_batch(ep), opt, "train")
    assert abs(float(loss) - float(gold[f"{enc}.loss"])) < 2e-5 and abs(float(acc) - float(gold[f"{enc}.acc"])) < 1e-6
    assert np.array_equal(preds.numpy().astype(np.int64), gold[f"{enc}.preds"])
    for n, p in m.named_parameters():
        if p.requires_grad:
            g = gold[f"{enc}.grad.{n}"]
            assert float((p.grad - torch.from_numpy(g)).abs().max()) <= 5e-5 * max(float(np.abs(g).max()), 1e-6), n
            np.testing.assert_allclose(cg.digest(p)[3:], gold[f"{enc}.post.{n}.digest"][3:], rtol=0, atol=2e-7)
    # a frozen encoder (no --fine_tune) keeps the LSTM out of the step
    m2 = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder=enc, text_emb_dim=c["Dt"], text_hid_dim=c["Ht"],
              dropout_rate=0.0, dictionary=dictionary)
    m2.load_state_dict(sd)
    l2, _, _, _ = m2.evaluate(_args(c["T"]), cg.to_batch(ep), torch.optim.Adam([p for p in m2.parameters() if p.requires_grad]), "train")
    assert abs(float(l2) - float(gold[f"{enc}.loss"])) < 2e-5 and all(p.grad is None for p in m2.text_encoder.parameters())


@pytest.mark.parametrize("enc", ["RNN", "RNNhid"])
def test_am3_trains_the_bilstm_under_fine_tune(enc, oracle_engine):
    """AM3(text_encoder=RNN / RNNhid, fine_tune=True).evaluate(train) (am3.py:61-76,128-212): all B*S token rows -> taped LSTM forward
    -> step with the text adjoint -> LSTM backward -> Adam.  Gradients and post-step parameters against the reference's own."""
    from helpers import rnn_finetune_case, RNN_KEYS
    from fumi_amd.models.am3 import AM3
    from fumi_amd.models import common
    gold, c, ep, _, _, table, lstm_w = rnn_finetune_case()
    words = [f"w{i}" for i in range(30)]
    common.register_word_vectors("glove", common.ArrayKeyedVectors(words, table[1:].numpy()))
    dictionary = {"PAD": 0, **{w: i + 1 for i, w in enumerate(words)}}
    P = int(gold["am3_P"])
    m = AM3("precomputed", c["D"], enc, text_emb_dim=c["Dt"], text_hid_dim=c["Ht"], prototype_dim=P, dropout=0.0, fine_tune=True,
            dictionary=dictionary)
    sd = cg.am3_state_dict(cg.make_am3_params(int(gold["seed"]), c["D"], c["Dt"], c["Ht"], P))
    sd.update({k: torch.from_numpy(gold[k]) for k in gold if k.startswith("text_encoder.")})
    m.load_state_dict(sd)
    opt = torch.optim.Adam(m.parameters(), lr=3e-5, weight_decay=5e-4)
    r = m.evaluate(cg.to_batch(ep), opt, None, c["N"], torch.device("cpu"), "train")
    assert abs(float(r[0]) - float(gold[f"am3.{enc}.loss"])) < 6e-5 and abs(float(r[1]) - float(gold[f"am3.{enc}.acc"])) < 1e-6
    n_checked = 0
    for n, p in m.named_parameters():
        if p.requires_grad:
            g = gold[f"am3.{enc}.grad.{n}"]
            assert float((p.grad - torch.from_numpy(g)).abs().max()) <= 5e-5 * max(float(np.abs(g).max()), 1e-6), n
            np.testing.assert_allclose(cg.digest(p)[3:], gold[f"am3.{enc}.post.{n}.digest"][3:], rtol=0, atol=2e-7)
            n_checked += 1
    assert n_checked == 18


def test_bench_self_launches_one_process_per_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` (how the driver calls it) starts torch.distributed.run as a CHILD process with N ranks and
    returns its exit code; the parent never touches the GPU.  (The child is faked here: no GPU in the CPU suite.)"""
    import subprocess
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 4)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("the parent must not touch torch.cuda")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)               # not enough devices: refuse, exit code 2
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2


def test_cli_maml_conv4_plumbing_on_cpu(oracle_engine, tmp_path, monkeypatch):
    """BASELINE.json configs[0] as worded: `main.py --model maml` 5-way 1-shot with the Conv4 encoder, meta-batch 4, on CPU
    (plumbing: flags -> image loader -> Conv4 module -> inner loop -> checkpoint -> test; tiny images keep it to seconds)."""
    from fumi_amd import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ["--model", "maml", "--dataset", "synthetic", "--disable_cuda", "--im_encoder", "conv4", "--image_size", "16",
            "--num_shots", "1", "--num_shots_test", "2", "--batch_size", "4", "--epochs", "1", "--eval_freq", "1",
            "--num_ep_test", "4", "--num_train_adapt_steps", "1", "--num_test_adapt_steps", "1",
            "--log_dir", str(tmp_path / "res"), "--synthetic_classes", "10"]
    args = cli.parse_args(argv)
    res = cli.main(args)
    assert np.isfinite(res["test_loss"]) and 0.0 <= res["test_acc"] <= 1.0
    runs = os.listdir(tmp_path / "res" / "runs")
    ck = torch.load(tmp_path / "res" / "runs" / runs[0] / "ckpt.pth.tar", weights_only=False)
    keys = set(ck["state_dict"])
    assert {"net.features.block0.conv.weight", "net.features.block3.norm.bias", "net.lin_final.weight"} <= keys
    assert ck["state_dict"]["net.features.block0.conv.weight"].shape == (64, 3, 3, 3)
    assert ck["state_dict"]["net.lin_final.weight"].shape == (5, 64)            # 16 -> 8 -> 4 -> 2 -> 1: 64 features


def test_fumi_conv4_module_trains_through_evaluate(oracle_engine):
    """FUMI(im_encoder='conv4'): the hypernetwork emits [N, F+1] rows for the Conv4 features, evaluate() fills .grad of the 12
    encoder tensors + 4 hypernetwork tensors and steps the optimizer; Conv4.forward / im_forward run on the feature op."""
    from fumi_amd.models.fumi import FUMI
    from oracle import conv4_ref as C
    torch.manual_seed(0)
    m = FUMI(n_way=3, im_encoder="conv4", image_size=16, image_channels=3, text_emb_dim=12, text_hid_dim=8, norm_hypernet=False)
    assert m.im_net.feature_dim == 64 and m.hyper_net[2].weight.shape == (65, 8)
    assert [k for k in m.state_dict() if k.startswith("im_net.")][:3] == ["im_net.block0.conv.weight", "im_net.block0.norm.weight",
                                                                           "im_net.block0.norm.bias"]
    ep = C.make_image_episodes(3, 2, 3, 2, 2, 3, 16, 16, 12)
    args = SimpleNamespace(device=torch.device("cpu"), num_train_adapt_steps=1, num_test_adapt_steps=1, step_size=0.05,
                           first_order=False, num_ways=3, batch_size=2)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    before = [p.detach().clone() for p in m.parameters()]
    loss, acc, _, _ = m.evaluate(args, cg.to_batch(ep), opt, "train")
    assert np.isfinite(float(loss)) and 0.0 <= float(acc) <= 1.0
    changed = [not torch.equal(a, b.detach()) for a, b in zip(before, m.parameters())]
    assert all(changed), "every meta-parameter receives a gradient (Conv4 weights, BN weight / bias, hypernetwork)"
    loss2, _, preds, tgt = m.evaluate(args, cg.to_batch(ep), None, "test")
    assert preds.shape == tgt.shape == (2, 6)
    feats = m.im_net(ep["x_q"])                                         # [B, Qn, F]: batch statistics per query set
    assert feats.shape == (2, 6, 64)
    assert torch.allclose(feats[1], C.conv4_features(ep["x_q"][1], [t.detach() for t in m.im_net.theta()]), atol=1e-6)


def test_fumi_resnet12_module_trains_through_evaluate(oracle_engine):
    """FUMI(im_encoder='resnet12') (BASELINE.json configs[4]'s backbone at the im_net seam): state_dict keys, the [N, 641] head, every
    one of the 48 encoder tensors + 4 hypernetwork tensors trained through evaluate(), the feature helper."""
    from fumi_amd.models.fumi import FUMI
    from oracle import conv4_ref as C, resnet12_ref as RR
    torch.manual_seed(0)
    m = FUMI(n_way=3, im_encoder="resnet12", image_size=16, image_channels=3, text_emb_dim=12, text_hid_dim=8, norm_hypernet=False)
    assert m.im_net.feature_dim == 640 and m.hyper_net[2].weight.shape == (641, 8) and len(m.im_net.theta()) == 48
    keys = [k for k in m.state_dict() if k.startswith("im_net.block1.")]
    assert keys[:3] == ["im_net.block1.conv1.weight", "im_net.block1.bn1.weight", "im_net.block1.bn1.bias"]
    assert m.state_dict()["im_net.block1.shortcut.weight"].shape == (160, 64, 1, 1)
    ep = C.make_image_episodes(3, 2, 3, 2, 2, 3, 16, 16, 12)
    args = SimpleNamespace(device=torch.device("cpu"), num_train_adapt_steps=1, num_test_adapt_steps=1, step_size=0.05,
                           first_order=False, num_ways=3, batch_size=2)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    before = [p.detach().clone() for p in m.parameters()]
    loss, acc, _, _ = m.evaluate(args, cg.to_batch(ep), opt, "train")
    assert np.isfinite(float(loss)) and 0.0 <= float(acc) <= 1.0
    assert all(not torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))
    feats = m.im_net(ep["x_q"])
    assert feats.shape == (2, 6, 640)
    assert torch.allclose(feats[1], RR.features(ep["x_q"][1], [t.detach() for t in m.im_net.theta()]), atol=1e-5)


def test_am3_conv4_module_trains_through_evaluate(oracle_engine):
    """AM3(im_encoder='conv4'): Conv4 features feed the reference's Linear into the prototype space; evaluate() fills .grad of the
    10 AM3 tensors and the 12 backbone tensors (through the step's image-row adjoints) and steps the optimizer."""
    from fumi_amd.models.am3 import AM3
    from oracle import conv4_ref as C
    torch.manual_seed(0)
    m = AM3(im_encoder="conv4", im_emb_dim=0, text_encoder="BERT", text_emb_dim=12, text_hid_dim=8, prototype_dim=6, dropout=0.0,
            image_size=16, image_channels=3)
    assert m.conv.feature_dim == 64 and m.image_encoder.weight.shape == (6, 64)
    assert {"conv.block0.conv.weight", "conv.block3.norm.bias", "image_encoder.weight", "g.0.weight", "h.3.bias"} <= set(m.state_dict())
    ep = C.make_image_episodes(4, 2, 3, 2, 2, 3, 16, 16, 12)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    before = [p.detach().clone() for p in m.parameters()]
    out = m.evaluate(cg.to_batch(ep), opt, None, 3, torch.device("cpu"), "train")
    assert len(out) == 6 and np.isfinite(float(out[0]))
    changed = [not torch.equal(a, b.detach()) for a, b in zip(before, m.parameters())]
    assert all(changed), "every parameter receives a gradient (image encoder, g, h and the Conv4 backbone)"
    r = m.evaluate(cg.to_batch(ep), None, None, 3, torch.device("cpu"), "test")
    assert len(r) == 11 and r[6].shape == (2, 6)
    im_emb = m([ep["idx_q"], None, ep["x_q"]], im_only=True)            # forward(): raw images -> prototype space
    assert im_emb.shape == (2, 6, 6)


def test_rnn_text_encoders_keep_the_reference_surface(oracle_engine):
    """RNN / RnnHid (common.py:44-161): same state_dict keys as the reference's modules (the golden fixture stores the
    reference's own state_dict) and the same outputs on ragged token rows."""
    from fumi_amd.models.common import RNN, RnnHid
    gold = load_golden("rnn")
    dictionary = {"PAD": 0, **{f"w{i}": i for i in range(1, 30)}}
    keys = {k for k in gold if k.startswith(("embed.", "rnn."))}
    for cls, name in ((RNN, "rnn"), (RnnHid, "rnnhid")):
        m = cls("rand", "mean", dictionary, 16)
        assert set(m.state_dict()) == keys
        m.load_state_dict({k: torch.from_numpy(gold[k]) for k in keys})
        m.eval()
        out = m(torch.from_numpy(gold["tokens"]))
        assert out.shape == (2, 5, 16)
        np.testing.assert_allclose(out.numpy(), gold[name], atol=1e-6)
    m.train()
    with pytest.raises(NotImplementedError):                           # fine-tuning the LSTM is not supported by the engine
        m(torch.from_numpy(gold["tokens"]))


def test_clip_baseline_surface_and_cli_on_cpu(oracle_engine, tmp_path, monkeypatch):
    """CLIP (clip.py): state_dict keys / similarity / loss / gradients against the reference's own outputs, then `--model clip`
    through main.py on a synthetic supervised loader (training_run + zero-shot evaluate + checkpoint)."""
    from fumi_amd import main as cli
    from fumi_amd.models.clip import CLIP
    gold = load_golden("clip")
    keys = [k for k in gold if k.split(".")[0] in ("text_fc", "text_fc2", "image_fc", "image_fc2")]
    m = CLIP(text_input_dim=20, image_input_dim=48, latent_dim=16)
    assert set(m.state_dict()) == set(keys)
    m.load_state_dict({k: torch.from_numpy(gold[k]) for k in keys})
    text, image = torch.from_numpy(gold["text"]), torch.from_numpy(gold["image"])
    np.testing.assert_allclose(m(text, image).numpy(), gold["sim"], atol=1e-6)
    loss = m.loss_and_grads(text, image)
    assert abs(float(loss) - float(gold["loss"])) < 1e-6
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), gold["grad." + k], atol=1e-6)
    monkeypatch.chdir(tmp_path)
    argv = ["--model", "clip", "--dataset", "synthetic", "--disable_cuda", "--text_encoder", "BERT", "--text_emb_dim", "24",
            "--im_emb_dim", "512", "--image_embedding_model", "resnet-34", "--clip_latent_dim", "16", "--batch_size", "16",
            "--num_ways", "5", "--epochs", "3", "--lr", "1e-2", "--log_dir", str(tmp_path / "res"), "--synthetic_classes", "12"]
    res = cli.main(cli.parse_args(argv))
    assert 0.0 <= res["test_acc"] <= 1.0
    runs = os.listdir(tmp_path / "res" / "runs")
    assert os.path.exists(tmp_path / "res" / "runs" / runs[0] / "ckpt.pth.tar")


def test_cached_parameter_views_follow_replaced_parameters(oracle_engine):
    """FUMI / AM3 hand the engine the SAME lists of detached parameter views every step (hip.py validates a list once); the cache
    must notice a parameter object that was replaced, a ``param.data`` that was re-pointed and a module that was moved."""
    from fumi_amd.models.am3 import AM3
    from fumi_amd.models.fumi import FUMI
    m = FUMI(n_way=3, im_emb_dim=16, im_hid_dim=[8, 4], text_encoder="BERT", text_emb_dim=6, text_hid_dim=8, dropout_rate=0.0)
    th1, ph1, _ = m._step_params(False)
    th2, ph2, _ = m._step_params(False)
    assert th1 is th2 and ph1 is ph2                                   # stable list objects from step to step
    m.hyper_net[2].weight = torch.nn.Parameter(torch.zeros_like(m.hyper_net[2].weight))
    _, ph3, _ = m._step_params(False)
    assert ph3 is not ph1 and ph3[2].data_ptr() == m.hyper_net[2].weight.data_ptr()
    m.im_net.linear0.bias.data = torch.ones_like(m.im_net.linear0.bias)
    th4, _, _ = m._step_params(False)
    assert th4 is not th1 and th4[1].data_ptr() == m.im_net.linear0.bias.data_ptr() and float(th4[1][0]) == 1.0
    m.double(); m.float()                                              # _apply: every tensor re-created
    th5, _, _ = m._step_params(False)
    assert th5[0].data_ptr() == m.im_net.linear0.weight.data_ptr()
    a = AM3(im_encoder="precomputed", im_emb_dim=16, text_encoder="BERT", text_emb_dim=6, text_hid_dim=8, prototype_dim=4, dropout=0.0)
    w1 = a._step_params(False, 3)[0]
    a.g[3].bias = torch.nn.Parameter(torch.zeros_like(a.g[3].bias))
    w2 = a._step_params(False, 3)[0]
    assert w2 is not w1 and w2[5].data_ptr() == a.g[3].bias.data_ptr()
