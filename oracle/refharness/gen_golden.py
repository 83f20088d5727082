"""Golden-vector generator: runs the REAL reference (/root/reference, imported unmodified through
``stubs.py``) on the cases of ``oracle/casegen.py`` and writes ``tests/golden/*.npz``.

TEST INFRASTRUCTURE ONLY.  Run in the build container (the reference never travels to the GPU box):

    python -m oracle.refharness.gen_golden            # regenerates every fixture

``FUMI.evaluate`` (fumi/models/fumi.py:115-196) is run byte-for-byte unmodified inside
``torch.autograd.graph.allow_mutation_on_saved_tensors()``: on torch>=2 the in-place
``hyper_params -=`` at :168 otherwise trips the saved-tensor version check at :172.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import casegen as cg                      # noqa: E402
from oracle.refharness import stubs                   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _args(T, first_order=False, n_way=5):
    return SimpleNamespace(device=torch.device("cpu"), num_train_adapt_steps=T, num_test_adapt_steps=T,
                           step_size=cg.ALPHA, first_order=first_order, num_ways=n_way, batch_size=1)


def _grad_entries(prefix, named_params, full):
    out = {}
    for name, p in named_params:
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"{prefix}.{name}.digest"] = cg.digest(g)
        if full or g.numel() <= 70000:
            out[f"{prefix}.{name}"] = g.detach().numpy().copy()
    return out


def gen_fumi(ref_fumi, name, c):
    torch.manual_seed(0)
    seed = sum(ord(ch) for ch in name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"], blocked=c["blocked"])
    theta, phi = cg.make_fumi_params(seed, c["D"], c["hid"], c["Dt"], c["Ht"])
    model = ref_fumi.FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="BERT",
                          text_emb_dim=c["Dt"], text_hid_dim=c["Ht"], dropout_rate=0.0,
                          norm_hypernet=c["tanh"], init_bias=c["init_bias"])
    sd = cg.fumi_state_dict(theta, phi)
    extra = {}
    if c["init_bias"]:
        # keep the reference's own initialiser output for the hypernet head (hypernet_init.py live path)
        extra["init_head_weight"] = model.hyper_net[2].weight.detach().numpy().copy()
        extra["init_head_bias"] = model.hyper_net[2].bias.detach().numpy().copy()
        sd["hyper_net.2.weight"] = model.hyper_net[2].weight.detach().clone()
        sd["hyper_net.2.bias"] = model.hyper_net[2].bias.detach().clone()
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)   # utils.py:280-283 defaults

    calls = []
    orig = model.im_forward

    def rec(im, p, h):
        o = orig(im, p, h)
        calls.append(o.detach().clone())
        return o
    model.im_forward = rec

    with torch.autograd.graph.allow_mutation_on_saved_tensors():
        loss, acc, preds, tgt = model.evaluate(_args(c["T"], n_way=c["N"]), cg.to_batch(ep), opt, "train")
    T = c["T"]
    logits_q = torch.stack([calls[b * (T + 1) + T] for b in range(c["B"])])
    out = dict(loss=np.float64(loss), acc=np.float64(acc), preds=preds.numpy().astype(np.int64),
               logits_q=logits_q.numpy(), seed=np.int64(seed),
               in_digest=cg.digest(torch.cat([ep["x_s"].reshape(-1), ep["x_q"].reshape(-1), ep["text_s"].reshape(-1)])),
               **extra)
    out.update(_grad_entries("grad", model.named_parameters(), full=False))
    for n, p in model.named_parameters():
        out[f"post.{n}.digest"] = cg.digest(p)
    # eval-mode call on the post-step parameters (task="test", fumi.py:126-127,154): loss/acc/preds only
    calls.clear()
    with torch.autograd.graph.allow_mutation_on_saved_tensors():
        l2, a2, p2, _ = model.evaluate(_args(c["T"], n_way=c["N"]), cg.to_batch(ep), None, "test")
    out.update(test_loss=np.float64(l2), test_acc=np.float64(a2), test_preds=p2.numpy().astype(np.int64))
    np.savez(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={float(loss):.6f} acc={float(acc):.4f}")


def gen_maml(ref_maml, name, c):
    seed = sum(ord(ch) for ch in name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], 8)
    p = cg.make_maml_params(seed, c["D"], c["hid"], c["N"])
    model = ref_maml.PureImageNetwork(im_embed_dim=c["D"], n_way=c["N"], hidden_dims=c["hid"])
    model.load_state_dict(cg.maml_state_dict(p))
    opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)
    outs = []
    hook = model.register_forward_hook(lambda m, i, o: outs.append(o.detach().clone()))
    loss, acc = ref_maml.evaluate(_args(c["T"], c["first_order"], c["N"]), model, cg.to_batch(ep), opt, "train")
    hook.remove()
    T = c["T"]
    logits_q = torch.stack([outs[b * (T + 1) + T] for b in range(c["B"])])
    out = dict(loss=np.float64(loss), acc=np.float64(acc), logits_q=logits_q.numpy(), seed=np.int64(seed),
               preds=logits_q.max(-1)[1].numpy(),
               in_digest=cg.digest(torch.cat([ep["x_s"].reshape(-1), ep["x_q"].reshape(-1)])))
    out.update(_grad_entries("grad", model.named_parameters(), full=False))
    np.savez(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={float(loss):.6f} acc={float(acc):.4f}")


def gen_am3(ref_am3, name, c):
    seed = sum(ord(ch) for ch in name)
    ep = cg.make_episodes(seed, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    w = cg.make_am3_params(seed, c["D"], c["Dt"], c["Ht"], c["P"])
    model = ref_am3.AM3(im_encoder="precomputed", im_emb_dim=c["D"], text_encoder="BERT", text_emb_dim=c["Dt"],
                        text_hid_dim=c["Ht"], prototype_dim=c["P"], dropout=0.0, lamda_fixed=c["lamda_fixed"])
    model.load_state_dict(cg.am3_state_dict(w))
    opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)
    loss, acc, f1, prec, rec, avg_lam = model.evaluate(cg.to_batch(ep), opt, None, c["N"], torch.device("cpu"), "train")
    out = dict(loss=np.float64(loss), acc=np.float64(acc), f1=np.float64(f1), prec=np.float64(prec),
               rec=np.float64(rec), avg_lamda=np.float64(avg_lam), seed=np.int64(seed),
               in_digest=cg.digest(torch.cat([ep["x_s"].reshape(-1), ep["x_q"].reshape(-1), ep["text_s"].reshape(-1)])))
    out.update(_grad_entries("grad", model.named_parameters(), full=False))
    # test-mode call on the post-step parameters: 11-tuple incl. preds and per-support lamda (am3.py:202-209)
    with torch.no_grad():
        r = model.evaluate(cg.to_batch(ep), None, None, c["N"], torch.device("cpu"), "test")
    out.update(test_loss=np.float64(r[0]), test_acc=np.float64(r[1]), test_preds=np.asarray(r[6]).astype(np.int64),
               test_lamda_s=np.asarray(r[10]), test_avg_lamda=np.float64(r[5]))
    for n, p in model.named_parameters():
        out[f"post.{n}.digest"] = cg.digest(p)
    np.savez(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={float(loss):.6f} acc={float(acc):.4f} lam={float(avg_lam):.4f}")


class _FakeKV:
    """Stand-in for a gensim KeyedVectors table (tiny vocabulary; the real one needs a download)."""

    def __init__(self, words, dim, seed):
        rs = np.random.RandomState(seed)
        self.vector_size = dim
        self.key_to_index = {w: i for i, w in enumerate(words)}
        self._v = rs.standard_normal((len(words), dim)).astype(np.float32)

    def __getitem__(self, w):
        return self._v[self.key_to_index[w]]


def gen_wordemb(ref_common):
    """WordEmbedding (common.py:8-41) mean and max pooling incl. rows with a single non-PAD token."""
    dim, L = 12, 9
    known = [f"w{i}" for i in range(30)]
    dictionary = {"PAD": 0}
    for i in range(40):                       # w30..w39 are OOV -> np.random rows (a fixture, not a formula)
        dictionary[f"w{i}"] = i + 1
    kv = _FakeKV(known, dim, 5)
    stubs.install()
    import gensim.downloader as api
    api.load = lambda name: kv
    ref_common.api = api
    np.random.seed(11)
    rs = np.random.RandomState(3)
    tokens = np.zeros((2, 7, L), dtype=np.int64)
    for b in range(2):
        for s in range(7):
            ln = 1 if s == 0 else rs.randint(1, L + 1)
            tokens[b, s, :ln] = rs.randint(1, 41, size=ln)
    out = dict(tokens=tokens, pad=np.int64(0))
    for mode in ("mean", "max"):
        np.random.seed(11)
        we = ref_common.WordEmbedding("glove", mode, dictionary)
        out["table"] = we.embed.weight.detach().numpy().copy()
        out[mode] = we(torch.from_numpy(tokens)).detach().numpy()
    np.savez(os.path.join(OUT, "wordemb.npz"), **out)
    print("wordemb: table", out["table"].shape)


def gen_clip():
    """CLIP baseline (clip.py): similarity matrix, symmetric-CE loss, all eight gradients, and the zero-shot call shape."""
    ref_clip = stubs.import_reference_clip()
    torch.manual_seed(0)
    n, Dt, D, P = 6, 20, 48, 16
    m = ref_clip.CLIP(text_input_dim=Dt, image_input_dim=D, latent_dim=P)
    rs = np.random.RandomState(17)
    text, image = torch.from_numpy(rs.standard_normal((n, Dt)).astype(np.float32)), torch.from_numpy(rs.standard_normal((n, D)).astype(np.float32))
    out = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    sim = m(text, image)
    labels = torch.arange(n)
    loss = (torch.nn.CrossEntropyLoss()(sim, labels) + torch.nn.CrossEntropyLoss()(sim.T, labels)) / 2.
    loss.backward()
    out.update(text=text.numpy(), image=image.numpy(), sim=sim.detach().numpy(), loss=np.float32(loss.item()))
    for k, p in m.named_parameters():
        out["grad." + k] = p.grad.numpy().copy()
    out["zero_shot"] = m(text[:1], image[:5]).detach().numpy()            # evaluate(): one text row against n_ways images
    np.savez(os.path.join(OUT, "clip.npz"), **out)
    print("clip: loss", float(loss))


def gen_rnn(ref_common):
    """bi-LSTM text encoders (common.py:44-161) with embedding_type='rand' (no gensim): RNN (output states) and RnnHid (cell
    states) on ragged token rows, same weights."""
    torch.manual_seed(3)
    V, L, hid = 30, 9, 16
    dictionary = {"PAD": 0, **{f"w{i}": i for i in range(1, V)}}
    rs = np.random.RandomState(5)
    tokens = np.zeros((2, 5, L), dtype=np.int64)
    for b in range(2):
        for s_ in range(5):
            ln = 1 if s_ == 0 else (L if s_ == 1 else rs.randint(1, L + 1))
            tokens[b, s_, :ln] = rs.randint(1, V, size=ln)
    r = ref_common.RNN("rand", "mean", dictionary, hid)
    rh = ref_common.RnnHid("rand", "mean", dictionary, hid)
    rh.load_state_dict(r.state_dict())
    out = {k: v.detach().numpy().copy() for k, v in r.state_dict().items()}
    with torch.no_grad():
        out.update(tokens=tokens, rnn=r(torch.from_numpy(tokens)).numpy(), rnnhid=rh(torch.from_numpy(tokens)).numpy())
    np.savez(os.path.join(OUT, "rnn.npz"), **out)
    print("rnn:", out["rnn"].shape)


def gen_fumi_rnn_finetune(ref_fumi, ref_common, ref_am3=None):
    """FUMI(text_encoder="RNN" / "RNNhid", fine_tune=True) (fumi.py:46-67): one training meta-batch through the unmodified
    ``evaluate`` -- the loss reaches the bi-LSTM through get_hyper_params (fumi.py:196-212), so .grad of rnn.* is the fixture
    the engine's LSTM backward is held to.  Token rows are distinct per support SAMPLE (the class text is the class's first
    support row), ragged, incl. a one-token and a full-length row; the word table is the stub KeyedVectors' (frozen)."""
    B, N, K, Q, D, hid, Dt, Ht, T, L, E = 3, 4, 2, 3, 40, [24], 16, 32, 2, 7, 12
    known = [f"w{i}" for i in range(30)]
    dictionary = {"PAD": 0, **{f"w{i}": i + 1 for i in range(30)}}
    kv = _FakeKV(known, E, 9)
    stubs.install()
    import gensim.downloader as api
    api.load = lambda name: kv
    ref_common.api = api
    seed = 4242
    ep = cg.make_episodes(seed, B, N, K, Q, D, Dt, blocked=False)
    rs = np.random.RandomState(seed + 1)
    S, Qn = N * K, N * Q

    def rows(n):
        t = np.zeros((B, n, L), dtype=np.int64)
        for b in range(B):
            for s_ in range(n):
                ln = 1 if s_ == 0 else (L if s_ == 1 else rs.randint(1, L + 1))
                t[b, s_, :ln] = rs.randint(1, 31, size=ln)
        return torch.from_numpy(t)
    ep["text_s"], ep["text_q"] = rows(S), rows(Qn)
    theta, phi = cg.make_fumi_params(seed, D, hid, Dt, Ht)
    out = dict(seed=np.int64(seed), text_s=ep["text_s"].numpy(), text_q=ep["text_q"].numpy(),
               dims=np.asarray([B, N, K, Q, D, hid[0], Dt, Ht, T, L, E], dtype=np.int64))
    rnn_sd = None
    for enc in ("RNN", "RNNhid"):
        torch.manual_seed(21)
        model = ref_fumi.FUMI(n_way=N, im_emb_dim=D, im_hid_dim=hid, text_encoder=enc, text_emb_dim=Dt, text_hid_dim=Ht,
                              dropout_rate=0.0, dictionary=dictionary, pooling_strat="mean", norm_hypernet=True, fine_tune=True)
        sd = cg.fumi_state_dict(theta, phi)
        if rnn_sd is None:
            rnn_sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("text_encoder.")}
            out.update({k: v.numpy().copy() for k, v in rnn_sd.items()})
        sd.update(rnn_sd)
        model.load_state_dict(sd)
        assert [n for n, p_ in model.named_parameters() if p_.requires_grad and n.startswith("text_encoder.")] == \
            [f"text_encoder.rnn.{n}" for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
                                              "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse")]
        opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)
        with torch.autograd.graph.allow_mutation_on_saved_tensors():
            loss, acc, preds, _ = model.evaluate(_args(T, n_way=N), cg.to_batch(ep), opt, "train")
        out[f"{enc}.loss"], out[f"{enc}.acc"] = np.float64(loss), np.float64(acc)
        out[f"{enc}.preds"] = preds.numpy().astype(np.int64)
        for n, p_ in model.named_parameters():
            if p_.requires_grad:
                out[f"{enc}.grad.{n}"] = p_.grad.detach().numpy().copy()
                out[f"{enc}.post.{n}.digest"] = cg.digest(p_)
        print(f"fumi_rnn_finetune[{enc}]: loss={float(loss):.6f} acc={float(acc):.4f}")
    # the same episodes and the same LSTM through AM3(text_encoder=RNN / RNNhid, fine_tune=True) (am3.py:61-76,113-126): there EVERY
    # support row's encoding feeds its class prototype, so all B*S rows carry an adjoint
    if ref_am3 is not None:
        P = 20
        w = cg.make_am3_params(seed, D, Dt, Ht, P)
        out["am3_P"] = np.int64(P)
        for enc in ("RNN", "RNNhid"):
            torch.manual_seed(21)
            model = ref_am3.AM3(im_encoder="precomputed", im_emb_dim=D, text_encoder=enc, text_emb_dim=Dt, text_hid_dim=Ht,
                                prototype_dim=P, dropout=0.0, fine_tune=True, dictionary=dictionary, pooling_strat="mean")
            sd = cg.am3_state_dict(w)
            sd.update(rnn_sd)
            model.load_state_dict(sd)
            opt = torch.optim.Adam(model.parameters(), lr=3e-5, weight_decay=5e-4)
            r = model.evaluate(cg.to_batch(ep), opt, None, N, torch.device("cpu"), "train")
            out[f"am3.{enc}.loss"], out[f"am3.{enc}.acc"], out[f"am3.{enc}.avg_lamda"] = np.float64(r[0]), np.float64(r[1]), np.float64(r[5])
            for n, p_ in model.named_parameters():
                if p_.requires_grad:
                    out[f"am3.{enc}.grad.{n}"] = p_.grad.detach().numpy().copy()
                    out[f"am3.{enc}.post.{n}.digest"] = cg.digest(p_)
            print(f"fumi_rnn_finetune[am3 {enc}]: loss={float(r[0]):.6f} acc={float(r[1]):.4f} lam={float(r[5]):.4f}")
    np.savez(os.path.join(OUT, "fumi_rnn_finetune.npz"), **out)


def gen_surface(ref_fumi, ref_maml, ref_am3, ref_utils):
    """The drop-in surface: every CLI flag's default/type (utils.py:19-229) and the state_dict keys/shapes of the three
    models at the CLI defaults -> tests/golden/surface.json."""
    import json
    p = ref_utils.parser()
    flags = {}
    for a in p._actions:
        if not a.option_strings or a.dest == "help":
            continue
        flags[a.dest] = dict(flag=a.option_strings[0], default=a.default,
                             type=(a.type.__name__ if a.type else None), nargs=a.nargs,
                             choices=list(a.choices) if a.choices else None,
                             store_true=type(a).__name__ == "_StoreTrueAction")
    d = p.parse_args([])
    f = ref_fumi.FUMI(n_way=d.num_ways, im_emb_dim=d.im_emb_dim, im_hid_dim=d.im_hid_dim, text_encoder="BERT",
                      text_emb_dim=d.text_emb_dim, text_hid_dim=d.text_hid_dim, dropout_rate=d.dropout,
                      norm_hypernet=d.norm_hypernet)
    m = ref_maml.PureImageNetwork(im_embed_dim=d.im_emb_dim, n_way=d.num_ways, hidden_dims=d.im_hid_dim)
    a3 = ref_am3.AM3(im_encoder=d.im_encoder, im_emb_dim=d.im_emb_dim, text_encoder="BERT", text_emb_dim=d.text_emb_dim,
                     text_hid_dim=d.text_hid_dim, prototype_dim=d.prototype_dim, dropout=d.dropout)
    sd = lambda mod: {k: list(v.shape) for k, v in mod.state_dict().items()}
    json.dump(dict(flags=flags, fumi=sd(f), maml=sd(m), am3=sd(a3)), open(os.path.join(OUT, "surface.json"), "w"),
              indent=1, sort_keys=True)
    print("surface:", len(flags), "flags")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_fumi, ref_maml, ref_am3, ref_utils, ref_common = stubs.import_reference()
    torch.set_num_threads(1)                  # deterministic summation order for the fixtures
    only = set(sys.argv[1:])                  # optional: names of the cases to (re)generate; default every fixture
    for name, c in cg.FUMI_CASES.items():
        if not only or name in only:
            gen_fumi(ref_fumi, name, c)
    for name, c in cg.MAML_CASES.items():
        if not only or name in only:
            gen_maml(ref_maml, name, c)
    for name, c in cg.AM3_CASES.items():
        if not only or name in only:
            gen_am3(ref_am3, name, c)
    if not only or "wordemb" in only:
        gen_wordemb(ref_common)
    if not only or "clip" in only:
        gen_clip()
    if not only or "rnn" in only:
        gen_rnn(ref_common)
    if not only or "fumi_rnn_finetune" in only:
        gen_fumi_rnn_finetune(ref_fumi, ref_common, ref_am3)
    if not only or "surface" in only:
        gen_surface(ref_fumi, ref_maml, ref_am3, ref_utils)


if __name__ == "__main__":
    main()
