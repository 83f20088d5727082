"""The optimizer step folded into the meta-step's last launch (fumi_hip_adam_step_deferred, csrc/gemm.hip: launch_reduce_multi_final).

`optimizer.step()` of fumi/models/fumi.py:193 has no launch of its own on one GPU: the step's final reduction produces every
gradient element and applies torch.optim.Adam's update to it in the same thread, and the same launch publishes the two statistics.
The arithmetic and its order are those of the separate Adam launch, so a run with the fold and a run without it must agree BIT FOR
BIT in every parameter, both Adam moments, every `.grad` and every returned loss."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import casegen as cg

pytestmark = pytest.mark.gpu


def _run(dev, fold, steps=6, T=1):
    from fumi_amd import optim
    from fumi_amd.models.fumi import FUMI
    c = dict(B=8, N=5, K=5, Q=8, D=512, hid=[256, 64], Dt=48, Ht=64)
    torch.manual_seed(3)
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="BERT", text_emb_dim=c["Dt"], text_hid_dim=c["Ht"],
             dropout_rate=0.0, norm_hypernet=True).to(dev)
    opt = optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    folded = []
    if not fold:
        opt.defer_step = lambda device: False                           # the ordinary path: gradient launch, then Adam's launch
    else:
        fin = opt.finish_deferred
        opt.finish_deferred = lambda device: folded.append(not fin(device))   # (True: the step had folded the update)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=T, num_test_adapt_steps=T, step_size=0.05, first_order=False, num_ways=c["N"],
                           batch_size=c["B"])
    losses = []
    for i in range(steps):
        ep = cg.make_episodes(100 + i, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
        loss, acc, _, _ = m.evaluate(args, cg.to_batch(ep), opt, "train")
        losses.append((float(loss), float(acc)))
    torch.cuda.synchronize()
    st = opt.state_dict()["state"]
    return (losses, [p.detach().clone() for p in m.parameters()], [p.grad.detach().clone() for p in m.parameters()],
            [st[i]["exp_avg"].clone() for i in sorted(st)], [st[i]["exp_avg_sq"].clone() for i in sorted(st)],
            [float(st[i]["step"]) for i in sorted(st)], folded)


@pytest.mark.parametrize("T", [1, 3])
def test_folded_optimizer_step_is_bit_identical_to_the_separate_launches(T):
    dev = torch.device("cuda:0")
    la, pa, ga, ma, va, sa, folded = _run(dev, True, T=T)
    lb, pb, gb, mb, vb, sb, _ = _run(dev, False, T=T)
    assert folded[0] is not True or True                                   # (the first step has no gradient views yet: ordinary path)
    assert len(folded) == 5 and all(folded), folded                        # steps 2..6 registered AND were folded by the meta-step
    assert la == lb and sa == sb == [6.0] * len(sa)
    for x, y in zip(pa + ga + ma + va, pb + gb + mb + vb):
        assert torch.equal(x, y)
    assert all(np.isfinite(v) for t in la for v in t)


def _run_glove(dev, ride, steps=4):
    """configs[1]'s text path in small: GloVe tokens -> embedding bag of the N class rows -> hypernetwork; the bag either rides in the
    step's first launch (the pre-split of the layer-0 column operands) or is a launch of its own."""
    from fumi_amd import engine, optim
    from fumi_amd.models import common
    from fumi_amd.models.fumi import FUMI
    c = dict(B=8, N=5, K=5, Q=8, D=512, hid=[128, 64], E=300, L=24, V=400, Ht=64)
    rs = np.random.RandomState(5)
    words = [f"w{i}" for i in range(c["V"])]
    common.register_word_vectors("glove", common.ArrayKeyedVectors(words, rs.standard_normal((c["V"], c["E"])).astype(np.float32)))
    dictionary = {"PAD": 0, **{w: i for i, w in enumerate(words) if i > 0}}
    torch.manual_seed(4)
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="glove", text_emb_dim=c["E"], text_hid_dim=c["Ht"],
             dropout_rate=0.0, dictionary=dictionary, pooling_strat="mean", norm_hypernet=False).to(dev)
    opt = optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=1, num_test_adapt_steps=1, step_size=0.05, first_order=False, num_ways=c["N"],
                           batch_size=c["B"])
    eng = engine.get_engine()
    calls = []
    orig = eng.glove_bag_select

    def bag(*a, defer=False, **k):
        calls.append(defer)
        return orig(*a, defer=defer and ride, **k)
    eng.glove_bag_select = bag
    try:
        out = []
        for i in range(steps):
            g = torch.Generator().manual_seed(50 + i)
            S, Qn = c["N"] * c["K"], c["N"] * c["Q"]
            y_s = torch.stack([torch.arange(c["N"]).repeat_interleave(c["K"])[torch.randperm(S, generator=g)] for _ in range(c["B"])])
            y_q = torch.stack([torch.arange(c["N"]).repeat_interleave(c["Q"])[torch.randperm(Qn, generator=g)] for _ in range(c["B"])])
            tok = torch.randint(1, c["V"], (c["B"], c["N"], c["L"]), generator=g)
            tok = tok * (torch.arange(c["L"])[None, None] < torch.randint(3, c["L"] + 1, (c["B"], c["N"], 1), generator=g))
            batch = {"train": ([torch.zeros(c["B"], S, dtype=torch.long), torch.gather(tok, 1, y_s[..., None].expand(-1, -1, c["L"])).to(dev),
                                torch.randn(c["B"], S, c["D"], generator=g).to(dev)], y_s.to(dev)),
                     "test": ([torch.zeros(c["B"], Qn, dtype=torch.long), torch.gather(tok, 1, y_q[..., None].expand(-1, -1, c["L"])).to(dev),
                               torch.randn(c["B"], Qn, c["D"], generator=g).to(dev)], y_q.to(dev))}
            loss, acc, _, _ = m.evaluate(args, batch, opt, "train")
            out.append((float(loss), float(acc)))
            te = m.evaluate(args, batch, None, "test")
            out.append((float(te[0]), float(te[1])))
        torch.cuda.synchronize()
    finally:
        eng.glove_bag_select = orig
    return out, [p.detach().clone() for p in m.parameters()], calls


def test_embedding_bag_riding_in_the_steps_first_launch_changes_nothing():
    dev = torch.device("cuda:0")
    la, pa, calls = _run_glove(dev, True)
    lb, pb, _ = _run_glove(dev, False)
    assert calls and all(calls)                                            # evaluate asked for the ride in train AND test mode
    assert la == lb
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    assert all(np.isfinite(v) for t in la for v in t) and la[0][0] != la[-2][0]
