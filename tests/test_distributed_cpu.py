"""CPU suite: episode sharding over 2 ranks (gloo) gives the same step as one process on the same meta-batch --
the N>1 path of bench.py / main.py (fumi_amd/dist.py), with the oracle standing in for the GPU engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(kind):
    from oracle import casegen as cg
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.models import maml
    from fumi_amd.models.am3 import AM3
    c = dict(B=4, N=5, K=2, Q=3, D=48, hid=[24, 12], Dt=16, Ht=12, P=10)
    if kind == "fumi_resnet12":
        # BASELINE.json configs[4]'s model through the same sharded evaluate (tiny images: the host oracle is eager autograd)
        from oracle import conv4_ref as CR
        c = dict(c, N=3, K=2, Q=2, Dt=8, Ht=6)
        ep = CR.make_image_episodes(11, c["B"], c["N"], c["K"], c["Q"], 3, 16, 16, c["Dt"])
        torch.manual_seed(5)
        m = FUMI(n_way=c["N"], im_encoder="resnet12", image_size=16, text_encoder="BERT", text_emb_dim=c["Dt"], text_hid_dim=c["Ht"],
                 norm_hypernet=False)
        return c, ep, m
    if kind == "fumi_rnn":
        # --fine_tune with a bi-LSTM text encoder (fumi.py:46-67): the LSTM's eight gradients are summed over the ranks like the rest
        from fumi_amd.models import common
        V, L, E = 30, 6, 10
        rs = np.random.RandomState(2)
        words = [f"w{i}" for i in range(V)]
        common.register_word_vectors("glove", common.ArrayKeyedVectors(words, rs.standard_normal((V, E)).astype(np.float32)))
        dictionary = {"PAD": 0, **{w: i + 1 for i, w in enumerate(words)}}
        ep = cg.make_episodes(11, c["B"], c["N"], c["K"], c["Q"], c["D"], 1, tokens=(V + 1, L, 0))
        theta, phi = cg.make_fumi_params(11, c["D"], c["hid"], c["Dt"], c["Ht"])
        torch.manual_seed(9)
        m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="RNN", text_emb_dim=c["Dt"], text_hid_dim=c["Ht"],
                 dictionary=dictionary, fine_tune=True)
        m.load_state_dict(cg.fumi_state_dict(theta, phi), strict=False)
        return c, ep, m
    ep = cg.make_episodes(11, c["B"], c["N"], c["K"], c["Q"], c["D"], c["Dt"])
    if kind == "fumi":
        theta, phi = cg.make_fumi_params(11, c["D"], c["hid"], c["Dt"], c["Ht"])
        m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="BERT", text_emb_dim=c["Dt"],
                 text_hid_dim=c["Ht"], norm_hypernet=True)
        m.load_state_dict(cg.fumi_state_dict(theta, phi))
    elif kind == "maml":
        m = maml.PureImageNetwork(im_embed_dim=c["D"], n_way=c["N"], hidden_dims=c["hid"])
        m.load_state_dict(cg.maml_state_dict(cg.make_maml_params(11, c["D"], c["hid"], c["N"])))
    else:
        m = AM3("precomputed", c["D"], "BERT", text_emb_dim=c["Dt"], text_hid_dim=c["Ht"], prototype_dim=c["P"], dropout=0.0)
        m.load_state_dict(cg.am3_state_dict(cg.make_am3_params(11, c["D"], c["Dt"], c["Ht"], c["P"])))
    return c, ep, m


def _step(kind, c, ep, m):
    from types import SimpleNamespace
    from oracle import casegen as cg
    from fumi_amd.models import maml
    args = SimpleNamespace(device=torch.device("cpu"), num_train_adapt_steps=2, num_test_adapt_steps=2,
                           step_size=cg.ALPHA, first_order=False, num_ways=c["N"], batch_size=c["B"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    if kind == "fumi_resnet12":
        # (Adam's first step is lr * sign(g): among 12 M encoder weights some gradients are zero to fp32 noise, and the sharded sum
        # may land on the other side of zero -- SGD keeps the update linear in the gradient)
        opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    if kind in ("fumi", "fumi_resnet12", "fumi_rnn"):
        tr = m.evaluate(args, cg.to_batch(ep), opt, "train")[:2]
        te = m.evaluate(args, cg.to_batch(ep), None, "test")
        extra = te[2].numpy()
    elif kind == "maml":
        tr = maml.evaluate(args, m, cg.to_batch(ep), opt, "train")
        te = maml.evaluate(args, m, cg.to_batch(ep), None, "test")
        extra = np.zeros(1)
    else:
        tr = m.evaluate(cg.to_batch(ep), opt, None, c["N"], torch.device("cpu"), "train")[:2]
        te = m.evaluate(cg.to_batch(ep), None, None, c["N"], torch.device("cpu"), "test")
        extra = np.asarray(te[6])
    params = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy()
    return np.array([float(tr[0]), float(tr[1]), float(te[0]), float(te[1])]), params, extra


def _worker(rank, world, port, kind, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from fumi_amd import engine
    from oracle_engine import OracleEngine
    engine.set_engine(OracleEngine())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c, ep, m = _build(kind)
    stats, params, extra = _step(kind, c, ep, m)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), stats=stats, params=params, extra=extra)
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["fumi", "maml", "am3", "fumi_resnet12", "fumi_rnn"])
def test_two_rank_sharded_step_equals_single_process(kind, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fumi_amd import engine
    from oracle_engine import OracleEngine
    old = engine.set_engine(OracleEngine())
    nthr = torch.get_num_threads()
    try:
        if kind == "fumi_resnet12":
            # (the ranks run one thread each; the encoder's fp32 gradients -- up to 1e2 here -- move by 4e-4 of their size with the
            # thread count of the host convolutions, so the single-process reference uses the same)
            torch.set_num_threads(1)
        c, ep, m = _build(kind)
        ref_stats, ref_params, ref_extra = _step(kind, c, ep, m)
    finally:
        torch.set_num_threads(nthr)
        engine.set_engine(old)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    np.testing.assert_array_equal(r0["params"], r1["params"])            # replicas stay bit-identical
    np.testing.assert_allclose(r0["params"], ref_params, rtol=0, atol=2e-6)
    # (resnet12: a randomly initialised 12-layer encoder with a hypernetwork head has losses of ~30 and inner-loop gradients of ~1e2:
    # two test-time inner steps amplify the 1e-6 parameter differences to a few 1e-4 of the loss)
    np.testing.assert_allclose(r0["stats"], ref_stats, rtol=2e-3 if kind == "fumi_resnet12" else 0, atol=1e-5)
    np.testing.assert_array_equal(r0["stats"], r1["stats"])
    if kind != "fumi_resnet12":
        np.testing.assert_array_equal(r0["extra"], ref_extra)              # gathered test-time predictions, full batch
    else:
        assert r0["extra"].shape == ref_extra.shape and (r0["extra"] == ref_extra).mean() > 0.9


def test_uneven_shard_is_refused():
    from fumi_amd import dist as fdist
    assert fdist.shard(7) == (0, 7)
