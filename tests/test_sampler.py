"""Episode sampler: oracle properties on CPU, device kernels against the oracle on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sampler_ref as SR


def _classes(rs, C, lo, hi):
    counts = rs.randint(lo, hi + 1, size=C)
    coi = np.repeat(np.arange(C), counts)
    rs.shuffle(coi)
    order = np.argsort(coi, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(coi, minlength=C))]).astype(np.int64)
    return coi, ptr, order.astype(np.int64)


def test_oracle_sampler_semantics():
    """N distinct classes per episode, K + Q distinct images of the right class, support and query disjoint, reproducible
    from (seed, step), different steps differ; every class / image is reachable."""
    rs = np.random.RandomState(0)
    coi, ptr, items = _classes(rs, 12, 9, 30)
    B, N, K, Q = 6, 5, 3, 4
    a = SR.sample_episodes(7, 3, B, N, K, Q, ptr, items)
    b = SR.sample_episodes(7, 3, B, N, K, Q, ptr, items)
    c = SR.sample_episodes(7, 4, B, N, K, Q, ptr, items)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[1], c[1])
    cls, it_s, it_q = a
    for e in range(B):
        assert len(set(cls[e])) == N
        for n in range(N):
            both = list(it_s[e, n]) + list(it_q[e, n])
            assert len(set(both)) == K + Q
            assert all(coi[i] == cls[e, n] for i in both)
    seen_cls, seen_img = set(), set()
    for step in range(60):
        cls, it_s, it_q = SR.sample_episodes(1, step, 4, N, K, Q, ptr, items)
        seen_cls |= set(cls.ravel()); seen_img |= set(it_s.ravel()) | set(it_q.ravel())
    assert len(seen_cls) == 12 and len(seen_img) > 0.9 * len(coi)


def test_oracle_sampler_is_uniform_enough():
    """First-moment check of Floyd + shuffle: every position of every class is drawn about equally often."""
    ptr = np.array([0, 10], np.int64); items = np.arange(10, dtype=np.int64)
    hits = np.zeros((4, 10))
    for step in range(3000):
        sel = SR.sample_distinct(SR.step_key(5, step), 0, 0, 10, 4)
        for pos, v in enumerate(sel):
            hits[pos, v] += 1
    assert np.all(np.abs(hits / 3000 - 0.1) < 0.03)
    _ = ptr, items


@pytest.mark.gpu
def test_sample_episodes_matches_oracle_bit_exact():
    from fumi_amd import hip
    dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
    rs = np.random.RandomState(1)
    for (C, lo, hi, B, N, K, Q) in [(12, 9, 30, 6, 5, 3, 4), (40, 40, 300, 32, 5, 5, 32), (7, 3, 3, 3, 7, 1, 2), (20, 20, 60, 4, 20, 5, 3)]:
        coi, ptr, items = _classes(rs, C, lo, hi)
        for step in (0, 1, 12345678901):
            cls, it_s, it_q = hip.sample_episodes(ws, 99, step, B, N, K, Q, torch.from_numpy(ptr).to(dev), torch.from_numpy(items).to(dev))
            r_cls, r_s, r_q = SR.sample_episodes(99, step, B, N, K, Q, ptr, items)
            assert ws.read_status() == 0
            assert np.array_equal(cls.cpu().numpy(), r_cls)
            assert np.array_equal(it_s.cpu().numpy(), r_s) and np.array_equal(it_q.cpu().numpy(), r_q)


@pytest.mark.gpu
def test_sample_episodes_flags_small_classes():
    from fumi_amd import hip
    dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
    ptr = torch.tensor([0, 2, 4, 6, 8, 10], dtype=torch.int64, device=dev); items = torch.arange(10, dtype=torch.int64, device=dev)
    hip.sample_episodes(ws, 1, 0, 2, 3, 2, 2, ptr, items)                      # classes of 2 images, 4 requested
    assert ws.read_status() & hip.ST_CLASS_MISSING


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((1000, 2048), torch.float32), ((37, 300), torch.float32), ((50, 7), torch.float32),
                                         ((64, 128), torch.int64), ((9, 33), torch.int32)])
def test_gather_rows_is_a_byte_copy(shape, dtype):
    from fumi_amd import hip
    dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
    g = torch.Generator().manual_seed(0)
    table = torch.randn(shape, generator=g) if dtype == torch.float32 else torch.randint(-2 ** 31, 2 ** 31 - 1, shape, generator=g, dtype=dtype)
    idx = torch.randint(0, shape[0], (513,), generator=g)
    out = hip.gather_rows(ws, table.to(dev), idx.to(dev))
    assert ws.read_status() == 0
    assert torch.equal(out.cpu(), table[idx])
    hip.gather_rows(ws, table.to(dev), torch.tensor([shape[0]], device=dev))    # out of range: flagged, not a fault
    assert ws.read_status() & hip.ST_LABEL_RANGE


@pytest.mark.gpu
def test_gpu_sampler_batches_feed_the_engine():
    """Loader contract + a training run straight from the HBM-resident table: loss drops on a learnable task."""
    from fumi_amd.dataset.gpu_sampler import GpuEpisodeSampler
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.utils import utils as U
    from types import SimpleNamespace
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(2)
    C, D, Dt, per = 30, 256, 64, 40
    mu = rs.standard_normal((C, D)).astype(np.float32)
    coi = np.repeat(np.arange(C), per); rs.shuffle(coi)
    images = torch.from_numpy(mu[coi] + 1.5 * rs.standard_normal((C * per, D)).astype(np.float32))
    P = rs.standard_normal((Dt, D)).astype(np.float32) / np.sqrt(D)
    text = torch.from_numpy(mu @ P.T)
    smp = GpuEpisodeSampler(images, coi, text, num_ways=5, num_shots=5, num_shots_test=8, batch_size=8, seed=3)
    b0, b0_again, b1 = smp.batch(0), smp.batch(0), smp.batch(1)
    (idx_s, text_s, x_s), y_s = b0['train']
    (idx_q, text_q, x_q), y_q = b0['test']
    assert x_s.shape == (8, 25, D) and x_q.shape == (8, 40, D) and text_s.shape == (8, 25, Dt) and y_s.shape == (8, 25)
    assert torch.equal(x_s, b0_again['train'][0][2]) and not torch.equal(x_s, b1['train'][0][2])
    assert torch.equal(x_s.cpu(), images[idx_s.cpu()]) and torch.equal(x_q.cpu(), images[idx_q.cpu()])
    cls_s = torch.from_numpy(coi)[idx_s.cpu()]
    assert torch.equal(text_s.cpu(), text[cls_s])
    for e in range(8):                                    # labels are the class's slot, class-major
        for n in range(5):
            assert len(set(cls_s[e, n * 5:(n + 1) * 5].tolist())) == 1 and bool((y_s[e, n * 5:(n + 1) * 5] == n).all())
    torch.manual_seed(0)
    model = FUMI(n_way=5, im_emb_dim=D, im_hid_dim=[64, 32], text_encoder="BERT", text_emb_dim=Dt, text_hid_dim=32,
                 dropout_rate=0.0).to(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=2, num_test_adapt_steps=2, step_size=0.05, first_order=False,
                           optim="adam", lr=2e-3, weight_decay=0.0, momentum=0.9, batch_size=8, num_ways=5)
    opt = U.init_optim(args, model)
    losses = [float(model.evaluate(args, smp.batch(i), opt, "train")[0]) for i in range(60)]
    assert np.mean(losses[-10:]) < np.mean(losses[:10]) - 0.1


@pytest.mark.gpu
def test_cli_with_the_resident_dataset(tmp_path, monkeypatch):
    """`--dataset synthetic-resident`: loaders are GpuEpisodeSamplers over an HBM-resident table; train -> checkpoint -> test."""
    from fumi_amd import main as cli
    monkeypatch.chdir(tmp_path)
    argv = ["--model", "fumi", "--dataset", "synthetic-resident", "--text_encoder", "BERT", "--text_emb_dim", "64", "--batch_size", "8",
            "--im_emb_dim", "512", "--image_embedding_model", "resnet-34", "--im_hid_dim", "64", "32", "--num_shots_test", "8",
            "--epochs", "40", "--eval_freq", "20", "--num_ep_test", "16", "--num_train_adapt_steps", "2",
            "--num_test_adapt_steps", "2", "--lr", "1e-3", "--dropout", "0", "--log_dir", str(tmp_path / "res"),
            "--synthetic_classes", "24", "--wandb_offline"]
    args = cli.parse_args(argv)
    res = cli.main(args)
    # 24 fixed training classes are memorised quickly (train accuracy 1.0); held-out classes only have to beat chance (0.2)
    assert np.isfinite(res["test_loss"]) and res["test_acc"] > 0.25


@pytest.mark.gpu
def test_zero_copy_step_is_bit_identical_to_the_gathered_step():
    """fumi_hip_fumi_step_indexed reads the rows where they lie in the table: same arithmetic, same bits, no gathered copy."""
    from fumi_amd import hip
    from oracle import casegen as cg
    dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
    for (B, N, K, Q, D, hid, Dt, Ht, T) in [(4, 5, 5, 8, 256, [64, 32], 24, 20, 2), (3, 5, 5, 32, 2048, [256, 64], 300, 256, 1)]:
        rs = np.random.RandomState(5)
        table = torch.from_numpy(rs.standard_normal((700, D)).astype(np.float32)).to(dev)
        S, Qn = N * K, N * Q
        idx_s = torch.from_numpy(rs.randint(0, 700, (B, S))).to(dev); idx_q = torch.from_numpy(rs.randint(0, 700, (B, Qn))).to(dev)
        y_s = torch.arange(N).repeat_interleave(K).expand(B, S).contiguous().to(dev)
        y_q = torch.from_numpy(rs.randint(0, N, (B, Qn))).to(dev)
        text = torch.from_numpy(rs.standard_normal((B, S, Dt)).astype(np.float32)).to(dev)
        theta, phi = cg.make_fumi_params(5, D, hid, Dt, Ht)
        th, ph = [t.to(dev) for t in theta], [t.to(dev) for t in phi]
        a = hip.fumi_step_select(ws, N, table[idx_s].contiguous(), y_s, table[idx_q].contiguous(), y_q, text, th, ph, T, 0.01, False)
        b = hip.fumi_step_select(ws, N, hip.RowRef(table, idx_s), y_s, hip.RowRef(table, idx_q), y_q, text, th, ph, T, 0.01, False)
        assert ws.read_status() == 0
        assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["loss_b"], b["loss_b"]) and torch.equal(a["preds"], b["preds"])
        for x, y in zip(a["g_theta"] + a["g_phi"], b["g_theta"] + b["g_phi"]):
            assert torch.equal(x, y)
    bad = idx_s.clone(); bad[0, 0] = 700
    hip.fumi_step_select(ws, N, hip.RowRef(table, bad), y_s, hip.RowRef(table, idx_q), y_q, text, th, ph, T, 0.01, False)
    assert ws.read_status() & hip.ST_LABEL_RANGE                      # flagged, read as row 0, no fault


@pytest.mark.gpu
def test_zero_copy_sampler_trains_like_the_gathering_one():
    from fumi_amd.dataset.gpu_sampler import GpuEpisodeSampler
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.utils import utils as U
    from types import SimpleNamespace
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(2)
    C, D, Dt, per = 30, 256, 64, 40
    mu = rs.standard_normal((C, D)).astype(np.float32)
    coi = np.repeat(np.arange(C), per); rs.shuffle(coi)
    images = torch.from_numpy(mu[coi] + 1.5 * rs.standard_normal((C * per, D)).astype(np.float32))
    text = torch.from_numpy(mu @ (rs.standard_normal((Dt, D)).astype(np.float32) / np.sqrt(D)).T)
    losses = {}
    for zc in (False, True):
        smp = GpuEpisodeSampler(images, coi, text, num_ways=5, num_shots=5, num_shots_test=8, batch_size=8, seed=3, zero_copy=zc)
        torch.manual_seed(0)
        model = FUMI(n_way=5, im_emb_dim=D, im_hid_dim=[64, 32], text_encoder="BERT", text_emb_dim=Dt, text_hid_dim=32,
                     dropout_rate=0.0).to(dev)
        args = SimpleNamespace(device=dev, num_train_adapt_steps=2, num_test_adapt_steps=2, step_size=0.05, first_order=False,
                               optim="adam", lr=2e-3, weight_decay=0.0, momentum=0.9, batch_size=8, num_ways=5)
        opt = U.init_optim(args, model)
        losses[zc] = [float(model.evaluate(args, smp.batch(i), opt, "train")[0]) for i in range(30)]
    assert losses[True] == losses[False]                                # identical trajectories, bit for bit


def test_gpu_sampler_refuses_underpopulated_classes():
    """torchmeta's ClassSplitter raises ValueError for a class with fewer than K + Q images; the resident sampler raises it in
    its constructor (before touching the GPU) when asked to be strict; by default it warns and leaves such classes out (a
    dataset the reference accepts still loads), never wrapping indices and leaking support rows into the query set."""
    from fumi_amd.dataset.gpu_sampler import GpuEpisodeSampler
    coi = np.concatenate([np.repeat(np.arange(6), 20), np.repeat([6], 5)])          # class 6 has 5 images, class 7 none
    images, text = torch.zeros(len(coi), 8), torch.zeros(8, 4)
    with pytest.raises(ValueError, match="fewer than num_shots"):
        GpuEpisodeSampler(images, coi, text, num_ways=5, num_shots=5, num_shots_test=8, batch_size=2, skip_small_classes=False)
    with pytest.raises(ValueError, match="fewer than num_ways classes"):
        GpuEpisodeSampler(images, coi, text, num_ways=7, num_shots=5, num_shots_test=8, batch_size=2, skip_small_classes=True)


@pytest.mark.gpu
def test_gpu_sampler_skips_small_classes_on_request():
    from fumi_amd import hip
    from fumi_amd.dataset.gpu_sampler import GpuEpisodeSampler
    rs = np.random.RandomState(5)
    coi = np.concatenate([np.repeat(np.arange(6), 20), np.repeat([6], 5), np.repeat([8], 30)]); rs.shuffle(coi)   # 6 small, 7 empty
    images = torch.from_numpy(rs.standard_normal((len(coi), 16)).astype(np.float32))
    text = torch.arange(9, dtype=torch.float32)[:, None].repeat(1, 4)
    smp = GpuEpisodeSampler(images, coi, text, num_ways=5, num_shots=5, num_shots_test=8, batch_size=16, seed=1,
                            skip_small_classes=True)
    seen = set()
    for step in range(8):
        b = smp.batch(step)
        (idx_s, text_s, x_s), _ = b['train']
        (idx_q, text_q, x_q), _ = b['test']
        cls_s, cls_q = torch.from_numpy(coi)[idx_s.cpu()], torch.from_numpy(coi)[idx_q.cpu()]
        assert torch.equal(text_s[..., 0].cpu(), cls_s.float()) and torch.equal(text_q[..., 0].cpu(), cls_q.float())
        for e in range(16):                               # support and query never share an image
            assert not set(idx_s[e].tolist()) & set(idx_q[e].tolist())
            assert len(set(idx_s[e].tolist())) == 25 and len(set(idx_q[e].tolist())) == 40
        seen |= set(cls_s.unique().tolist())
    assert seen <= {0, 1, 2, 3, 4, 5, 8} and 8 in seen
    assert hip.Workspace.get(torch.device("cuda:0")).read_status() == 0


@pytest.mark.gpu
def test_torchmeta_task_semantics_match_oracle_and_hold_their_properties():
    """fumi_hip_sample_episodes_tm: bit-exact against oracle/sampler_ref.py; labels are a permutation of 0..N-1 per task
    (torchmeta's Categorical); with fixed_split a class tuple drawn again has the same support / query members (ClassSplitter's
    hash(task) + seed seeding), without it the members change from step to step."""
    from fumi_amd import hip
    dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
    rs = np.random.RandomState(4)
    coi, ptr, items = _classes(rs, 6, 12, 30)                       # 6 classes, 5-way: tuples repeat quickly
    tp, ti = torch.from_numpy(ptr).to(dev), torch.from_numpy(items).to(dev)
    seen, seen_free = {}, {}
    for step in range(12):
        for fixed in (True, False):
            cls, lab, it_s, it_q = hip.sample_episodes_tm(ws, 7, step, 8, 5, 2, 3, tp, ti, fixed)
            r = SR.sample_episodes_tm(7, step, 8, 5, 2, 3, ptr, items, fixed)
            for a, b in zip((cls, lab, it_s, it_q), r):
                assert np.array_equal(a.cpu().numpy(), b)
            assert all(sorted(row) == list(range(5)) for row in lab.cpu().tolist())
            store = seen if fixed else seen_free
            for b in range(8):
                key = tuple(cls[b].tolist())
                val = (it_s[b].cpu().numpy().tobytes(), it_q[b].cpu().numpy().tobytes())
                store.setdefault(key, set()).add(val)
    assert ws.read_status() == 0
    assert any(len(v) > 0 for v in seen.values()) and all(len(v) == 1 for v in seen.values())       # one split per class tuple
    assert any(len(v) > 1 for v in seen_free.values())


@pytest.mark.gpu
def test_gpu_sampler_torchmeta_tasks_feed_the_engine():
    from fumi_amd.dataset.gpu_sampler import GpuEpisodeSampler
    rs = np.random.RandomState(2)
    C, D, Dt, per = 12, 64, 16, 30
    coi = np.repeat(np.arange(C), per); rs.shuffle(coi)
    images = torch.from_numpy(rs.standard_normal((C * per, D)).astype(np.float32))
    text = torch.arange(C, dtype=torch.float32)[:, None].repeat(1, Dt)
    smp = GpuEpisodeSampler(images, coi, text, num_ways=5, num_shots=2, num_shots_test=3, batch_size=4, seed=3, torchmeta_tasks=True)
    b = smp.batch(0)
    (idx_s, text_s, x_s), y_s = b['train']
    (idx_q, text_q, x_q), y_q = b['test']
    cls_s = torch.from_numpy(coi)[idx_s.cpu()]
    for e in range(4):
        assert sorted(y_s[e].cpu().tolist()) == sorted(list(range(5)) * 2) and sorted(y_q[e].cpu().tolist()) == sorted(list(range(5)) * 3)
        # one label per class, the same in support and query, text row = the class's
        m = {int(c): int(y) for c, y in zip(cls_s[e], y_s[e].cpu())}
        assert len(m) == 5 and len(set(m.values())) == 5
        cls_q = torch.from_numpy(coi)[idx_q[e].cpu()]
        assert all(m[int(c)] == int(y) for c, y in zip(cls_q, y_q[e].cpu()))
    assert torch.equal(text_s[..., 0].cpu(), cls_s.float())
