"""GPU suite: the sharded path of bench.py / main.py with the real HIP engine -- two ranks (both on cuda:0, gloo carrying the
all-reduce: RCCL refuses two ranks on one device and the test box has one GPU) against one process on the same meta-batch.
Covers what the CPU twin (test_distributed_cpu.py, oracle engine) cannot: device shards, the flat gradient buffer written by
the library, the fused Adam step and the event-free statistics on every rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_distributed_cpu import _build, _free_port  # noqa: E402


def _step_gpu(kind, c, ep, m, dev):
    from types import SimpleNamespace
    from oracle import casegen as cg
    from fumi_amd.models import maml
    from fumi_amd.optim import Adam
    m.to(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=2, num_test_adapt_steps=2, step_size=cg.ALPHA, first_order=False,
                           num_ways=c["N"], batch_size=c["B"])
    opt = Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    batch = cg.to_batch(ep)
    stats = []
    for _ in range(3):                                   # several steps: replicas must stay together
        if kind == "fumi":
            tr = m.evaluate(args, batch, opt, "train")[:2]
        elif kind == "maml":
            tr = maml.evaluate(args, m, batch, opt, "train")
        else:
            tr = m.evaluate(batch, opt, None, c["N"], dev, "train")[:2]
        stats += [float(tr[0]), float(tr[1])]
    if kind == "fumi":
        te = m.evaluate(args, batch, None, "test")
        extra = te[2].cpu().numpy()
    elif kind == "maml":
        te = maml.evaluate(args, m, batch, None, "test")
        extra = np.zeros(1)
    else:
        te = m.evaluate(batch, None, None, c["N"], dev, "test")
        extra = np.asarray(te[6])
    stats += [float(te[0]), float(te[1])]
    params = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy()
    return np.array(stats), params, extra


def _worker(rank, world, port, kind, out_dir, backend="gloo"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dev = torch.device("cuda", rank if backend == "nccl" else 0)      # RCCL: one device per rank
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c, ep, m = _build(kind)
        stats, params, extra = _step_gpu(kind, c, ep, m, dev)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), stats=stats, params=params, extra=extra)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fumi", "maml", "am3"])
def test_two_rank_sharded_gpu_step_equals_single_process(kind, tmp_path):
    dev = torch.device("cuda", 0)
    c, ep, m = _build(kind)
    ref_stats, ref_params, ref_extra = _step_gpu(kind, c, ep, m, dev)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    np.testing.assert_array_equal(r0["params"], r1["params"])            # replicas stay bit-identical
    np.testing.assert_allclose(r0["params"], ref_params, rtol=0, atol=5e-6)
    np.testing.assert_allclose(r0["stats"], ref_stats, rtol=0, atol=2e-5)
    np.testing.assert_array_equal(r0["stats"], r1["stats"])
    np.testing.assert_array_equal(r0["extra"], ref_extra)                  # gathered test-time predictions, full batch


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs two GPUs (one rank per device)")
@pytest.mark.parametrize("kind", ["fumi"])
def test_two_rank_rccl_step_equals_single_process(kind, tmp_path):
    """The production backend: `nccl` (= RCCL over xGMI), one rank per GPU, three training steps + a test step against one
    process on the same meta-batch.  Skipped on the one-GPU test box; runs wherever two devices are visible."""
    dev = torch.device("cuda", 0)
    c, ep, m = _build(kind)
    ref_stats, ref_params, ref_extra = _step_gpu(kind, c, ep, m, dev)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path), "nccl"), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    np.testing.assert_array_equal(r0["params"], r1["params"])
    np.testing.assert_allclose(r0["params"], ref_params, rtol=0, atol=5e-6)
    np.testing.assert_allclose(r0["stats"], ref_stats, rtol=0, atol=2e-5)
    np.testing.assert_array_equal(r0["extra"], ref_extra)


@pytest.mark.gpu
def test_bench_rehearsal_runs_the_two_rank_code_path():
    """`FUMI_BENCH_REHEARSAL=1 python bench.py --gpus 2` (how the driver launches N > 1, on the one GPU of this box: both ranks on
    cuda:0, gloo carries the all-reduce): the self-launching parent, the per-rank sharding, the flat-gradient all-reduce and the
    max-over-ranks timing all execute; rank 0 prints one JSON line with the all-reduce object."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUMI_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-as-worded", "--no-configs4", "--no-extra"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_meta_batch"] == 64 and out["value"] > 0
    assert out["allreduce"]["bytes"] > 2_000_000 and out["allreduce"]["avg_us"] > 0 and "REHEARSAL" in out["data"]
    # the line itself proves N ranks took part: world size, the collective's backend, every rank's device and episode share
    assert out["world_size"] == 2 and out["allreduce"]["torch_backend"] == "gloo"
    assert sorted(r_["rank"] for r_ in out["ranks"]) == [0, 1] and all(r_["episodes"] == 32 for r_ in out["ranks"])
    assert out["repeats"]["n"] >= 3 and out["repeats"]["ms_per_step_min"] <= out["ms_per_step"] <= out["repeats"]["ms_per_step_max"]
