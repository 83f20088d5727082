"""Test accuracy vs the reference path (BASELINE.json metric: "... test acc vs reference", north_star: within +-0.2 pp).

SURVEY.md section 8(d), "learnable variant for accuracy": both sides are trained from the identical initialisation on the
identical sampled episodes (class prototypes mu_c, x = mu_c + 2 eps, text = P mu_c + 0.5 eps) -- this repo's FUMI module
through the HIP engine on the GPU, and the same module through the oracle engine (the CPU restatement of
fumi/models/fumi.py:115-196 that the golden fixtures pin to the real reference) -- then tested on the identical held-out
episodes.  The short form runs with the GPU suite; FUMI_ACC_FULL=1 runs BASELINE.json's configs[1] shapes for 1500
meta-steps / 1024 test episodes and writes gpurun_out/accuracy_vs_reference.json (committed under profiles/)."""
import json
import os
import time
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from fumi_amd import engine as E
from fumi_amd.dataset.synthetic import SyntheticEpisodes
from fumi_amd.models import common
from fumi_amd.models.fumi import FUMI
from fumi_amd.utils import utils as U
from oracle_engine import OracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHORT = dict(N=5, K=5, Q=8, Q_test=20, D=256, hid=[64, 32], Dt=64, Ht=64, T=1, T_test=2, B=8, steps=120, test_batches=25,
             classes=40, lr=1e-3, tokens=None)
FULL = dict(N=5, K=5, Q=32, Q_test=20, D=2048, hid=[256, 64], Dt=300, Ht=256, T=1, T_test=5, B=32, steps=1500, test_batches=32,
            classes=64, lr=1e-3, tokens=(2000, 32, 0))         # configs[1]: GloVe token text, 1 inner step, meta-batch 32


def _build(c, device):
    dictionary = None
    if c["tokens"] is not None:
        V = c["tokens"][0]
        words = [f"tok{i}" for i in range(1, V)]
        vecs = np.random.RandomState(5).standard_normal((V - 1, c["Dt"])).astype(np.float32)
        common.register_word_vectors("glove", common.ArrayKeyedVectors(words, vecs))
        dictionary = {"PAD": 0}
        dictionary.update({w: i + 1 for i, w in enumerate(words)})
    torch.manual_seed(11)
    np.random.seed(11)
    m = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="glove" if dictionary else "BERT",
             text_emb_dim=c["Dt"], text_hid_dim=c["Ht"], dropout_rate=0.0, dictionary=dictionary, pooling_strat="mean",
             norm_hypernet=False)
    return m.to(device)


def _run(c, dev, progress=None):
    gpu = _build(c, dev)
    cpu = _build(c, torch.device("cpu"))
    cpu.load_state_dict({k: v.cpu() for k, v in gpu.state_dict().items()})          # identical initialisation
    mk = lambda d: SimpleNamespace(device=d, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T_test"], step_size=0.01,
                                   first_order=False, optim="adam", lr=c["lr"], weight_decay=5e-4, momentum=0.9,
                                   batch_size=c["B"], num_ways=c["N"])
    a_gpu, a_cpu = mk(dev), mk(torch.device("cpu"))
    o_gpu, o_cpu = U.init_optim(a_gpu, gpu), U.init_optim(a_cpu, cpu)
    train = SyntheticEpisodes(c["classes"], c["D"], c["Dt"], c["N"], c["K"], c["Q"], c["B"], 3, "train", c["tokens"])
    test = SyntheticEpisodes(c["classes"], c["D"], c["Dt"], c["N"], c["K"], c["Q_test"], c["B"], 3, "test", c["tokens"])
    oracle, hip_engine = OracleEngine(), E.get_engine()
    torch.set_num_threads(min(16, os.cpu_count() or 1))       # the eager oracle is dispatch-bound: more threads only contend
    t_gpu = t_cpu = 0.0
    curve = []
    for i in range(c["steps"]):
        batch = train.batch(i)
        t0 = time.perf_counter()
        lg, ag, _, _ = gpu.evaluate(a_gpu, batch, o_gpu, "train")
        lg = float(lg); t_gpu += time.perf_counter() - t0
        old = E.set_engine(oracle)
        try:
            t0 = time.perf_counter()
            lc, ac, _, _ = cpu.evaluate(a_cpu, batch, o_cpu, "train")
            lc = float(lc); t_cpu += time.perf_counter() - t0
        finally:
            E.set_engine(old)
        if i % max(1, c["steps"] // 20) == 0 or i == c["steps"] - 1:
            curve.append((i, lg, lc))
            if progress:                       # a long run keeps writing (the GPU box kills silent commands)
                with open(progress, "a") as f:
                    f.write(f"step {i} loss hip {lg:.5f} ref {lc:.5f} t_hip {t_gpu:.1f}s t_ref {t_cpu:.1f}s\n")
    assert E.get_engine() is hip_engine
    n = agree = 0
    corr_g = corr_c = 0
    loss_g = loss_c = 0.0
    for i in range(c["test_batches"]):
        batch = test.batch(i)
        lg, ag, pg, tg = gpu.evaluate(a_gpu, batch, o_gpu, "test")
        old = E.set_engine(oracle)
        try:
            lc, ac, pc, tc = cpu.evaluate(a_cpu, batch, o_cpu, "test")
        finally:
            E.set_engine(old)
        pg, pc, tg = pg.cpu().long(), pc.cpu().long(), tg.cpu()
        n += tg.numel(); agree += int((pg == pc).sum())
        corr_g += int((pg == tg).sum()); corr_c += int((pc == tg).sum())
        loss_g += float(lg); loss_c += float(lc)
    drift = max(float((p.detach().cpu() - q.detach()).abs().max()) for p, q in zip(gpu.parameters(), cpu.parameters()))
    return dict(config={k: v for k, v in c.items()}, meta_steps=c["steps"], test_episodes=c["test_batches"] * c["B"],
                test_predictions=n, acc_hip=corr_g / n, acc_reference_path=corr_c / n,
                acc_difference_pp=100.0 * (corr_g - corr_c) / n, predictions_agreeing=agree / n,
                test_loss_hip=loss_g / c["test_batches"], test_loss_reference_path=loss_c / c["test_batches"],
                max_abs_parameter_difference_after_training=drift, train_loss_curve_step_hip_ref=curve,
                train_seconds_hip=round(t_gpu, 2), train_seconds_reference_path_cpu=round(t_cpu, 2))


@pytest.mark.gpu
def test_trained_accuracy_full_size_short_training():
    """BASELINE.json configs[1]'s own shapes (D = 2048, [256, 64], GloVe token text, meta-batch 32, 160 query rows per episode) in the
    default GPU suite: 300 meta-steps (the full run's loss curve is flat from step ~75 on) and 512 held-out episodes with 5 test-time
    inner steps -- about 40 s of the reference path on the host.  Result kept next to the long run's as
    gpurun_out/accuracy_vs_reference_300.json."""
    dev = torch.device("cuda:0")
    c = dict(FULL, steps=300, test_batches=16)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    r = _run(c, dev, None)
    with open(os.path.join(ROOT, "gpurun_out", "accuracy_vs_reference_300.json"), "w") as f:
        json.dump(r, f, indent=1)
    print(json.dumps({k: v for k, v in r.items() if k != "train_loss_curve_step_hip_ref"}))
    first, last = r["train_loss_curve_step_hip_ref"][0], r["train_loss_curve_step_hip_ref"][-1]
    assert last[1] < first[1] - 0.1 and last[2] < first[2] - 0.1
    assert r["acc_hip"] > 1.5 / c["N"]
    assert abs(r["acc_difference_pp"]) <= 0.2
    assert r["predictions_agreeing"] >= 0.99


@pytest.mark.gpu
def test_trained_accuracy_matches_reference_path():
    dev = torch.device("cuda:0")
    full = os.environ.get("FUMI_ACC_FULL", "0") == "1"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    r = _run(FULL if full else SHORT, dev, os.path.join(ROOT, "gpurun_out", "accuracy_progress.log") if full else None)
    if full:
        with open(os.path.join(ROOT, "gpurun_out", "accuracy_vs_reference.json"), "w") as f:
            json.dump(r, f, indent=1)
    print(json.dumps({k: v for k, v in r.items() if k != "train_loss_curve_step_hip_ref"}))
    first, last = r["train_loss_curve_step_hip_ref"][0], r["train_loss_curve_step_hip_ref"][-1]
    assert last[1] < first[1] - 0.1 and last[2] < first[2] - 0.1            # both sides actually learn the task
    assert r["acc_hip"] > 1.5 / r["config"]["N"]                            # well above chance
    assert abs(r["acc_difference_pp"]) <= 0.2                               # north_star: within +-0.2 pp of the reference
    assert r["predictions_agreeing"] >= 0.99
