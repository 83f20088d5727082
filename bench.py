"""Benchmark of the FuMI meta-training step on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one meta-batch through ``FUMI.evaluate(task='train')``: GloVe embedding bag -> class text select ->
hypernetwork -> T inner SGD steps on the support set -> query loss -> second-order meta-gradients ->
[all-reduce over ranks] -> Adam step (the reference defaults: lr 3e-5, coupled weight decay 5e-4), on synthetic
episodes of the reference's batch layout that are already resident in HBM when the timed region starts.
Workload (configs[1]): 5-way 5-shot, 32 query/class, D=2048 ResNet-152-style embeddings, im_hid_dim [256,64],
GloVe-300 token text (L=128, V=20000, mean pooling), text_hid 256, 1 inner step, 32 episodes per GPU (weak scaling:
the global meta-batch is 32*N episodes sharded as contiguous blocks, one RCCL all-reduce of the flat gradient).

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel (xpanel_bwd: gW0 = sum_b Abar0_b^T [Xs_b;Xq_b], split-bf16 on the
bf16 matrix pipe since round 2; with the forward X-panel pass the longest kernel of a step), timed with
HIP events on the launch stream inside the timed region; ``cpu_baseline`` is the oracle restatement of the reference
path timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(N=5, K=5, Q=32, D=2048, hid=[256, 64], E=300, L=128, V=20000, Ht=256, T=1, B_per_gpu=32, alpha=0.01)
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 MFMA peak (same guide)
SPLIT_PRODUCTS = 6                    # bf16 piece products per fp32 product of the split-bf16 kernels (xpanel.hip)
PROF_EVERY = 8                        # HIP-event timing of the roofline kernel: every 8th step (every steps//16-th of a short run)
NBATCH = 4                            # distinct pre-generated meta-batches cycled through the steps
SETTLE_STEPS = 1000                   # untimed steps after the W warm-up steps (clock ramp), see main()


def make_batches(B, dev, seed0, nbatch=NBATCH):
    """Synthetic meta-batches in the reference loader's layout, generated on the device."""
    c = CFG
    S, Qn = c["N"] * c["K"], c["N"] * c["Q"]
    out = []
    for i in range(nbatch):
        g = torch.Generator(device=dev).manual_seed(seed0 + i)
        cg_ = torch.Generator().manual_seed(seed0 + i)
        y_s = torch.stack([torch.arange(c["N"]).repeat_interleave(c["K"])[torch.randperm(S, generator=cg_)] for _ in range(B)])
        y_q = torch.stack([torch.arange(c["N"]).repeat_interleave(c["Q"])[torch.randperm(Qn, generator=cg_)] for _ in range(B)])
        lens = torch.randint(8, c["L"] + 1, (B, c["N"]), generator=cg_)
        tok = torch.randint(1, c["V"], (B, c["N"], c["L"]), generator=cg_)
        tok = tok * (torch.arange(c["L"])[None, None, :] < lens[..., None])              # PAD (=0) after the length
        text_s = torch.gather(tok, 1, y_s[..., None].expand(-1, -1, c["L"]))             # same text for a class's shots
        text_q = torch.gather(tok, 1, y_q[..., None].expand(-1, -1, c["L"]))
        x_s = torch.randn(B, S, c["D"], device=dev, generator=g)
        x_q = torch.randn(B, Qn, c["D"], device=dev, generator=g)
        idx_s = torch.arange(B * S, device=dev).view(B, S)
        idx_q = torch.arange(B * Qn, device=dev).view(B, Qn)
        out.append({'train': ([idx_s, text_s.to(dev), x_s], y_s.to(dev)),
                    'test': ([idx_q, text_q.to(dev), x_q], y_q.to(dev))})
    return out


def make_model(dev, seed=123):
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.models import common
    c = CFG
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    vecs = (torch.rand(c["V"], c["E"], generator=g) * 2 - 1).numpy()           # stands in for glove-wiki-gigaword-300
    words = [f"w{i}" for i in range(c["V"])]
    common.register_word_vectors("glove", common.ArrayKeyedVectors(words, vecs))
    dictionary = {"PAD": 0}
    dictionary.update({w: i for i, w in enumerate(words) if i > 0})
    model = FUMI(n_way=c["N"], im_emb_dim=c["D"], im_hid_dim=c["hid"], text_encoder="glove", text_emb_dim=c["E"],
                 text_hid_dim=c["Ht"], dropout_rate=0.0, dictionary=dictionary, pooling_strat="mean",
                 norm_hypernet=False)
    return model.to(dev), model.text_encoder.embed.weight.detach().cpu().clone()


def flops_dominant(B):
    """xpanel_bwd: gW0 = sum_b Abar0_b^T [Xs_b;Xq_b] -> 2 (S+Qn) h0 D flops per episode."""
    c = CFG
    S, Qn = c["N"] * c["K"], c["N"] * c["Q"]
    # + the hypernetwork backward that rides in the same launch (hyper_bwd.h): ubar, hp^T u, ubar^T c over the B N class rows
    H1 = c["hid"][-1] + 1
    rider = 2.0 * B * c["N"] * (2 * H1 * c["Ht"] + c["Ht"] * c["E"])
    return 2.0 * B * (S + Qn) * c["hid"][0] * c["D"] + rider


def flops_query(B):
    """query_lds_kernel<true> (the longest launch of the step): per episode one inner step on the S support rows (G_ss D, layer 1
    forward, head, their backward products and the fast-weight updates) + forward, cross-entropy and first-order backward of the Qn
    query rows through the adapted layers (layer 0 in its low-rank form: A0 - alpha G D).  Algorithmic: the inner step is counted
    once per episode although each of the episode's tiles re-runs it."""
    c = CFG
    S, Qn, N = c["N"] * c["K"], c["N"] * c["Q"], c["N"]
    h0, h1 = c["hid"][0], c["hid"][1]
    inner = 2.0 * S * (S * h0 + 3 * h0 * h1 + 3 * h1 * N)                      # G D; layer 1 fwd, dz0, dW1; head fwd, dz1, dWh
    query = 2.0 * Qn * (2 * S * h0 + 3 * h0 * h1 + 3 * h1 * N)                 # G D and its adjoint; layer 1 / head fwd + two backward products
    return B * (inner + query)


def bytes_dominant(B):
    """xpanel_bwd algorithmic HBM bytes: X and Abar0 read once, gW0 written once."""
    c = CFG
    S, Qn, h0 = c["N"] * c["K"], c["N"] * c["Q"], c["hid"][0]
    H1, R = c["hid"][-1] + 1, B * c["N"]
    rider = R * (c["E"] + c["Ht"] + 2 * H1) + H1 * c["Ht"] + (R + 15) // 16 * (c["Ht"] * c["E"] + H1 * c["Ht"])   # rows in, slabs out
    return 4.0 * (B * (S + Qn) * c["D"] + B * (S + Qn) * h0 + h0 * c["D"] + rider)


def flops_step_algorithmic(B):
    """F_ep(T) = S T (4u0 + 9 sum u_i + 9 u_head) + Qn (2u0 + 3 sum u_i + 3 u_head) + hypernet."""
    c = CFG
    S, Qn, N = c["N"] * c["K"], c["N"] * c["Q"], c["N"]
    u0 = 2 * c["D"] * c["hid"][0]
    ui = sum(2 * c["hid"][i - 1] * c["hid"][i] for i in range(1, len(c["hid"])))
    uh = 2 * c["hid"][-1] * N
    hyper = 3 * N * 2 * (c["E"] * c["Ht"] + c["Ht"] * (c["hid"][-1] + 1))
    return B * (S * c["T"] * (4 * u0 + 9 * ui + 9 * uh) + Qn * (2 * u0 + 3 * ui + 3 * uh) + hyper)


def cpu_baseline(table, budget_s=12.0):
    """The oracle restatement of the reference path (oracle/fumi_ref.py, eager PyTorch CPU, per-episode Python loop,
    autograd.grad(create_graph=True)) on this box's host cores: GloVe pooling + meta-step + Adam step per meta-batch."""
    from oracle import fumi_ref as R
    from oracle import casegen as cg
    c = CFG
    B = c["B_per_gpu"]
    theta, phi = cg.make_fumi_params(5, c["D"], c["hid"], c["E"], c["Ht"])
    params = [t.clone().requires_grad_(True) for t in theta + phi]
    opt = torch.optim.Adam(params, lr=3e-5, weight_decay=5e-4)
    batches = make_batches(B, torch.device("cpu"), 900, nbatch=2)

    def step(bt, nb=None):
        (_, tok, x_s), y_s = bt['train']
        (_, _, x_q), y_q = bt['test']
        if nb:                                   # the first nb episodes only (the loop is per episode: same rate per episode)
            tok, x_s, y_s, x_q, y_q = tok[:nb], x_s[:nb], y_s[:nb], x_q[:nb], y_q[:nb]
        text = R.word_embedding_pool(tok, table, 0, "mean")
        out = R.fumi_meta_step(params[:len(theta)], params[len(theta):], text, x_s, y_s, x_q, y_q, c["N"], c["T"],
                               c["alpha"], False)
        for p, g in zip(params, out["g_theta"] + out["g_phi"]):
            p.grad = g
        opt.step()
        return out
    step(batches[0])                                                          # warm-up
    # the eager per-episode loop is dispatch-bound: more threads than ~16 only add contention, so the port
    # gets the thread count that is fastest on this host
    best = (0.0, torch.get_num_threads())
    for nt_ in sorted({1, 8, 16, max(1, (os.cpu_count() or 1) // 2)}):
        if nt_ > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt_)
        step(batches[1])
        t1 = time.perf_counter(); step(batches[0]); r_ = 1.0 / (time.perf_counter() - t1)
        if r_ > best[0]:
            best = (r_, nt_)
    def median_rate(nt_, reps=5, budget=8.0):
        """warm-up, then the median of 5 meta-batches at a fixed thread count -- fewer when the box has
        so many cores that the dispatch-bound loop crawls under thread contention (the sample is bounded in time)."""
        torch.set_num_threads(nt_)
        t1 = time.perf_counter(); step(batches[1], 1); warm = time.perf_counter() - t1     # one episode: how slow is this setting?
        nb = max(1, min(8, int(1.0 / max(warm, 1e-3))))     # ~1 s per sample: bounded even when thread contention is severe
        ts, t_all = [], time.perf_counter()
        for i in range(reps):
            if ts and time.perf_counter() - t_all + ts[-1] > budget:
                break
            t1 = time.perf_counter(); step(batches[i % 2], nb); ts.append(time.perf_counter() - t1)
            if warm > budget:
                break
        if len(ts) < 3:                                     # one or two samples are not a measurement: the line is dropped
            print(f"[bench] cpu baseline, {nt_} threads: {len(ts)} sample(s) within {budget:.0f} s, not reported", file=sys.stderr, flush=True)
            return None
        ts.sort()
        print(f"[bench] cpu baseline, {nt_} threads: {nb / ts[len(ts) // 2]:.1f} episodes/s ({len(ts)} samples)", file=sys.stderr, flush=True)
        return {"value": round(nb / ts[len(ts) // 2], 2), "cores": nt_,
                "how": f"median of {len(ts)} meta-batches of {nb} episodes after 1 warm-up"}
    ncpu = os.cpu_count() or 1
    one_thread, all_cores = median_rate(1), median_rate(ncpu)
    torch.set_num_threads(best[1])
    n, t0 = 0, time.perf_counter()
    while True:
        step(batches[n % 2])
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    extra = {k: v for k, v in (("all_cores", all_cores), ("one_thread", one_thread)) if v is not None}
    return dict(value=round(n * B / el, 2), unit="episodes/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} meta-batches of {B} episodes ({el:.1f} s) of the same workload through oracle/fumi_ref.py "
                       f"(eager PyTorch CPU, {ncpu} logical CPUs visible) at the fastest of the probed thread counts "
                       f"(1, 8, 16, ncpu/2): the eager per-episode loop is dispatch-bound, more threads only add contention",
                **extra)


# ---- BASELINE.json configs[1] AS WORDED: the same FuMI meta-step with the Conv4 encoder on 3 x 84 x 84 images ------------------
CW = dict(N=5, K=5, Q=32, C=3, H=84, E=300, L=128, V=20000, Ht=256, T=1, B_per_gpu=32, alpha=0.01)


def conv_unit_flops():
    """2 * H_l * W_l * 576 * 64 per image for the three 64 -> 64 blocks (42, 21, 10) and 2 * 84^2 * 27 * 64 for block 1."""
    return [2.0 * 84 * 84 * 27 * 64] + [2.0 * h * h * 576 * 64 for h in (42, 21, 10)]


def conv4_flops_per_episode(T, S, Qn):
    """F_img = sum of the four blocks' forward products; support images cost 9 products per inner step
    (forward, 2 backward, 2 + 4 tangent; block 1 has no input gradient: 4), query images 3 (block 1: 2)."""
    u = conv_unit_flops()
    return S * T * (4 * u[0] + 9 * sum(u[1:])) + Qn * (2 * u[0] + 3 * sum(u[1:]))


def as_worded(dev, steps, warmup, cpu_budget_s, with_cpu, T=None):
    from fumi_amd import hip
    from fumi_amd.models import common
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.utils import utils as U
    c = dict(CW, T=T) if T else CW
    B, S, Qn = c["B_per_gpu"], c["N"] * c["K"], c["N"] * c["Q"]
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(7)
    words = [f"w{i}" for i in range(c["V"])]
    common.register_word_vectors("glove", common.ArrayKeyedVectors(words, (torch.rand(c["V"], c["E"], generator=g) * 2 - 1).numpy()))
    dictionary = {"PAD": 0, **{w: i for i, w in enumerate(words) if i > 0}}
    model = FUMI(n_way=c["N"], im_encoder="conv4", image_size=c["H"], image_channels=c["C"], text_encoder="glove",
                 text_emb_dim=c["E"], text_hid_dim=c["Ht"], dropout_rate=0.0, dictionary=dictionary, pooling_strat="mean",
                 norm_hypernet=False).to(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=c["alpha"],
                           first_order=False, optim="adam", lr=3e-5, weight_decay=5e-4, momentum=0.9, batch_size=B, num_ways=c["N"])
    opt = U.init_optim(args, model)

    def batch(seed, device, nb):
        gg = torch.Generator(device=device).manual_seed(seed)
        cg_ = torch.Generator().manual_seed(seed)
        y_s = torch.stack([torch.arange(c["N"]).repeat_interleave(c["K"])[torch.randperm(S, generator=cg_)] for _ in range(nb)])
        y_q = torch.stack([torch.arange(c["N"]).repeat_interleave(c["Q"])[torch.randperm(Qn, generator=cg_)] for _ in range(nb)])
        lens = torch.randint(8, c["L"] + 1, (nb, c["N"]), generator=cg_)
        tok = torch.randint(1, c["V"], (nb, c["N"], c["L"]), generator=cg_) * (torch.arange(c["L"])[None, None, :] < lens[..., None])
        text_s = torch.gather(tok, 1, y_s[..., None].expand(-1, -1, c["L"]))
        text_q = torch.gather(tok, 1, y_q[..., None].expand(-1, -1, c["L"]))
        x_s = torch.randn(nb, S, c["C"], c["H"], c["H"], device=device, generator=gg)
        x_q = torch.randn(nb, Qn, c["C"], c["H"], c["H"], device=device, generator=gg)
        to = lambda t: t.to(device)
        return {'train': ([to(torch.arange(nb * S).view(nb, S)), to(text_s), x_s], to(y_s)),
                'test': ([to(torch.arange(nb * Qn).view(nb, Qn)), to(text_q), x_q], to(y_q))}
    batches = [batch(2000 + i, dev, B) for i in range(2)]
    ws = hip.Workspace.get(dev)
    for i in range(warmup):
        model.evaluate(args, batches[i % 2], opt, "train")
    hip.raise_on_status(ws.read_status())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = model.evaluate(args, batches[i % 2], opt, "train")
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = el / steps * 1e3
    # two more steps with every launch bracketed by HIP events for the roofline object: with phase timing on the library keeps
    # everything on one stream (in the timed steps the two halves of the meta-batch run on two streams)
    psteps = 2
    ws.set_profiling(True, ["conv_gemm", "conv_first", "conv_ew"], every=1)
    for i in range(psteps):
        model.evaluate(args, batches[i % 2], opt, "train")
    torch.cuda.synchronize()
    prof = ws.profile()
    ws.set_profiling(False)
    f_ep = conv4_flops_per_episode(c["T"], S, Qn)
    u = conv_unit_flops()
    gemm_flops = B * sum(u[1:]) * (S * c["T"] * 9 + Qn * 3)
    out = {"workload": "FuMI 5-way 5-shot, 32 query/class, Conv4 (4 x [conv3x3(64) . BN(batch stats) . ReLU . maxpool2]) on 3x84x84 "
                       "images -> 1600 features, hypernetwork head [5,1601] from GloVe-300 token text, 1 inner step over encoder + "
                       "head, second-order meta-gradient + Adam step; BASELINE.json configs[1] as worded (the reference has no "
                       "convolutional encoder: parity unpinned, oracle = oracle/conv4_ref.py)",
           "value": round(B * steps / el, 2), "unit": "episodes/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
           "episodes_per_gpu": B, "dtype": "f32", "data": "synthetic",
           "gflop_per_episode_algorithmic": round(f_ep / 1e9, 2),
           "step_tflops_algorithmic": round(B * f_ep / (ms * 1e-3) / 1e12, 2),
           "final_loss": float(last[0]), "final_acc": float(last[1]), "workspace_GiB": round(ws.bytes() / 2 ** 30, 1)}
    if "conv_gemm" in prof:
        tot, n = prof["conv_gemm"]
        per_step = tot / psteps * 1e-3
        ach = gemm_flops / per_step / 1e12
        # HBM bytes per meta-step of these kernels, from separate rocprofv3 --pmc passes (profiles/<round>/conv4_pmc_traffic.json)
        traffic, pix = None, [S * c["T"] * B, Qn * B]                # images per step: support (per inner step), query
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "conv4_pmc_traffic.json")))[-1:]:
            ks = json.load(open(f))["per_step"]
            traffic = int(sum(v["hbm_bytes"] for k, v in ks.items() if k.startswith("conv64_kernel") or k.startswith("wgrad64_kernel")))
        if c["T"] != 1:
            traffic = None                                           # (the committed passes are of the 1-step form)
        # algorithmic bytes: every map (padded channels-last: (H+2)^2 x 64 floats per image) read or written once per product
        maps = [256.0 * (hh + 2) ** 2 for hh in (42, 21, 10)]
        alg = sum(m * (pix[0] * (2 + 2 + 3 + 3 + 2 + 4) + pix[1] * (2 + 2 + 2)) for m in maps)
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                           "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of tools/bench_conv4.py, "
                                             "bytes per meta-step summed over these kernels' launches (not re-measured in this run)",
                           "algorithmic_bytes": int(alg),
                           "kernel": "conv64_kernel + wgrad64_kernel: the 64 -> 64 channel 3x3 products (forward, input-gradient, "
                                     "weight-gradient and their tangents) as implicit GEMMs on v_mfma_f32_32x32x2_f32",
                           "flops_per_step": gemm_flops, "ms_per_step": round(per_step * 1e3, 3), "launches_per_step": n // psteps,
                           "timed": "HIP events around every launch of two extra steps after the timed region (single stream)"}
        out["phase_ms_per_step"] = {k: round(v[0] / psteps, 3) for k, v in prof.items()}
    if with_cpu:
        from oracle import casegen as cg
        from oracle import conv4_ref as C
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        theta = [p.detach().cpu().clone().requires_grad_(True) for p in model._theta()]
        phi = [p.detach().cpu().clone().requires_grad_(True) for p in model._phi()]
        cb = batch(3000, torch.device("cpu"), 1)
        (_, tok, x_s), y_s = cb['train']
        (_, _, x_q), y_q = cb['test']
        from oracle import fumi_ref as R
        text = R.word_embedding_pool(tok, model.text_encoder.embed.weight.detach().cpu(), 0, "mean")
        n, t0 = 0, time.perf_counter()
        while True:
            C.fumi_conv4_meta_step(theta, phi, text, x_s, y_s, x_q, y_q, c["N"], c["T"], c["alpha"], False)
            n += 1
            el2 = time.perf_counter() - t0
            print(f"[bench] as-worded cpu baseline: {n} episode(s) in {el2:.1f} s", file=sys.stderr, flush=True)
            if el2 >= cpu_budget_s or n >= 8:
                break
        out["cpu_baseline"] = {"value": round(n / el2, 4), "unit": "episodes/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{n} episode(s) of the same shape ({el2:.1f} s) through oracle/conv4_ref.py (eager PyTorch "
                                         f"CPU, autograd second order)"}
    return out


# ---- BASELINE.json configs[4]: FuMI 20-way 5-shot, ResNet-12 backbone in bf16, 5 inner steps, second-order outer gradients -----------
C4 = dict(N=20, K=5, Q=15, C=3, H=84, E=768, Ht=256, T=5, B_per_gpu=64, alpha=0.01, channels=(64, 160, 320, 640))


def resnet12_layer_flops():
    """[(forward flops per image, has an input gradient)] of the 16 convolutions (3 x 3x3 + the 1x1 shortcut per block)."""
    out, h, ci = [], C4["H"], C4["C"]
    for l, c in enumerate(C4["channels"]):
        for (cin, k) in ((ci, 9), (c, 9), (c, 9), (ci, 1)):
            out.append((2.0 * h * h * k * cin * c, not (l == 0 and cin == ci)))
        h //= 2; ci = c
    return out


def resnet12_flops_per_episode(T, S, Qn):
    """Products per support image and inner step: forward 1, backward 2, tangent forward 2, tangent backward 4 = 9 (SURVEY.md 8d's
    c_s); the two convolutions that read the image have no input gradient and no x' term: 4.  Query images: 3 (2)."""
    f = resnet12_layer_flops()
    full = sum(x for x, g in f if g); first = sum(x for x, g in f if not g)
    return S * T * (9 * full + 4 * first) + Qn * (3 * full + 2 * first)


def configs4_leg(dev, steps, warmup, with_cpu, episodes):
    """One GPU's share of BASELINE.json configs[4] (512 / 8 = 64 episodes) through FUMI.evaluate(task='train') with
    im_encoder='resnet12': class text rows (BERT width) -> hypernetwork -> [20, 641] heads; 5 inner steps over the 48 encoder
    tensors + head on 100 support images; 300 query images; second-order meta-gradient; Adam step.  Episodes are processed in
    chunks that fit the workspace budget (csrc/rn12.hip)."""
    from fumi_amd import hip
    from fumi_amd.models.fumi import FUMI
    from fumi_amd.utils import utils as U
    c = C4
    B, S, Qn = episodes, c["N"] * c["K"], c["N"] * c["Q"]
    torch.manual_seed(11)
    model = FUMI(n_way=c["N"], im_encoder="resnet12", image_size=c["H"], image_channels=c["C"], text_encoder="BERT",
                 text_emb_dim=c["E"], text_hid_dim=c["Ht"], dropout_rate=0.0, norm_hypernet=False).to(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=c["alpha"],
                           first_order=False, optim="adam", lr=3e-5, weight_decay=5e-4, momentum=0.9, batch_size=B, num_ways=c["N"])
    opt = U.init_optim(args, model)

    def batch(seed, device, nb):
        gg = torch.Generator(device=device).manual_seed(seed)
        cg_ = torch.Generator().manual_seed(seed)
        y_s = torch.stack([torch.arange(c["N"]).repeat_interleave(c["K"])[torch.randperm(S, generator=cg_)] for _ in range(nb)])
        y_q = torch.stack([torch.arange(c["N"]).repeat_interleave(c["Q"])[torch.randperm(Qn, generator=cg_)] for _ in range(nb)])
        cls = torch.randn(nb, c["N"], c["E"], generator=cg_)
        text_s = torch.gather(cls, 1, y_s[..., None].expand(-1, -1, c["E"]))
        text_q = torch.gather(cls, 1, y_q[..., None].expand(-1, -1, c["E"]))
        x_s = torch.randn(nb, S, c["C"], c["H"], c["H"], device=device, generator=gg)
        x_q = torch.randn(nb, Qn, c["C"], c["H"], c["H"], device=device, generator=gg)
        to = lambda t: t.to(device)
        return {'train': ([to(torch.arange(nb * S).view(nb, S)), to(text_s), x_s], to(y_s)),
                'test': ([to(torch.arange(nb * Qn).view(nb, Qn)), to(text_q), x_q], to(y_q))}
    bt = batch(4000, dev, B)
    ws = hip.Workspace.get(dev)
    for _ in range(warmup):
        model.evaluate(args, bt, opt, "train")
    hip.raise_on_status(ws.read_status())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = model.evaluate(args, bt, opt, "train")
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = el / steps * 1e3
    # one more step with every launch bracketed by HIP events for the roofline objects: with phase timing on the library keeps
    # everything on one stream (in the timed steps above the weight gradients run on a second stream beside the backward chain)
    ws.set_profiling(True, ["rn_conv", "rn_wgrad", "rn_ew"], every=1)
    model.evaluate(args, bt, opt, "train")
    torch.cuda.synchronize()
    prof = ws.profile()
    ws.set_profiling(False)
    psteps = 1
    f_ep = resnet12_flops_per_episode(c["T"], S, Qn)
    lf = resnet12_layer_flops()
    full = sum(x for x, g in lf if g); first = sum(x for x, g in lf if not g)
    # the forward / input-gradient / tangent convolutions (rn_conv_kernel): 6 of a support step's 9 products, 2 of a query image's 3
    conv_flops = B * (S * c["T"] * (6 * full + 2 * first) + Qn * (2 * full + first))
    wgrad_flops = B * f_ep - conv_flops
    out = {"workload": "FuMI 20-way 5-shot, 15 query/class, ResNet-12 (channels 64/160/320/640, 3 x conv3x3 . BN(batch stats) . "
                       "LeakyReLU(0.1) + conv1x1 shortcut, maxpool2; global average pool -> 640) on 3x84x84 images in bf16 (fp32 "
                       "accumulation, fp32 master weights), hypernetwork head [20,641] from 768-d class text rows, 5 inner steps "
                       "over encoder + head, second-order meta-gradient + Adam step; BASELINE.json configs[4], one GPU's share "
                       "(512 / 8 episodes) -- the reference has no ResNet-12: parity unpinned, oracle = oracle/resnet12_manual.py",
           "value": round(B * steps / el, 3), "unit": "episodes/s", "ms_per_step": round(ms, 1), "steps": steps, "warmup": warmup,
           "episodes_per_gpu": B, "dtype": "bf16", "data": "synthetic",
           "tflop_per_episode_algorithmic": round(f_ep / 1e12, 3),
           "step_tflops_algorithmic": round(B * f_ep / (ms * 1e-3) / 1e12, 1),
           "final_loss": float(last[0]), "final_acc": float(last[1]), "workspace_GiB": round(ws.bytes() / 2 ** 30, 1)}
    if "rn_conv" in prof:
        tot, n = prof["rn_conv"]
        per_step = tot / psteps * 1e-3
        ach = conv_flops / per_step / 1e12
        # algorithmic HBM bytes of those launches: every source map read once and the output written once per product, bf16
        h, ci, alg = c["H"], 16, 0.0
        for l, ch in enumerate(c["channels"]):
            px = (h + 2) ** 2 * 2.0
            per_prod = [px * (ci + ch), px * 2 * ch, px * 2 * ch, px * (ci + ch)]          # c1, c2, c3, shortcut: input + output maps
            nprod_s = [3 if l == 0 else 6, 6, 6, 3 if l == 0 else 6]                      # (tangent products read two sources: counted as products)
            nprod_q = [1 if l == 0 else 2, 2, 2, 1 if l == 0 else 2]
            alg += B * sum(pp * (S * c["T"] * ns + Qn * nq) for pp, ns, nq in zip(per_prod, nprod_s, nprod_q))
            h //= 2; ci = ch
        # HBM bytes of the conv launches from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE over `tools/bench_resnet12.py 8 1 5
        # 15` = two 8-episode steps), scaled to this leg's episodes; not re-measured in this run
        traffic, tsrc = None, None
        import glob
        for f_ in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "rn12_pmc_traffic.json")))[-1:]:
            pk = json.load(open(f_))
            tot_b = sum(e.get("hbm_bytes_all_launches", 0.0) for k_, e in pk["kernels"].items() if k_.startswith("rn_conv_kernel"))
            if tot_b > 0:
                traffic = int(tot_b / pk.get("steps", 2) * (B / pk.get("episodes", 8)))
                tsrc = os.path.relpath(f_, ROOT)
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": tsrc,
                           "algorithmic_bytes": int(alg),
                           "kernel": "rn_conv_kernel: forward / input-gradient convolutions and their tangent forms as implicit GEMMs "
                                     "over shifted pixel slabs on v_mfma_f32_16x16x32_bf16 (32x32x16 for the two 3-channel image layers; csrc/rn12_conv.hip)",
                           "flops_per_step": conv_flops, "ms_per_step": round(per_step * 1e3, 2), "launches_per_step": n // psteps,
                           "timed": "HIP events around every launch of one extra step after the timed region (single stream)"}
        if "rn_wgrad" in prof:
            wt = prof["rn_wgrad"][0] / psteps * 1e-3
            out["roofline_wgrad"] = {"bound": "mfma", "achieved": round(wgrad_flops / wt / 1e12, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                     "unit": "TFLOP/s", "frac": round(wgrad_flops / wt / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                                     "kernel": "rn_wgrad_kernel (+ the reduction of its pixel-slab partial sums)",
                                     "ms_per_step": round(wt * 1e3, 2)}
        out["phase_ms_per_step"] = {k: round(v[0] / psteps, 2) for k, v in prof.items()}
    if with_cpu:
        # bounded sample of the same workload on the host: one forward + first-order backward of one episode's 100 support images
        # through oracle/resnet12_ref.py (autograd, fp32) = 300 of the episode's 5400 image-passes (S T 9 + Qn 3); extrapolated
        from oracle import resnet12_ref as RR
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        theta = [p.detach().cpu().clone().requires_grad_(True) for p in model._theta()]
        cb = batch(5000, torch.device("cpu"), 1)
        x, y = cb['train'][0][2][0], cb['train'][1][0]
        h0 = (torch.randn(c["N"], 641) * 0.05).requires_grad_(True)
        t1 = time.perf_counter()
        loss = torch.nn.functional.cross_entropy(RR.forward(x, theta, h0), y)
        torch.autograd.grad(loss, theta + [h0])
        el2 = time.perf_counter() - t1
        units = S * c["T"] * 9 + Qn * 3
        out["cpu_baseline"] = {"value": round(1.0 / (el2 * units / (3.0 * S)), 5), "unit": "episodes/s", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": f"one forward + first-order backward of one episode's {S} support images through "
                                         f"oracle/resnet12_ref.py (eager PyTorch CPU, fp32 autograd): {el2:.1f} s for {3 * S} of the "
                                         f"episode's {units} image-passes (S T 9 + Qn 3), extrapolated linearly"}
    return out


def visible_gpus():
    """GPUs of this node counted from the KFD topology in sysfs (a node with simd_count > 0 and a non-zero gfx target is a GPU),
    narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set; None when sysfs cannot tell."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0 and int(props.get("gfx_target_version", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start one fresh process per GPU through torch.distributed.run as a CHILD of
    this process (which has not touched the GPU and never will), let rank 0's JSON line through and hand back the child's exit
    code.  (Replacing this process with the launcher after a HIP call is what the pool forbids; a child process is not that.)"""
    import socket
    import subprocess
    if os.environ.get("FUMI_BENCH_REHEARSAL", "0") != "1":
        have = visible_gpus()                       # from sysfs: the parent makes no HIP / torch.cuda call at all
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {have} "
                  f"(FUMI_BENCH_REHEARSAL=1 rehearses the N-rank code path on one GPU over gloo)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-as-worded", action="store_true", help="skip the Conv4-as-worded leg (runs at N = 1 only)")
    ap.add_argument("--as-worded-steps", type=int, default=10)
    ap.add_argument("--no-configs4", action="store_true", help="skip the ResNet-12 / bf16 leg (BASELINE.json configs[4], N = 1 only)")
    ap.add_argument("--configs4-episodes", type=int, default=C4["B_per_gpu"], help="episodes of the ResNet-12 leg (one GPU's share: 64)")
    ap.add_argument("--configs4-steps", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (EXACTLY --steps steps between two barriers) is repeated this many times back to back; "
                         "`value` / `ms_per_step` are the MEDIAN repeat's, min / max are reported beside it")
    ap.add_argument("--no-extra", action="store_true", help="skip the per-rank lines of BASELINE.json configs[0], [2], [3] (N = 1 only)")
    ap.add_argument("--no-phase-timing", action="store_true", help="do not record HIP events around the library's phases")
    ap.add_argument("--all-phases", action="store_true",
                    help="time every phase of the library (adds ~10 us of stream time per phase and step); default: only the "
                         "roofline kernel's phase")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a.gpus))             # (nothing has touched the GPU in this process)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    # rehearsal on a one-GPU box (the N > 1 code path, not a measurement): FUMI_BENCH_REHEARSAL=1 puts every rank on cuda:0
    # and lets gloo carry the collectives (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("FUMI_BENCH_REHEARSAL", "0") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from fumi_amd import hip
    from fumi_amd.utils import utils as U
    c = CFG
    Bg = c["B_per_gpu"] * world
    model, table = make_model(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=c["alpha"],
                           first_order=False, optim="adam", lr=3e-5, weight_decay=5e-4, momentum=0.9,
                           batch_size=Bg, num_ways=c["N"])
    opt = U.init_optim(args, model)
    batches = make_batches(Bg, dev, 1000)                                     # same global meta-batches on every rank
    ws = hip.Workspace.get(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        model.evaluate(args, batches[i % NBATCH], opt, "train")
    hip.raise_on_status(ws.read_status())
    # W warm-up steps are a few milliseconds of this workload: too short for the clocks of a GPU that sat idle while the process
    # started (one run in a dozen timed 0.30 ms per step instead of 0.23 behind 20 warm-up steps, every kernel at its usual
    # duration).  A fixed number of further untimed steps (~0.25 s; the same count on every rank: each step is a collective) runs
    # before the barrier that opens the timed region.
    settle = SETTLE_STEPS
    for i in range(settle):
        model.evaluate(args, batches[i % NBATCH], opt, "train")
        if i % 64 == 63:
            torch.cuda.synchronize()
    if not a.no_phase_timing:
        # an event record is a ~6 us bubble on the stream: the roofline kernel is timed at every 8th step of the timed region
        prof_every = 1 if a.all_phases else max(1, min(PROF_EVERY, a.steps // 16))     # >= 16 samples from a short run too
        ws.set_profiling(True, None if a.all_phases else ["xpanel_bwd", "query"], every=prof_every)
    # the timed region: EXACTLY a.steps steps bracketed by barrier + synchronize on both sides -- repeated a.repeats times back to
    # back (a short region on a freshly started process sees clock and queue jitter of several per cent: the median repeat is the
    # measurement, min and max say how far the repeats spread)
    reps = max(1, a.repeats)
    times = []
    last = None
    for r_ in range(reps):
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            last = model.evaluate(args, batches[i % NBATCH], opt, "train")
        barrier()
        times.append(time.perf_counter() - t0)
    prof = ws.profile() if not a.no_phase_timing else {}
    ws.set_profiling(False)
    t = torch.tensor(times, device="cpu" if rehearsal else dev, dtype=torch.float64)
    allreduce = None
    ranks_info = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                              # every repeat: the slowest rank's time
        # who took part: (rank, device index, episodes of the global meta-batch it owns), gathered so that the line proves N ranks
        from fumi_amd import dist as fdist
        lo, hi = fdist.shard(Bg)
        mine = torch.tensor([rank, dev.index if dev.index is not None else 0, hi - lo], device="cpu" if rehearsal else dev, dtype=torch.int64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        ranks_info = [{"rank": int(g_[0]), "device": int(g_[1]), "episodes": int(g_[2])} for g_ in gathered]
        # the step's one collective, timed on its own after the timed region: the flat [grads | loss | acc] buffer
        flat = model._flat_grads().flat
        buf = flat.clone() if not rehearsal else flat.cpu()
        for _ in range(3):
            dist.all_reduce(buf)
        barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf)
        barrier()
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if not rehearsal else None
        except Exception:                                                     # (version query is informational only)
            rccl = None
        allreduce = {"bytes": int(flat.numel() * 4), "avg_us": round((time.perf_counter() - t1) / 20 * 1e6, 1),
                     "backend": "gloo (rehearsal)" if rehearsal else "nccl (RCCL)", "torch_backend": dist.get_backend(),
                     "rccl_version": rccl, "per_step": 1, "overlapped": False,
                     "why_not_overlapped": "the gradient is final only after the step's last reduction (gW0's split-K slabs and the "
                                           "hypernetwork slabs leave the last matrix launch); what is ready earlier is 70 KB of the "
                                           "3 MB, and a 3 MB ring all-reduce is latency-bound (DESIGN.md section 6)"}
    times = sorted(float(x) for x in t.tolist())
    el = times[len(times) // 2]                                               # the median repeat
    # what the HOST costs per step (Python + ctypes + launches), measured where it cannot be confused with waiting for the device:
    # bursts of 4 steps enqueued into an EMPTY queue, median of 15
    host_ms = None
    if world == 1:
        burst = []
        for r_ in range(15):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for i in range(4):
                model.evaluate(args, batches[i % NBATCH], opt, "train")
            burst.append((time.perf_counter() - tb) / 4)
        torch.cuda.synchronize()
        burst.sort()
        host_ms = round(burst[len(burst) // 2] * 1e3, 4)

    if rank == 0:
        print(f"[bench] timed region done: {el / a.steps * 1e3:.4f} ms/step", file=sys.stderr, flush=True)
        ms = el / a.steps * 1e3
        out = {
            "metric": "episodes/sec (5-way 5-shot FuMI)", "value": round(Bg * a.steps / el, 2), "unit": "episodes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "settle_steps_untimed": settle, "ms_per_step": round(ms, 4),
            "repeats": {"n": len(times), "of_steps": a.steps, "reported": "median",
                        "ms_per_step_min": round(times[0] / a.steps * 1e3, 4), "ms_per_step_median": round(ms, 4),
                        "ms_per_step_max": round(times[-1] / a.steps * 1e3, 4)},
            "host_ms_per_step": host_ms,
            "world_size": world, "ranks": ranks_info if ranks_info else [{"rank": 0, "device": dev.index or 0, "episodes": Bg}],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo, not a measurement)",
            "config": {"workload": "FuMI 5-way 5-shot, 32 query/class, ResNet-152-style 2048-d embeddings, im_hid [256,64], "
                                   "GloVe-300 token text (L=128, V=20000, mean pool), text_hid 256, 1 inner step, "
                                   "second-order meta-gradient + Adam step; BASELINE.json configs[1]",
                       "episodes_per_gpu": c["B_per_gpu"], "global_meta_batch": Bg,
                       "layer0_fwd": "split-bf16x3 operands on the bf16 MFMA with fp32 accumulation (fp32-equivalent: error vs "
                                     "fp64 at the fp32-MFMA kernel's level, tests/test_hip_parity.py::test_xpanel_fwd_presplit_column_operand_has_fp32_accuracy"
                                     " and ::test_xpanel_fwd_split_bf16_has_fp32_accuracy); W0 and the support rows are split once per step",
                       "layer0_bwd": "the same split for gW0 (test_xpanel_bwd_split_bf16_has_fp32_accuracy)",
                       "parallelism": f"episode-sharded x{world}, 1 all-reduce of the flat gradient"},
            "final_loss": float(last[0]), "final_acc": float(last[1]),
            "step_tflops_algorithmic": round(flops_step_algorithmic(c["B_per_gpu"]) / (ms * 1e-3) / 1e12, 3),
        }
        if "xpanel_bwd" in prof:
            tot, n = prof["xpanel_bwd"]
            dur = tot / n * 1e-3
            ach = flops_dominant(c["B_per_gpu"]) / dur / 1e12
            traffic = None       # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/<round>/pmc_traffic.json)
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))[-1:]:
                k = json.load(open(f))["kernels"]
                key = [x for x in k if x.startswith("xpanel_bwd256")] or [x for x in k if x.startswith("xpanel_bwd")]
                if key and "hbm_bytes_per_launch" in k[key[0]]:
                    traffic = int(k[key[0]]["hbm_bytes_per_launch"])
            # the kernel computes the fp32 product from six bf16 piece products on the bf16 matrix pipe: its roofline is that pipe's
            # dense peak divided by six (fp32-equivalent); against the fp32-MFMA peak (157.3) the same rate is `vs_fp32_mfma_peak`
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS if os.environ.get("FUMI_XPB_SB", "1") != "0" else PEAK_F32_MFMA_TFLOPS
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic,
                               "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of this command, "
                                                 "committed as profiles/<round>/pmc_traffic.json (not re-measured in this run)",
                               "algorithmic_bytes": int(bytes_dominant(c["B_per_gpu"])),
                               "vs_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                               "executed_bf16_tflops": round(ach * SPLIT_PRODUCTS, 1),
                               "kernel": "xpanel_bwd256_sb_kernel (gW0 = sum_b Abar0_b^T [Xs_b;Xq_b]: 256 x 2048 outputs in 256 x 128 tiles, contraction over 32 x 185 rows "
                                         "in 16 slabs; fp32 operands split exactly into three bf16 pieces, six piece products per fp32 product on v_mfma_f32_32x32x16_bf16 with "
                                         "fp32 accumulation -- `achieved` counts the ALGORITHMIC fp32 flops, `peak` is the dense bf16 MFMA peak / 6; "
                                         "the launch also carries the hypernetwork backward as 40 rider workgroups, hyper_bwd.h: its 0.09 GFLOP are counted, its ~5 us stretch the launch)",
                               "avg_us": round(dur * 1e6, 2), "launches": n,
                               "timed": f"HIP events around every {prof_every}th launch of the timed region"}
            out["phase_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items()}
        if "query" in prof and c["T"] == 1:
            # the LONGEST launch of the step is not the kernel with the most flops: the per-episode chain (inner step + query tiles)
            tot, n = prof["query"]
            dur = tot / n * 1e-3
            ach = flops_query(c["B_per_gpu"]) / dur / 1e12
            out["roofline_longest"] = {"bound": "latency (a chain of ~25 dependent LDS-resident products per workgroup; priced against the fp32 MFMA peak)",
                                       "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                                       "kernel": "query_lds_kernel<true> (inner step fused into the query tiles: 5 workgroups per episode, "
                                                 "160 on 256 CUs; v_mfma_f32_16x16x4_f32 from LDS)",
                                       "avg_us": round(dur * 1e6, 2), "launches": n,
                                       "timed": f"HIP events around every {prof_every}th launch of the timed region"}
        if allreduce:
            out["allreduce"] = allreduce
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(table)
        if not a.no_extra and world == 1:
            # BASELINE.json's other embedding-input configurations at their per-rank shapes (one GPU's share), with the library's
            # phases timed in a second loop: configs[0]'s shapes on the GPU, configs[2] (FuMI, BERT text, 5 inner steps), configs[3] (AM3)
            print("[bench] extra legs: configs[0], [2], [3] at per-rank shapes", file=sys.stderr, flush=True)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs
            del batches, model, opt
            torch.cuda.empty_cache()
            out["extra"] = {k: bench_configs.run_config(k, dev, steps=100, warmup=10, roofline=True)
                            for k in ("maml_5w1s_b4_t5", "fumi_bert_t5_b32", "am3_b32")}
            batches = None
        if not a.no_as_worded and world == 1:
            print("[bench] as-worded (Conv4, 84x84) leg", file=sys.stderr, flush=True)
            del batches
            torch.cuda.empty_cache()
            out["as_worded"] = as_worded(dev, a.as_worded_steps, 2, 20.0, not a.no_cpu_baseline)
            # configs[2]'s inner-loop depth (the reference's default: 5 steps) on the same images -- the as-worded form of configs[2]
            # up to its text encoder (GloVe token text here, BERT rows there: the hypernetwork's input width, not the encoder's work)
            out["as_worded_t5"] = as_worded(dev, max(2, a.as_worded_steps // 3), 1, 0.0, False, T=5)
            out["as_worded_t5"]["workload"] = out["as_worded_t5"]["workload"].replace("1 inner step", "5 inner steps").replace(
                "BASELINE.json configs[1] as worded", "BASELINE.json configs[2] as worded, GloVe instead of BERT text rows")
        if not a.no_configs4 and world == 1:
            print("[bench] configs[4] (ResNet-12, bf16, 20-way, 5 inner steps) leg", file=sys.stderr, flush=True)
            torch.cuda.empty_cache()
            out["configs4"] = configs4_leg(dev, a.configs4_steps, 1, not a.no_cpu_baseline, a.configs4_episodes)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
