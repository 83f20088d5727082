"""Per-GPU throughput of BASELINE.json's OTHER configurations at their per-rank shapes (informational; bench.py measures
configs[1]).  One JSON line per configuration: model, shapes, ms per meta-step, episodes/s on one MI355X.

    python tools/bench_configs.py [--steps 100] [--warmup 10]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fumi_amd import hip
from fumi_amd.models import am3 as am3_mod, maml as maml_mod
from fumi_amd.utils import utils as U

CONFIGS = {
    # configs[0] shapes on the GPU (the reference runs it on the CPU): MAML 5-way 1-shot, meta-batch 4, 5 inner steps
    "maml_5w1s_b4_t5": ["--model", "maml", "--num_shots", "1", "--batch_size", "4", "--num_train_adapt_steps", "5"],
    # configs[2] per rank: FuMI 5-way 5-shot, BERT description embeddings (768-d), 5 inner steps, 32 of the 256 episodes
    "fumi_bert_t5_b32": ["--model", "fumi", "--text_encoder", "BERT", "--text_emb_dim", "768", "--batch_size", "32",
                         "--num_train_adapt_steps", "5"],
    # configs[3] per rank: AM3 5-way 5-shot, 32 of the 256 episodes
    "am3_b32": ["--model", "am3", "--text_encoder", "BERT", "--text_emb_dim", "768", "--batch_size", "32"],
    # configs[1] with the other text path, for comparison with bench.py's GloVe line
    "fumi_bert_t1_b32": ["--model", "fumi", "--text_encoder", "BERT", "--text_emb_dim", "768", "--batch_size", "32",
                         "--num_train_adapt_steps", "1"],
    # half of configs[1]'s per-rank meta-batch (dev: what two concurrent half-batches could give)
    "fumi_bert_t1_b16": ["--model", "fumi", "--text_encoder", "BERT", "--text_emb_dim", "768", "--batch_size", "16",
                         "--num_train_adapt_steps", "1"],
}


def batches(a, dev, n=4):
    B, N, K, Q, D, Dt = a.batch_size, a.num_ways, a.num_shots, a.num_shots_test, a.im_emb_dim, a.text_emb_dim
    S, Qn = N * K, N * Q
    out = []
    for i in range(n):
        g = torch.Generator(device=dev).manual_seed(100 + i)
        cg = torch.Generator().manual_seed(100 + i)
        y_s = torch.stack([torch.arange(N).repeat_interleave(K)[torch.randperm(S, generator=cg)] for _ in range(B)]).to(dev)
        y_q = torch.stack([torch.arange(N).repeat_interleave(Q)[torch.randperm(Qn, generator=cg)] for _ in range(B)]).to(dev)
        ct = torch.randn(B, N, Dt, device=dev, generator=g)
        text_s = torch.gather(ct, 1, y_s[..., None].expand(-1, -1, Dt))
        text_q = torch.gather(ct, 1, y_q[..., None].expand(-1, -1, Dt))
        x_s = torch.randn(B, S, D, device=dev, generator=g)
        x_q = torch.randn(B, Qn, D, device=dev, generator=g)
        out.append({'train': ([torch.zeros(B, S, dtype=torch.int64, device=dev), text_s, x_s], y_s),
                    'test': ([torch.zeros(B, Qn, dtype=torch.int64, device=dev), text_q, x_q], y_q)})
    return out


def run_config(name, dev, steps=100, warmup=10, roofline=False):
    """One configuration at its per-rank shape on `dev`: the record tools/bench_configs.py prints (bench.py's `extra` legs call this)."""
    argv = CONFIGS[name]
    a = U.parser().parse_args(argv + ["--dropout", "0", "--dataset", "synthetic"])
    a.device = dev
    torch.manual_seed(1)
    model = U.init_model(a, None, watch=False)
    opt = U.init_optim(a, model)
    bs = batches(a, dev)
    opt_, sched = opt if type(opt) == tuple else (opt, None)

    def step(b):
        if a.model == "maml":
            return maml_mod.evaluate(a, model, b, opt_, "train")
        if a.model == "fumi":
            return model.evaluate(a, b, opt_, "train")
        return model.evaluate(b, opt_, sched, a.num_ways, dev, "train")
    for i in range(warmup):
        step(bs[i % len(bs)])
    ws = hip.Workspace.get(dev)
    hip.raise_on_status(ws.read_status())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        last = step(bs[i % len(bs)])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # What the HOST costs per step: bursts of 4 steps enqueued into an EMPTY queue (the host never waits for the device inside such
    # a burst), median of 15.  The time the steady loop above spends enqueueing (`host_enqueue_in_loop_ms`) is NOT that: once the
    # lazily read statistics ring is 16 steps ahead the host blocks on the oldest publication, i.e. it runs at the device's pace
    # whatever its own cost -- the figure reads 0.84 x ms_per_step on a device-bound step.
    burst = []
    for r_ in range(15):
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(4):
            step(bs[i % len(bs)])
        burst.append((time.perf_counter() - tb) / 4)
    torch.cuda.synchronize()
    burst.sort()
    rec = {"config": name, "argv": " ".join(argv), "episodes_per_meta_batch": a.batch_size,
           "ms_per_step": round(el / steps * 1e3, 4), "episodes_per_s": round(a.batch_size * steps / el, 1),
           "host_ms_per_step": round(burst[len(burst) // 2] * 1e3, 4),
           "host_ms_per_step_how": "median of 15 bursts of 4 steps enqueued into an empty queue (pure Python + ctypes + launch cost)",
           "host_enqueue_in_loop_ms": round(t_enq / steps * 1e3, 4), "final_loss": float(last[0])}
    if roofline:
        ws.set_profiling(True, None, every=1)                 # every phase bracketed (adds event bubbles: separate loop)
        for i in range(steps):
            step(bs[i % len(bs)])
        prof = ws.profile()
        ws.set_profiling(False)
        rec["phase_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items()}
        if a.model == "am3" and "xpanel_fwd" in prof:
            B, R, D, P = a.batch_size, a.num_ways * (a.num_shots + a.num_shots_test), a.im_emb_dim, a.prototype_dim
            byt = 4.0 * (B * R * D + P * D + B * R * P)         # rows read once, encoder weight once, embeddings written once
            dur = prof["xpanel_fwd"][0] / prof["xpanel_fwd"][1] * 1e-3
            rec["roofline"] = {"bound": "hbm", "achieved": round(byt / dur / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(byt / dur / 8e12, 4), "traffic": None, "algorithmic_bytes": int(byt),
                               "kernel": "xpanel_fwd (AM3 image encoder: [B*(S+Qn), 2048] x [2048, 64] in one pass over the "
                                         "episode panels; 4 FLOP/B: HBM-bound)", "avg_us": round(dur * 1e6, 2),
                               "launches": prof["xpanel_fwd"][1], "timed": "HIP events around every launch"}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--only", type=str, default=None, help="run one configuration")
    ap.add_argument("--roofline", action="store_true",
                    help="time every phase of the library with HIP events and add a roofline object for the configuration's "
                         "dominant kernel (AM3: the image encoder pass over the 2048-wide rows, HBM-bound)")
    o = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for name in CONFIGS:
        if o.only and name != o.only:
            continue
        print(json.dumps(run_config(name, dev, o.steps, o.warmup, o.roofline)), flush=True)


if __name__ == "__main__":
    main()
