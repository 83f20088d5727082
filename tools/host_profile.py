"""Host-side cost of a training step: cProfile of `evaluate(task='train')` for one of tools/bench_configs.py's configurations or
bench.py's configs[1] ("fumi_glove").  The GPU runs asynchronously; what is profiled is the Python + ctypes + HIP launch path.
    python tools/host_profile.py [fumi_glove|am3_b32|maml_5w1s_b4_t5|fumi_bert_t5_b32] [steps]"""
import cProfile
import os
import pstats
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

name = sys.argv[1] if len(sys.argv) > 1 else "fumi_glove"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
from fumi_amd.utils import utils as U
if name == "fumi_glove":
    import bench
    c = bench.CFG
    model, table = bench.make_model(dev)
    args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=c["alpha"], first_order=False,
                           optim="adam", lr=3e-5, weight_decay=5e-4, momentum=0.9, batch_size=c["B_per_gpu"], num_ways=c["N"])
    opt = U.init_optim(args, model)
    bs = bench.make_batches(c["B_per_gpu"], dev, 1000)
    step = lambda b: model.evaluate(args, b, opt, "train")
else:
    sys.argv = sys.argv[:1]
    import tools.bench_configs as BC
    from fumi_amd.models import maml as maml_mod
    a = U.parser().parse_args(BC.CONFIGS[name] + ["--dropout", "0", "--dataset", "synthetic"])
    a.device = dev
    torch.manual_seed(1)
    model = U.init_model(a, None, watch=False)
    opt = U.init_optim(a, model)
    opt_, sched = opt if type(opt) == tuple else (opt, None)
    bs = BC.batches(a, dev)
    if a.model == "maml":
        step = lambda b: maml_mod.evaluate(a, model, b, opt_, "train")
    elif a.model == "fumi":
        step = lambda b: model.evaluate(a, b, opt_, "train")
    else:
        step = lambda b: model.evaluate(b, opt_, sched, a.num_ways, dev, "train")
for i in range(20):
    step(bs[i % len(bs)])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(bs[i % len(bs)])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name}: enqueue {(t1 - t0) / steps * 1e6:.1f} us/step, total {(t2 - t0) / steps * 1e6:.1f} us/step", flush=True)
# pure host cost: a few steps enqueued into an EMPTY queue (the host never waits for the device inside such a burst)
burst = []
for r in range(30):
    torch.cuda.synchronize()
    a_ = time.perf_counter()
    for i in range(4):
        step(bs[i % len(bs)])
    burst.append((time.perf_counter() - a_) / 4)
torch.cuda.synchronize()
burst.sort()
print(f"{name}: host cost of a step (4-step bursts into an empty queue): median {burst[len(burst) // 2] * 1e6:.1f} us, "
      f"min {burst[0] * 1e6:.1f} us", flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(steps):
    step(bs[i % len(bs)])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
