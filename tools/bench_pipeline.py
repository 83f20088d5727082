"""Dev tool (GPU box): end-to-end meta-training steps at the bench shapes fed by the GPU-resident episode sampler
(gathered rows vs zero-copy RowRefs), next to pre-generated resident batches (what bench.py times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from types import SimpleNamespace
import bench
from fumi_amd.dataset.gpu_sampler import GpuEpisodeSampler
from fumi_amd.utils import utils as U
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
c = bench.CFG
model, table = bench.make_model(dev)
args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=c["alpha"], first_order=False,
                       optim="adam", lr=3e-5, weight_decay=5e-4, momentum=0.9, batch_size=c["B_per_gpu"], num_ways=c["N"])
opt = U.init_optim(args, model)
n_img, C = 195000, 675
g = torch.Generator(device=dev).manual_seed(0)
images = torch.randn(n_img, c["D"], device=dev, generator=g)
coi = np.random.RandomState(0).randint(0, C, n_img)
tokens = torch.randint(1, c["V"], (C, c["L"]), device=dev, generator=g)
def run(name, get):
    for i in range(20): model.evaluate(args, get(i), opt, "train")
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 300
    for i in range(n): model.evaluate(args, get(100 + i), opt, "train")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} {dt * 1e6:7.1f} us/step  {c['B_per_gpu'] / dt:9.0f} episodes/s")
fixed = bench.make_batches(c["B_per_gpu"], dev, 1000)
run("resident pre-generated batches", lambda i: fixed[i % len(fixed)])
for zc in (False, True):
    smp = GpuEpisodeSampler(images, coi, tokens, c["N"], c["K"], c["Q"], c["B_per_gpu"], seed=1, zero_copy=zc)
    run("sampler, zero-copy rows" if zc else "sampler, gathered rows", smp.batch)
