import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from types import SimpleNamespace
from fumi_amd.utils import utils as U
from fumi_amd import hip
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
c = bench.CFG; Bg = c["B_per_gpu"]
model, table = bench.make_model(dev)
args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=c["alpha"], first_order=False,
                       optim="adam", lr=3e-5, weight_decay=5e-4, momentum=0.9, batch_size=Bg, num_ways=c["N"])
opt = U.init_optim(args, model)
batches = bench.make_batches(Bg, dev, 1000)
for i in range(20): model.evaluate(args, batches[i % bench.NBATCH], opt, "train")
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for i in range(300):
    a = time.perf_counter()
    model.evaluate(args, batches[i % bench.NBATCH], opt, "train")
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host.sort()
print("per-call host time: median %.1f us, mean %.1f us; enqueue loop %.1f us/step; total %.1f us/step" % (
    host[len(host)//2]*1e6, sum(host)/len(host)*1e6, (t1-t0)/300*1e6, (t2-t0)/300*1e6))
if len(sys.argv) > 1 and sys.argv[1] == "--profile":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(300): model.evaluate(args, batches[i % bench.NBATCH], opt, "train")
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
