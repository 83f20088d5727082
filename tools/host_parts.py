import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if len(sys.argv) > 1 and sys.argv[1] == '--tiny':      # same launches, negligible GPU work: pure host cost
    bench.CFG.update(D=256, B_per_gpu=8, Q=8, L=16)
from types import SimpleNamespace
from fumi_amd.utils import utils as U
from fumi_amd import hip, engine, lazy, flatgrad
from fumi_amd.models import fumi as F
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
c = bench.CFG; Bg = c["B_per_gpu"]
model, table = bench.make_model(dev)
args = SimpleNamespace(device=dev, num_train_adapt_steps=c["T"], num_test_adapt_steps=c["T"], step_size=c["alpha"], first_order=False,
                       optim="adam", lr=3e-5, weight_decay=5e-4, momentum=0.9, batch_size=Bg, num_ways=c["N"])
opt = U.init_optim(args, model)
batches = bench.make_batches(Bg, dev, 1000)
for i in range(20): model.evaluate(args, batches[i % 4], opt, "train")
torch.cuda.synchronize()
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
eng = engine.get_engine()
wrap(eng, "glove_bag_select", "glove call"); wrap(eng, "fumi_step", "fumi_step call"); wrap(opt, "step", "optimizer.step")
wrap(lazy, "scalars", "lazy.scalars"); wrap(F.fdist, "all_reduce_sum_", "all_reduce (noop)")
wrap(hip.lib(), "fumi_hip_fumi_step", "  C: fumi_step"); wrap(hip.lib(), "fumi_hip_adam_step", "  C: adam"); wrap(hip.lib(), "fumi_hip_glove_bag_select", "  C: glove")
n = 300; t0 = time.perf_counter()
for i in range(n): model.evaluate(args, batches[i % 4], opt, "train")
tot = time.perf_counter() - t0
torch.cuda.synchronize()
print("host per step %.1f us" % (tot / n * 1e6))
for k, v in acc.items(): print("  %-22s %6.1f us" % (k, v / n * 1e6))
