"""Runs fumi_hip_am3_step at the configs[3] per-rank shapes in eval / train / train+stats mode (30 calls each) so that a
rocprofv3 --kernel-trace of this script shows what the head kernel costs in each mode (dev tool)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fumi_amd import hip
from oracle import casegen as cg
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
B, N, K, Q, D, Dt, Ht, P = 32, 5, 5, 32, 2048, 768, 256, 64
ep = cg.make_episodes(1, B, N, K, Q, D, Dt)
w = cg.make_am3_params(1, D, Dt, Ht, P)
wl = [w[k].to(dev) for k in hip.AM3_KEYS]
g = lambda t: t.to(dev).contiguous()
args = (ws, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), g(ep["text_s"]), wl, N, None)
stats = torch.zeros(3 + N * N, device=dev)
for ng, st in ((False, None), (True, None), (True, stats)):
    for _ in range(30):
        hip.am3_step(*args, need_grad=ng, stats=st)
    torch.cuda.synchronize()
