"""Times the Conv4-as-worded FuMI meta-step (BASELINE.json configs[1] wording: 5-way 5-shot, 3x84x84 images, Conv4, 1 inner
step) straight through the C ABI on synthetic images resident in HBM.  python tools/bench_conv4.py [B] [steps] [T] [Q]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fumi_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 32
N, K, Cin, H, W, nblk, Dt, Ht = 5, 5, 3, 84, 84, 4, 300, 256
dev = torch.device("cuda:0")
ws = hip.Workspace.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
S, Qn = N * K, N * Q
x_s = torch.randn(B, S, Cin, H, W, device=dev, generator=g)
x_q = torch.randn(B, Qn, Cin, H, W, device=dev, generator=g)
y_s = torch.arange(N, device=dev).repeat_interleave(K).repeat(B, 1)
y_q = torch.arange(N, device=dev).repeat_interleave(Q).repeat(B, 1)
cls_text = torch.randn(B, N, Dt, device=dev, generator=g)
F = hip.conv4_feature_dim(nblk, H, W)
theta = []
for l in range(nblk):
    ci = Cin if l == 0 else 64
    theta += [(torch.rand(64, ci, 3, 3, device=dev, generator=g) * 2 - 1) / (ci * 9) ** 0.5, torch.ones(64, device=dev), torch.zeros(64, device=dev)]
phi = [(torch.rand(Ht, Dt, device=dev, generator=g) * 2 - 1) / Dt ** 0.5, torch.zeros(Ht, device=dev),
       (torch.rand(F + 1, Ht, device=dev, generator=g) * 2 - 1) / Ht ** 0.5, torch.zeros(F + 1, device=dev)]
g_theta = [torch.empty_like(t) for t in theta]
g_phi = [torch.empty_like(t) for t in phi]


def step():
    return hip.fumi_conv4_step(ws, N, x_s, y_s, x_q, y_q, theta, phi, T, 0.01, False, cls_text=cls_text, g_theta=g_theta, g_phi=g_phi)


t0 = time.perf_counter(); out = step(); torch.cuda.synchronize()
print(f"first call {time.perf_counter() - t0:.2f} s, workspace {ws.bytes() / 2**30:.1f} GiB, loss {float(out['loss_b'].mean()):.4f}", flush=True)
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
fimg = 2 * (84 * 84 * 27 * 64 + 42 * 42 * 576 * 64 + 21 * 21 * 576 * 64 + 10 * 10 * 576 * 64)
flops = B * fimg * (S * T * 9 + Qn * 3)
print(f"B={B} T={T} Q={Q}: {ms:.2f} ms/step, {B / ms * 1e3:.1f} episodes/s, {flops / ms / 1e9:.1f} TFLOP/s (SURVEY 8d count)", flush=True)
