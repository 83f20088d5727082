"""Times the bf16 ResNet-12 FuMI meta-step (BASELINE.json configs[4]: 20-way 5-shot, 3x84x84 images, 5 inner steps, second-order)
straight through the C ABI on synthetic images resident in HBM.
python tools/bench_resnet12.py [B] [steps] [T] [Q] [chunk] [N]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fumi_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 5
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 15
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
N = int(sys.argv[6]) if len(sys.argv) > 6 else 20
K, Cin, H, W, Dt, Ht = 5, 3, 84, 84, 768, 256
CH = (64, 160, 320, 640)
dev = torch.device("cuda:0")
ws = hip.Workspace.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
S, Qn = N * K, N * Q
x_s = torch.randn(B, S, Cin, H, W, device=dev, generator=g)
x_q = torch.randn(B, Qn, Cin, H, W, device=dev, generator=g)
y_s = torch.arange(N, device=dev).repeat_interleave(K).repeat(B, 1)
y_q = torch.arange(N, device=dev).repeat_interleave(Q).repeat(B, 1)
cls_text = torch.randn(B, N, Dt, device=dev, generator=g)
F = CH[-1]
theta, ci = [], Cin
for c in CH:
    for (co, cin, k) in ((c, ci, 3), (c, c, 3), (c, c, 3), (c, ci, 1)):
        theta += [(torch.rand(co, cin, k, k, device=dev, generator=g) * 2 - 1) / (cin * k * k) ** 0.5, torch.ones(co, device=dev),
                  torch.zeros(co, device=dev)]
    ci = c
phi = [(torch.rand(Ht, Dt, device=dev, generator=g) * 2 - 1) / Dt ** 0.5, torch.zeros(Ht, device=dev),
       (torch.rand(F + 1, Ht, device=dev, generator=g) * 2 - 1) / Ht ** 0.5, torch.zeros(F + 1, device=dev)]
g_theta = [torch.empty_like(t) for t in theta]
g_phi = [torch.empty_like(t) for t in phi]


def flops_per_image():
    f, h, ci_ = 0, H, Cin
    for c in CH:
        f += 2 * h * h * (9 * ci_ * c + 2 * 9 * c * c + ci_ * c)
        h //= 2; ci_ = c
    return f


def step():
    return hip.fumi_resnet12_step(ws, N, x_s, y_s, x_q, y_q, theta, phi, T, 0.01, False, cls_text=cls_text, g_theta=g_theta, g_phi=g_phi,
                                  chunk=chunk)


t0 = time.perf_counter(); out = step(); torch.cuda.synchronize()
print(f"first call {time.perf_counter() - t0:.2f} s, workspace {ws.bytes() / 2**30:.1f} GiB, loss {float(out['loss_b'].mean()):.4f}", flush=True)
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
flops = B * flops_per_image() * (S * T * 9 + Qn * 3)
print(f"B={B} N={N} T={T} Q={Q} chunk={chunk}: {ms:.1f} ms/step, {B / ms * 1e3:.2f} episodes/s, {flops / ms / 1e9:.1f} TFLOP/s algorithmic "
      f"({flops_per_image() / 1e9:.2f} GFLOP per image forward)", flush=True)
if os.environ.get("RN12_PHASES"):
    ws.set_profiling(True, phases=["rn_conv", "rn_wgrad", "rn_ew"])
    step(); torch.cuda.synchronize()
    print(ws.profile(), flush=True)
