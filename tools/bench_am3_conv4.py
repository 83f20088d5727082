"""Times AM3 with the Conv4 backbone (BASELINE.json configs[3] as worded: 5-way 5-shot, 3x84x84 images, 32 queries per class)
straight through the C ABI: conv4_encode (tape kept) -> am3_step_dx -> conv4_encode_bwd, synthetic images resident in HBM.
python tools/bench_am3_conv4.py [B] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fumi_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, K, Q, Cin, H, W, nblk, Dt, Ht, P = 5, 5, 32, 3, 84, 84, 4, 768, 256, 64
dev = torch.device("cuda:0")
ws, ws_enc = hip.Workspace.get(dev), hip.Workspace.get(dev, "encoder")
g = torch.Generator(device=dev).manual_seed(0)
S, Qn = N * K, N * Q
x_s = torch.randn(B, S, Cin, H, W, device=dev, generator=g)
x_q = torch.randn(B, Qn, Cin, H, W, device=dev, generator=g)
y_s = torch.arange(N, device=dev).repeat_interleave(K).repeat(B, 1)
y_q = torch.arange(N, device=dev).repeat_interleave(Q).repeat(B, 1)
text = torch.randn(B, N, Dt, device=dev, generator=g)[:, y_s[0]]
F = hip.conv4_feature_dim(nblk, H, W)
theta = []
for l in range(nblk):
    ci = Cin if l == 0 else 64
    theta += [(torch.rand(64, ci, 3, 3, device=dev, generator=g) * 2 - 1) / (ci * 9) ** 0.5, torch.ones(64, device=dev), torch.zeros(64, device=dev)]
u = lambda *s, fan: (torch.rand(*s, device=dev, generator=g) * 2 - 1) / fan ** 0.5
w = [u(P, F, fan=F), u(P, fan=F), u(Ht, Dt, fan=Dt), u(Ht, fan=Dt), u(P, Ht, fan=Ht), u(P, fan=Ht), u(Ht, P, fan=P), u(Ht, fan=P),
     u(1, Ht, fan=Ht), u(1, fan=Ht)]
g_w = [torch.empty_like(t) for t in w]
g_theta = [torch.empty_like(t) for t in theta]


def step():
    f_s, f_q = hip.conv4_encode(ws_enc, x_s, x_q, theta, keep_tape=True)
    out = hip.am3_step(ws, f_s, y_s, f_q, y_q, text, w, N, None, g_w=g_w, dropout_p=0.25, seed=1, want_dx=True)
    hip.conv4_encode_bwd(ws_enc, x_s, x_q, out["dx_s"], out["dx_q"], theta, g_theta=g_theta)
    return out


t0 = time.perf_counter(); out = step(); torch.cuda.synchronize()
print(f"first call {time.perf_counter() - t0:.2f} s, workspaces {ws.bytes() / 2**30:.1f} + {ws_enc.bytes() / 2**30:.1f} GiB, "
      f"loss {float(out['loss']):.4f}", flush=True)
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
fimg = 2 * (84 * 84 * 27 * 64 + 42 * 42 * 576 * 64 + 21 * 21 * 576 * 64 + 10 * 10 * 576 * 64)
flops = B * fimg * (S + Qn) * 3          # forward + both backward products of every image
print(f"AM3 + Conv4, B={B}: {ms:.2f} ms/step, {B / ms * 1e3:.1f} episodes/s, {flops / ms / 1e9:.1f} TFLOP/s in the conv products", flush=True)
