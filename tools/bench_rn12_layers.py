"""Per-layer rates of the ResNet-12 matrix kernels at configs[4]'s shapes (B episodes x M images, channels 64/160/320/640, 84 -> 10):
forward convolution, input-gradient convolution and weight gradient of every layer shape through the unit ops of the C ABI.
HIP events around 5 calls (the unit op's weight preparation + output memset ride along: 2-5 %); run under
`rocprofv3 --kernel-trace` and tools/kernel_by_grid.py for kernel-only durations per shape.
python tools/bench_rn12_layers.py [B] [M] [what=fwd,bwd,wgrad] [filter]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fumi_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100
what = (sys.argv[3] if len(sys.argv) > 3 else "fwd,bwd,wgrad").split(",")
flt = sys.argv[4] if len(sys.argv) > 4 else ""
dev = torch.device("cuda:0")
ws = hip.Workspace.get(dev)
SHAPES = [(16, 64, 84, 3), (64, 64, 84, 3), (16, 64, 84, 1), (64, 160, 42, 3), (160, 160, 42, 3), (64, 160, 42, 1),
          (160, 320, 21, 3), (320, 320, 21, 3), (160, 320, 21, 1), (320, 640, 10, 3), (640, 640, 10, 3), (320, 640, 10, 1)]
g = torch.Generator(device=dev).manual_seed(0)
tot = {}
for (ci, co, H, k) in SHAPES:
    name = f"{ci}->{co} {H}x{H} k{k}"
    if flt and flt not in name:
        continue
    x = torch.randn(B, M * (H + 2) ** 2, ci, device=dev, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, M * (H + 2) ** 2, co, device=dev, generator=g).to(torch.bfloat16)
    Wt = torch.randn(B, co, ci, k, k, device=dev, generator=g) / (ci * k * k) ** 0.5
    fl = 2.0 * B * M * H * H * k * k * ci * co
    row = [f"{name:22s}"]
    for w in what:
        if w == "bwd" and ci % 32:
            row.append("bwd      -    "); continue
        fn = {"fwd": lambda: hip.rn12_conv(ws, x, Wt, H, H, want_stats=True), "bwd": lambda: hip.rn12_conv(ws, dy, Wt, H, H, transpose=True),
              "wgrad": lambda: hip.rn12_wgrad(ws, x, dy, H, H, k)}[w]
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tot[w] = tot.get(w, 0.0) + ms
        row.append(f"{w} {ms * 1e3:7.0f} us {fl / ms / 1e9:6.0f} TF")
    print("  ".join(row), flush=True)
print("total ms:", {k_: round(v, 2) for k_, v in tot.items()}, flush=True)
