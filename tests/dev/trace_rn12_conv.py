"""Dev: cycle stamps of one workgroup of the ResNet-12 convolution kernel (wave 0 of the middle workgroup): where a tile's time goes.
python tests/dev/trace_rn12_conv.py [Cin] [Cout] [H] [B] [M]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fumi_amd import hip
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 320
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 320
H = int(sys.argv[3]) if len(sys.argv) > 3 else 21
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
M = int(sys.argv[5]) if len(sys.argv) > 5 else 100
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev); L = hip.lib()
x = torch.randn(B, M * (H + 2) ** 2, Cin, device=dev).to(torch.bfloat16)
Wt = torch.randn(B, Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
for _ in range(3):
    hip.rn12_conv(ws, x, Wt, H, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    hip.rn12_conv(ws, x, Wt, H, H)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 2.0 * B * M * H * H * 9 * Cin * Cout
print(f"{Cin}->{Cout} {H}x{H} B={B} M={M}: {ms * 1e3:.0f} us per call (incl. weight prep), {fl / ms / 1e9:.0f} TFLOP/s")
import time
t_end = time.time() + 2.5                                  # (the clock settles under a sustained load: >= 2 s of back-to-back launches)
while time.time() < t_end:
    for _ in range(20):
        hip.rn12_conv(ws, x, Wt, H, H)
    torch.cuda.synchronize()
tr = torch.zeros(512, dtype=torch.int64, device=dev)
L.fumi_hip_set_trace_buffer(2, ctypes.c_void_p(tr.data_ptr()))
hip.rn12_conv(ws, x, Wt, H, H)
torch.cuda.synchronize()
L.fumi_hip_set_trace_buffer(2, None)
full = tr.cpu().tolist()
t = full[:8]
if full[9] > full[11]:
    print(f"in-kernel clock of the traced workgroup: {(full[8] - full[10]) / (full[9] - full[11]) * 0.1:.3f} GHz (s_memtime / s_memrealtime x 100 MHz)")
names = ["between chunks + slab issue", "slab + first weight tile landed", "weight loads issued", "k-steps (MFMA)", "wait for the older weight set",
         "weight tile -> LDS", "barrier", "epilogue"]
tot = sum(t)
print(f"cycles of wave 0 of the middle workgroup: {tot}")
for n_, v in zip(names, t):
    print(f"  {n_:36s} {v:9d}  {100.0 * v / tot:5.1f} %")
