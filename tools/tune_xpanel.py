"""Dev tool (GPU box): time the two X-panel kernels at the bench shape.  FUMI_XP_* env knobs select variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fumi_amd import hip

dev = torch.device("cuda:0")
ws = hip.Workspace.get(dev)
B, S, Qn, D, h0 = 32, 25, 160, 2048, 256
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(B, S, D, device=dev, generator=g) for _ in range(4)]
xq = [torch.randn(B, Qn, D, device=dev, generator=g) for _ in range(4)]
W0 = torch.randn(h0, D, device=dev, generator=g) / 45
Ab = torch.randn(B, S + Qn, h0, device=dev, generator=g)

def timeit(fn, n=50):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

A0, G = hip.xpanel_fwd(ws, xs[0], xq[0], W0)
X = torch.cat([xs[0], xq[0]], 1)
refA = torch.einsum("brd,hd->brh", X.double(), W0.double())
refG = torch.einsum("brd,bsd->brs", X.double(), xs[0].double())
print("fwd err", float((A0 - refA).abs().max() / refA.abs().max()), float((G - refG).abs().max() / refG.abs().max()))
gW = hip.xpanel_bwd(ws, xs[0], xq[0], Ab)
refW = torch.einsum("brh,brd->hd", Ab.double(), X.double())
print("bwd err", float((gW - refW).abs().max() / refW.abs().max()))
tf = timeit(lambda i: hip.xpanel_fwd(ws, xs[i % 4], xq[i % 4], W0))
tb = timeit(lambda i: hip.xpanel_bwd(ws, xs[i % 4], xq[i % 4], Ab))
ff = 2.0 * B * (S + Qn) * (h0 + S) * D
fb = 2.0 * B * (S + Qn) * h0 * D
print(f"fwd {tf:.1f} us  {ff / tf / 1e6:.1f} TFLOP/s | bwd(+reduce) {tb:.1f} us {fb / tb / 1e6:.1f} TFLOP/s | env", {k: v for k, v in os.environ.items() if k.startswith("FUMI_XP")})
