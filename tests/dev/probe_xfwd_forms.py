"""Dev: forward X-panel unit op against fp64 at configs[2]'s per-rank shape under the current environment's kernel choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
B, S, Qn, D, h0 = 32, 25, 160, 2048, 256
g = torch.Generator().manual_seed(0)
x_s, x_q, W0 = torch.randn(B, S, D, generator=g), torch.randn(B, Qn, D, generator=g), torch.randn(h0, D, generator=g) / D ** 0.5
A0, G = hip.xpanel_fwd(ws, x_s.to(dev), x_q.to(dev), W0.to(dev))
X = torch.cat([x_s, x_q], 1).double()
A0r = X @ W0.double().t(); Gr = X @ x_s.double().transpose(1, 2)
eA = (A0.cpu().double() - A0r).abs(); eG = (G.cpu().double() - Gr).abs()
print("env", {k: v for k, v in os.environ.items() if k.startswith("FUMI_")})
print("A0 max err", float(eA.max()), "of", float(A0r.abs().max()), " G max err", float(eG.max()), "of", float(Gr.abs().max()))
bad = (eG > 1e-2).nonzero()
print("bad G entries:", bad.shape[0], bad[:10].tolist())
badA = (eA > 1e-3).nonzero()
print("bad A0 entries:", badA.shape[0], badA[:10].tolist())
