import sys, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
g = torch.Generator().manual_seed(5)
B, S, Qn, D, h0 = 8, 25, 160, 2048, 256
x_s = torch.randn(B, S, D, generator=g).abs(); x_q = torch.randn(B, Qn, D, generator=g).abs() * 3.0
W0 = torch.randn(h0, D, generator=g) * 0.02
A0, G = hip.xpanel_fwd(ws, x_s.to(dev), x_q.to(dev), W0.to(dev))
X = torch.cat([x_s, x_q], 1).double()
A0r = X @ W0.double().T; Gr = X @ x_s.double().transpose(1, 2)
print(json.dumps({"A0 max rel": float((A0.cpu().double() - A0r).abs().max() / A0r.abs().max()), "A0 rms rel": float(((A0.cpu().double() - A0r) ** 2).mean().sqrt() / (A0r ** 2).mean().sqrt()),
                  "G max rel": float((G.cpu().double() - Gr).abs().max() / Gr.abs().max()), "G rms rel": float(((G.cpu().double() - Gr) ** 2).mean().sqrt() / (Gr ** 2).mean().sqrt())}))
