"""Dev: configs[2]'s per-rank FuMI step against the oracle, per-tensor errors, under the current environment's kernel choice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fumi_amd import hip
from oracle import casegen as cg, fumi_ref as R
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
B, N, K, Q, D, hid, Dt, Ht, T = 32, 5, 5, 32, 2048, [256, 64], 768, 256, int(os.environ.get("PROBE_T", "5"))
ep = cg.make_episodes(2025, B, N, K, Q, D, Dt)
theta, phi = cg.make_fumi_params(2025, D, hid, Dt, Ht)
g = lambda t: t.to(dev).contiguous()
out = hip.fumi_step_select(ws, N, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), g(ep["text_s"]), [g(t) for t in theta],
                           [g(t) for t in phi], T, cg.ALPHA, False)
torch.cuda.synchronize()
th = [t.clone().requires_grad_(True) for t in theta]; ph = [t.clone().requires_grad_(True) for t in phi]
ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, cg.ALPHA, False)
print("env", {k: v for k, v in os.environ.items() if k.startswith("FUMI_")}, "T", T)
print("logits", float((out["logits"].cpu() - ref["logits"]).abs().max() / ref["logits"].abs().max()))
for i, (a, b) in enumerate(zip(out["g_theta"] + out["g_phi"], ref["g_theta"] + ref["g_phi"])):
    e = (a.cpu() - b).abs()
    print(i, tuple(b.shape), f"max err {float(e.max()):.3e} of {float(b.abs().max()):.3e}", "worst idx", [int(v) for v in (e == e.max()).nonzero()[0]])
if os.environ.get("PROBE_SAVE"):
    torch.save([t.cpu() for t in out["g_theta"]], os.environ["PROBE_SAVE"])
