"""Dev probe: is a 5e-3 gradient difference between the fused / plain block-1 paths and the fp32 host oracle a bug or a ReLU / arg-max
decision that fell the other way?  Prints pairwise distances of the first conv weight's meta-gradient (fp64 oracle as arbiter)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fumi_amd import hip  # noqa: E402
from oracle import conv4_ref as C  # noqa: E402
from helpers import rel_to_max  # noqa: E402

dev = torch.device("cuda:0")
ws = hip.Workspace.get(dev)
g = lambda t: t.to(dev).contiguous()
B, N, K, Q, Cin, H, W, nblk, alpha = 3, 5, 1, 3, 3, 16, 16, 4, 0.05
for T, fo in ((2, True), (1, False), (3, False)):
    ep = C.make_image_episodes(21 + T, B, N, K, Q, Cin, H, W, 12)
    theta = C.make_conv4_params(21 + T, Cin, 64, nblk)
    gen = torch.Generator().manual_seed(9)
    p = theta + [torch.randn(N, 64, generator=gen) * 0.2, torch.randn(N, generator=gen) * 0.1]
    res = {}
    for mode in (1, 0):
        hip.conv4_set_option(0, mode)
        out = hip.maml_conv4_step(ws, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), [g(t) for t in p], T, alpha, fo)
        res["fused" if mode else "plain"] = [t.cpu() for t in out["g_params"]]
    hip.conv4_set_option(0, 1)
    for dt, nm in ((torch.float32, "cpu32"), (torch.float64, "cpu64")):
        pl = [t.to(dt).clone().requires_grad_(True) for t in p]
        r = C.maml_conv4_meta_step(pl, ep["x_s"].to(dt), ep["y_s"], ep["x_q"].to(dt), ep["y_q"], T, alpha, fo)
        res[nm] = r["g_params"]
    for i in (0, 3, 6, 9):
        ref = res["cpu64"][i]
        print(f"T={T} fo={fo} grad{i}: " + "  ".join(f"{k} vs f64 {rel_to_max(res[k][i], ref, 1e-9):.2e}" for k in ("fused", "plain", "cpu32")) +
              f"  fused vs plain {rel_to_max(res['fused'][i], res['plain'][i], 1e-9):.2e}", flush=True)
