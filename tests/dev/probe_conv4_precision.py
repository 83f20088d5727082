"""Dev probe: where does the 84x84 Conv4 step's logit difference come from?  GPU (fp32) and the fp32 host oracle are both
compared with the float64 host oracle, pass by pass."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fumi_amd import hip  # noqa: E402
from oracle import casegen as cg, conv4_ref as C  # noqa: E402
from helpers import rel_to_max  # noqa: E402

dev = torch.device("cuda:0")
ws = hip.Workspace.get(dev)
B, N, K, Q, Cin, H, W, nblk, Dt, Ht, alpha = 2, 5, 5, 3, 3, 84, 84, 4, 12, 24, 0.01
ep = C.make_image_episodes(77, B, N, K, Q, Cin, H, W, 12)
theta = C.make_conv4_params(77, Cin, 64, nblk)
Fd = 1600
_, phi = cg.make_fumi_params(77, 8, [Fd], Dt, Ht, head_scale=float(os.environ.get("HEAD_SCALE", "0.5")))
g = lambda t: t.to(dev).contiguous()
for T in (0, 1):
    out = hip.fumi_conv4_step(ws, N, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), [g(t) for t in theta], [g(t) for t in phi],
                              T, alpha, False, text_s=g(ep["text_s"]), need_grad=False)
    rg = lambda ts, dt: [t.to(dt).clone().requires_grad_(True) for t in ts]
    r32 = C.fumi_conv4_meta_step(rg(theta, torch.float32), rg(phi, torch.float32), ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, alpha, False, need_grad=False)
    d = lambda t: t.double()
    r64 = C.fumi_conv4_meta_step(rg(theta, torch.float64), rg(phi, torch.float64), d(ep["text_s"]), d(ep["x_s"]), ep["y_s"], d(ep["x_q"]),
                                 ep["y_q"], N, T, alpha, False, need_grad=False)
    print(f"T={T}: max|logit| {float(r64['logits'].abs().max()):.2f}  gpu vs f64 {rel_to_max(out['logits'].cpu(), r64['logits']):.2e}  "
          f"cpu32 vs f64 {rel_to_max(r32['logits'], r64['logits']):.2e}  gpu vs cpu32 {rel_to_max(out['logits'].cpu(), r32['logits']):.2e}", flush=True)
# features only
f = hip.conv4_features(ws, g(ep["x_q"]), [g(t) for t in theta]).cpu()
f32 = torch.stack([C.conv4_features(ep["x_q"][b], theta) for b in range(B)])
f64 = torch.stack([C.conv4_features(ep["x_q"][b].double(), [t.double() for t in theta]) for b in range(B)])
print(f"features: gpu vs f64 {rel_to_max(f, f64):.2e}  cpu32 vs f64 {rel_to_max(f32, f64):.2e}")

# gradients (second order, T = 1): engine and fp32 host oracle against the float64 host oracle, tensor by tensor
T = 1
out = hip.fumi_conv4_step(ws, N, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), [g(t) for t in theta], [g(t) for t in phi],
                          T, alpha, False, text_s=g(ep["text_s"]))
rg = lambda ts, dt: [t.to(dt).clone().requires_grad_(True) for t in ts]
r32 = C.fumi_conv4_meta_step(rg(theta, torch.float32), rg(phi, torch.float32), ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, alpha, False)
r64 = C.fumi_conv4_meta_step(rg(theta, torch.float64), rg(phi, torch.float64), ep["text_s"].double(), ep["x_s"].double(), ep["y_s"],
                             ep["x_q"].double(), ep["y_q"], N, T, alpha, False)
names = [f"theta{i}" for i in range(12)] + [f"phi{i}" for i in range(4)]
for n_, a, b32, b64 in zip(names, out["g_theta"] + out["g_phi"], r32["g_theta"] + r32["g_phi"], r64["g_theta"] + r64["g_phi"]):
    print(f"{n_:8s} max|g| {float(b64.abs().max()):.3e}  gpu vs f64 {rel_to_max(a.cpu(), b64, 1e-12):.2e}  cpu32 vs f64 {rel_to_max(b32, b64, 1e-12):.2e}", flush=True)
