"""FuMI at 20-way shapes (BASELINE.json configs[4] without its ResNet-12): parity against the oracle at B = 4 and the
time of a 64-episode step (dev tool)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from fumi_amd import hip
from oracle import casegen as cg, fumi_ref as R
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
for (B, N, K, Q, D, hid, Dt, Ht, T) in [(4, 20, 5, 15, 2048, [256, 64], 768, 256, 5), (4, 20, 5, 32, 2048, [256, 64], 768, 256, 2)]:
    ep = cg.make_episodes(7, B, N, K, Q, D, Dt); theta, phi = cg.make_fumi_params(7, D, hid, Dt, Ht)
    g = lambda t: t.to(dev).contiguous()
    args = (ws, N, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), g(ep["text_s"]), [g(t) for t in theta], [g(t) for t in phi], T, 0.01, False)
    out = hip.fumi_step_select(*args)
    assert ws.read_status() == 0
    th = [t.clone().requires_grad_(True) for t in theta]; ph = [t.clone().requires_grad_(True) for t in phi]
    ref = R.fumi_meta_step(th, ph, ep["text_s"], ep["x_s"], ep["y_s"], ep["x_q"], ep["y_q"], N, T, 0.01, False)
    el = float((out["logits"].cpu() - ref["logits"]).abs().max() / ref["logits"].abs().max())
    per = [(float((a.cpu() - b).abs().max()), float(b.abs().max())) for a, b in zip(out["g_theta"] + out["g_phi"], ref["g_theta"] + ref["g_phi"])]
    print("per-tensor (max abs err, max abs ref):", [(f"{e:.2e}", f"{m:.2e}") for e, m in per], flush=True)
    floor = 1e-3 * max(m for _, m in per)                 # (a gradient that is analytically zero has no relative error)
    eg = max(e / max(m, floor) for e, m in per)
    B2 = 64
    ep = cg.make_episodes(8, B2, N, K, Q, D, Dt)
    args = (ws, N, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), g(ep["text_s"]), [g(t) for t in theta], [g(t) for t in phi], T, 0.01, False)
    for _ in range(3): hip.fumi_step_select(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hip.fumi_step_select(*args)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"N={N} K={K} Q={Q} T={T}: logits rel err {el:.2e}, worst grad rel err {eg:.2e}; B=64: {ms:.3f} ms/step = {B2 / ms * 1e3:.0f} episodes/s", flush=True)
