"""Dev probe: every stage of the bf16 ResNet-12 path against torch / the manual sweep, errors printed (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from fumi_amd import hip
from oracle import resnet12_manual as M, resnet12_ref as RR, conv4_ref as CR, fumi_ref as R

dev = torch.device("cuda:0")
ws = hip.Workspace.get(dev)
bf = lambda t: t.to(torch.bfloat16).to(t.dtype)


def to_cl(x, Cpad=None):
    """[B, M, C, H, W] -> bf16 [B, M*(H+2)*(W+2), C]"""
    B, Mi, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 1, 3, 4, 2)
    if Cpad and Cpad > C:
        xp = F.pad(xp, (0, Cpad - C))
    return xp.reshape(B, Mi * (H + 2) * (W + 2), -1).contiguous().to(torch.bfloat16)


def from_cl(y, Mi, H, W):
    B, _, C = y.shape
    return y.float().reshape(B, Mi, H + 2, W + 2, C).permute(0, 1, 4, 2, 3)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def unit_ops():
    g = torch.Generator().manual_seed(0)
    for (B, Mi, H, W, Cin, Cout, k) in [(2, 3, 10, 10, 16, 64, 3), (1, 5, 21, 21, 64, 160, 3), (2, 2, 12, 9, 160, 64, 1), (1, 4, 7, 7, 320, 320, 3),
                                       (1, 2, 42, 42, 64, 64, 3), (1, 7, 5, 5, 640, 160, 3), (2, 3, 8, 8, 32, 96, 3), (1, 3, 9, 11, 96, 128, 1)]:
        x = bf(torch.randn(B, Mi, Cin, H, W, generator=g)); Wt = torch.randn(B, Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        ref = torch.stack([F.conv2d(x[b], bf(Wt[b]), None, padding=k // 2) for b in range(B)])
        y, st = hip.rn12_conv(ws, to_cl(x).to(dev), Wt.to(dev), H, W, want_stats=True)
        yc = from_cl(y.cpu(), Mi, H, W)
        border = float(yc[..., 0, :].abs().max() + yc[..., -1, :].abs().max() + yc[..., :, 0].abs().max() + yc[..., :, -1].abs().max())
        yi = yc[..., 1:-1, 1:-1]
        s_ref = torch.stack([yi.sum((1, 3, 4)), (yi * yi).sum((1, 3, 4))], 1)
        print(f"conv {B}x{Mi}x{H}x{W} {Cin}->{Cout} k{k}: err {rel(yi, ref):.2e} border {border:.1e} stats {rel(st.cpu(), s_ref):.2e}", flush=True)
        if Cin % 32 == 0:
            dy = bf(torch.randn(B, Mi, Cout, H, W, generator=g))
            ref = torch.stack([F.conv2d(dy[b], bf(Wt[b]).flip(2, 3).transpose(0, 1), None, padding=k // 2) for b in range(B)])
            dx = hip.rn12_conv(ws, to_cl(dy).to(dev), Wt.to(dev), H, W, transpose=True)
            print(f"   bwd-data: err {rel(from_cl(dx.cpu(), Mi, H, W)[..., 1:-1, 1:-1], ref):.2e}", flush=True)
            dW = hip.rn12_wgrad(ws, to_cl(x).to(dev), to_cl(dy).to(dev), H, W, k)
            ref = torch.stack([M.conv_bwd_weight(x[b].double(), dy[b].double(), k) for b in range(B)]).float()
            print(f"   wgrad: err {rel(dW.cpu(), ref):.2e}", flush=True)


def case(seed, B, N, K, Q, H, channels, Dt=6, Ht=5):
    ep = CR.make_image_episodes(seed, B, N, K, Q, 3, H, H, Dt)
    theta = RR.make_params(seed, 3, channels, torch.float32)
    rs = np.random.RandomState(seed)
    F_ = channels[-1]
    phi = [torch.from_numpy((rs.standard_normal(s) * 0.3).astype(np.float32)) for s in ((Ht, Dt), (Ht,), (F_ + 1, Ht), (F_ + 1,))]
    return ep, theta, phi


def features(channels=(32, 64, 64, 128), H=32):
    ep, theta, phi = case(5, 2, 3, 3, 2, H, channels)
    f = hip.resnet12_features(ws, ep["x_s"].to(dev), [t.to(dev) for t in theta])
    th64 = [t.double() for t in theta]
    for b in range(2):
        _, tb = M.net_fwd(ep["x_s"][b].double(), th64, torch.zeros(3, channels[-1] + 1, dtype=torch.float64), M.bf16_round)
        _, t64 = M.net_fwd(ep["x_s"][b].double(), th64, torch.zeros(3, channels[-1] + 1, dtype=torch.float64), M._id)
        print(f"features ep {b}: vs bf16 sweep {rel(f[b].cpu().double(), tb['f']):.2e}   vs fp64 {rel(f[b].cpu().double(), t64['f']):.2e}", flush=True)


def gerr(got, ref):
    num = sum(float(((a.cpu().double() - b) ** 2).sum()) for a, b in zip(got, ref))
    den = sum(float((b ** 2).sum()) for b in ref)
    worst = max(float((a.cpu().double() - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(got, ref))
    return (num / den) ** 0.5, worst


def steps(channels=(32, 64, 64, 128), H=32):
    for (T, fo, name) in [(0, False, "T=0"), (1, True, "T=1 first order"), (1, False, "T=1 second order"), (2, False, "T=2 second order")]:
        ep, theta, phi = case(7, 2, 3, 3, 2, H, channels)
        N = 3
        rs = np.random.RandomState(1)
        Wf = torch.from_numpy((rs.standard_normal((N, channels[-1])) * 0.1).astype(np.float32)); bfin = torch.zeros(N)
        params = theta + [Wf, bfin]
        out = hip.maml_resnet12_step(ws, ep["x_s"].to(dev), ep["y_s"].to(dev), ep["x_q"].to(dev), ep["y_q"].to(dev),
                                     [t.to(dev) for t in params], T, 0.05, fo)
        torch.cuda.synchronize()
        th64 = [t.double() for t in theta]
        h0 = torch.cat([Wf, bfin[:, None]], 1).double()
        for rname, rnd in (("bf16 sweep", M.bf16_round), ("fp64", M._id)):
            gs = [torch.zeros_like(t) for t in th64]; gh = torch.zeros_like(h0); zs = []
            for b in range(2):
                zq, loss, bth, bh = M.episode_grads(th64, h0, ep["x_s"][b].double(), ep["y_s"][b], ep["x_q"][b].double(), ep["y_q"][b], T, 0.05,
                                                    first_order=fo, rnd=rnd)
                zs.append(zq)
                for a, g_ in zip(gs, bth): a += g_ / 2
                gh += bh / 2
            e, w = gerr(out["g_params"][:-2], gs)
            if os.environ.get("PER_TENSOR") and rname == "bf16 sweep":
                print("      per tensor:", " ".join(f"{float((a.cpu().double() - b).norm() / b.norm().clamp_min(1e-30)):.3f}" for a, b in zip(out["g_params"][:-2], gs)), flush=True)
            ghg = torch.cat([out["g_params"][-2].cpu().double(), out["g_params"][-1].cpu().double()[:, None]], 1)
            print(f"{name} vs {rname}: logits {rel(out['logits'].cpu().double(), torch.stack(zs)):.2e}  grads relL2 {e:.2e} (worst tensor {w:.2e})  head {rel(ghg, gh):.2e}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["unit", "features", "steps"]
    if "unit" in what: unit_ops()
    if "features" in what: features()
    if "steps" in what: steps()
    if "small" in what:
        steps((32,), 8)
        steps((32, 64), 16)
        steps((64, 32, 96), 24)
