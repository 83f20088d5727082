"""Dev: the convolution unit op at small odd shapes against a torch reference (interior pixels, statistics), under the current
environment's kernel form (FUMI_RN_S16=0 | 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
g = torch.Generator().manual_seed(0)
print("env", {k: v for k, v in os.environ.items() if k.startswith("FUMI_")})
for (ci, co, H, M, k, tr, st) in [(96, 96, 7, 8, 3, False, True), (96, 96, 7, 4, 3, False, True), (32, 96, 7, 8, 3, False, True), (32, 96, 7, 8, 1, False, True),
                                  (96, 160, 3, 8, 3, True, False), (96, 160, 3, 8, 1, True, False), (96, 96, 7, 8, 3, True, False), (160, 160, 3, 8, 3, False, True),
                                  (64, 64, 9, 30, 3, False, True), (160, 320, 5, 13, 3, False, True)]:
    B = 2
    x4 = torch.randn(B * M, (co if tr else ci), H, H, generator=g)
    xp = F.pad(x4, (1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(B, M * (H + 2) ** 2, -1).to(torch.bfloat16)
    Wt = torch.randn(B, co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    out = hip.rn12_conv(ws, xp.to(dev), Wt.to(dev), H, H, transpose=tr, want_stats=st)
    y, sst = (out if st else (out, None))
    Cy = ci if tr else co
    y4 = y.float().cpu().reshape(B * M, H + 2, H + 2, Cy)[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
    xin = xp.float().reshape(B * M, H + 2, H + 2, -1)[:, 1:-1, 1:-1].permute(0, 3, 1, 2).reshape(B, M, -1, H, H)
    Wb = Wt.to(torch.bfloat16).float()
    refs = []
    for b in range(B):
        if tr:
            refs.append(F.conv_transpose2d(xin[b], Wb[b], padding=k // 2))
        else:
            refs.append(F.conv2d(xin[b], Wb[b], padding=k // 2))
    ref = torch.cat(refs)
    e = float((y4 - ref).abs().max() / ref.abs().max())
    msg = f"{ci}->{co} {H}x{H} M={M} k{k} {'T' if tr else 'N'}: out err {e:.2e}"
    if st:
        yb = y4.reshape(B, M, Cy, H, H)
        s1 = yb.sum((1, 3, 4)); s2 = (yb * yb).sum((1, 3, 4))
        e1 = float((sst[:, 0].cpu() - s1).abs().max() / s1.abs().max()); e2 = float((sst[:, 1].cpu() - s2).abs().max() / s2.abs().max())
        msg += f"  stats err {e1:.2e} {e2:.2e}"
    print(msg)
