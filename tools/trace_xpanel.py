"""Dev tool: per-workgroup start/end/placement trace of xpanel_fwd_kernel."""
import os, sys, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
B, S, Qn, D, h0 = 32, 25, 160, 2048, 256
g = torch.Generator(device=dev).manual_seed(0)
xs = torch.randn(B, S, D, device=dev, generator=g); xq = torch.randn(B, Qn, D, device=dev, generator=g)
W0 = torch.randn(h0, D, device=dev, generator=g) / 45
for _ in range(3): hip.xpanel_fwd(ws, xs, xq, W0)
tr = torch.zeros(480 * 6, dtype=torch.int64, device=dev)
L = hip.lib(); L.fumi_dbg_set_trace.argtypes = [ctypes.c_void_p]
L.fumi_dbg_set_trace(ctypes.c_void_p(tr.data_ptr()))
hip.xpanel_fwd(ws, xs, xq, W0); torch.cuda.synchronize()
L.fumi_dbg_set_trace(None)
t = tr.cpu().view(480, 6)
ghz = ((t[:, 3] - t[:, 2]).double() / ((t[:, 1] - t[:, 0]).double() * 10.0))     # cycles per ns
print(f"in-kernel shader clock: min {ghz.min():.3f} mean {ghz.mean():.3f} max {ghz.max():.3f} GHz")
t = torch.cat([t[:, :2], t[:, 4:]], 1)
t0 = int(t[:, 0].min())
start = (t[:, 0] - t0).float() / 100.0; end = (t[:, 1] - t0).float() / 100.0      # us
hw = t[:, 2]; xcc = t[:, 3] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = [(int(x), int(s_), int(h), int(c)) for x, s_, h, c in zip(xcc, se, sh, cu)]
cnt = collections.Counter(key)
print("distinct CUs used:", len(cnt), "max WGs on one CU:", max(cnt.values()), "hist:", collections.Counter(cnt.values()))
print("xcc of id%8==0..7 :", [sorted(set(int(xcc[i]) for i in range(j, 480, 8))) for j in range(8)])
print(f"start: min {start.min():.1f} max {start.max():.1f} us | end: min {end.min():.1f} max {end.max():.1f} | dur: min {(end-start).min():.1f} mean {(end-start).mean():.1f} max {(end-start).max():.1f}")
late = (start > 5).sum()
print("WGs starting later than 5us:", int(late))
