"""Dev probe: which workgroup ids of the forward X-panel grid share a CU (placement only affects speed; the id -> tile map can
use it to let co-resident workgroups stream the same operand).  FUMI_XP_SB=0 python tests/dev/trace_placement.py"""
import os, sys, ctypes, collections
os.environ["FUMI_XP_SB"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
B, S, Qn, D, h0 = 32, 25, 160, 2048, 256
g = torch.Generator(device=dev).manual_seed(0)
xs = torch.randn(B, S, D, device=dev, generator=g); xq = torch.randn(B, Qn, D, device=dev, generator=g)
W0 = torch.randn(h0, D, device=dev, generator=g) / 45
for _ in range(3): hip.xpanel_fwd(ws, xs, xq, W0)
L = hip.lib()
for rep in range(3):
    tr = torch.zeros(480 * 6, dtype=torch.int64, device=dev)
    L.fumi_hip_set_trace_buffer(1, ctypes.c_void_p(tr.data_ptr()))
    hip.xpanel_fwd(ws, xs, xq, W0); torch.cuda.synchronize()
    L.fumi_hip_set_trace_buffer(1, None)
    t = tr.cpu().view(480, 6)
    hw = t[:, 4]; xcc = t[:, 5] & 0xF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    by = collections.defaultdict(list)
    for i in range(480):
        by[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(i)
    diffs = collections.Counter()
    for k, v in by.items():
        if len(v) == 2: diffs[(v[1] - v[0])] += 1
    print("rep", rep, "CUs", len(by), "pair id differences (top):", diffs.most_common(8))
    if rep == 0:
        x0 = sorted((k, v) for k, v in by.items() if k[0] == int(xcc[0]))
        print("XCD of id 0:", [(k[1:], v) for k, v in x0][:40])
