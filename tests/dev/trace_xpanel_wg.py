"""Dev probe: per-workgroup timeline of the forward split-bf16 X-panel kernel at the bench shapes -- start, end of prologue, end of the
slab loop (100 MHz wall clock) and the CU each workgroup ran on.  python tests/dev/trace_xpanel_wg.py"""
import os, sys, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fumi_amd import hip
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
B, S, Qn, D, h0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 25, 160, 2048, 256
g = torch.Generator(device=dev).manual_seed(0)
xs = torch.randn(B, S, D, device=dev, generator=g); xq = torch.randn(B, Qn, D, device=dev, generator=g)
W0 = torch.randn(h0, D, device=dev, generator=g) / 45
L = hip.lib()
NW = 1024


def run(name, fn):
    for _ in range(3): fn()
    for rep in range(2):
        tr = torch.zeros(NW * 6, dtype=torch.int64, device=dev)
        L.fumi_hip_set_trace_buffer(1, ctypes.c_void_p(tr.data_ptr()))
        fn(); torch.cuda.synchronize()
        L.fumi_hip_set_trace_buffer(1, None)
        t = tr.cpu().view(NW, 6)
        live = t[:, 0] != 0
        t = t[live]
        n = t.shape[0]
        t00 = int(t[:, 0].min())
        start = (t[:, 0] - t00).float() / 100.0
        pro = (t[:, 1] - t[:, 0]).float() / 100.0
        loop = (t[:, 2] - t[:, 1]).float() / 100.0
        end = (t[:, 2] - t00).float() / 100.0
        hw = t[:, 4]; xcc = t[:, 5] & 0xF
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
        by = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
        occ = collections.Counter(by.values())
        print(f"{name} rep {rep}: {n} traced workgroups on {len(by)} CUs (workgroups per CU: {dict(occ)})")
        print(f"   start us: min {start.min():.1f} median {start.median():.1f} max {start.max():.1f}")
        print(f"   prologue us: median {pro.median():.2f} max {pro.max():.2f};  loop us: min {loop.min():.1f} median {loop.median():.1f} max {loop.max():.1f}")
        print(f"   loop end us: median {end.median():.1f} max {end.max():.1f}")
        kinds = collections.defaultdict(list)
        for i in range(n): kinds[int(t[i, 3])].append(float(loop[i]))
        for k, v in sorted(kinds.items()):
            v = torch.tensor(v); print(f"   kind/nslab {k}: {len(v)} wgs, loop median {v.median():.1f} max {v.max():.1f}")
        # loop time vs sharing a CU
        shared = [float(loop[i]) for i in range(n) if by[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))] > 1]
        alone = [float(loop[i]) for i in range(n) if by[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))] == 1]
        if shared and alone:
            print(f"   alone on a CU: {len(alone)} wgs, loop median {torch.tensor(alone).median():.1f};  sharing: {len(shared)} wgs, median {torch.tensor(shared).median():.1f}")


run("forward", lambda: hip.xpanel_fwd(ws, xs, xq, W0))
# (the backward kernel carries no trace code: six more VGPRs cost it its second workgroup per CU; DESIGN.md section 12 has what a temporary build showed)
