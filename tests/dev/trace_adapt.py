import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fumi_amd import hip
from oracle import casegen as cg
dev = torch.device("cuda:0"); ws = hip.Workspace.get(dev)
B, N, K, Q, D, hid, Dt, Ht, T = 32, 5, 5, 32, 2048, [256, 64], 300, 256, 1
ep = cg.make_episodes(1, B, N, K, Q, D, Dt); theta, phi = cg.make_fumi_params(1, D, hid, Dt, Ht)
g = lambda t: t.to(dev).contiguous()
args = (ws, N, g(ep["x_s"]), g(ep["y_s"]), g(ep["x_q"]), g(ep["y_q"]), g(ep["text_s"]), [g(t) for t in theta], [g(t) for t in phi], T, 0.01, False)
for _ in range(3): hip.fumi_step_select(*args)
tr = torch.zeros(256, dtype=torch.int64, device=dev)
L = hip.lib()
L.fumi_hip_set_trace_buffer(0, ctypes.c_void_p(tr.data_ptr()))
hip.fumi_step_select(*args); torch.cuda.synchronize()
L.fumi_hip_set_trace_buffer(0, None)
t = tr.cpu(); n = int((t[:32] > 0).sum())
d = [(int(t[i + 1]) - int(t[i])) / 100.0 for i in range(n - 1)]
mm = [(int(t[32 + i + 1]) - int(t[32 + i])) / 100.0 for i in range(6)]
print("z0 wg_mm2 internal (entry->acc init+pre, ->sync1, stage, ->sync2, compute, post):", [round(x, 2) for x in mm])
print("adapt phase durations (us):", [round(x, 1) for x in d], "total", round(sum(d), 1))
q = t[64:128]; nq = int((q > 0).sum())
dq = [(int(q[i + 1]) - int(q[i])) / 100.0 for i in range(nq - 1)]
print("query_lds phase durations (us):", [round(x, 2) for x in dq], "total", round(sum(dq), 1))
q = t[128:192]; nq = int((q > 0).sum())
dq = [(int(q[i + 1]) - int(q[i])) / 100.0 for i in range(nq - 1)]
print("reverse_lds phase durations (us):", [round(x, 2) for x in dq], "total", round(sum(dq), 1))
