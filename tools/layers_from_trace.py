"""Kernel-only durations per layer shape from a rocprofv3 --kernel-trace CSV of tools/bench_rn12_layers.py (launch order is known:
per shape 7 forward convs, 7 input-gradient convs where the layer has one, 7 weight gradients; the last 5 of each are averaged).
python tools/layers_from_trace.py <kernel_trace.csv> [B] [M]"""
import csv
import sys

B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
M = int(sys.argv[3]) if len(sys.argv) > 3 else 100
SHAPES = [(16, 64, 84, 3), (64, 64, 84, 3), (16, 64, 84, 1), (64, 160, 42, 3), (160, 160, 42, 3), (64, 160, 42, 1),
          (160, 320, 21, 3), (320, 320, 21, 3), (160, 320, 21, 1), (320, 640, 10, 3), (640, 640, 10, 3), (320, 640, 10, 1)]
conv, wg = [], []
with open(sys.argv[1]) as f:
    rows = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "rn_conv_kernel" in r["Kernel_Name"]:
        conv.append((d, r["Kernel_Name"].split("rn_conv_kernel")[1].split("(")[0], r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
    elif "rn_wgrad_kernel" in r["Kernel_Name"]:
        wg.append((d, r["Kernel_Name"].split("rn_wgrad_kernel")[1].split("(")[0]))
ci_ = wi_ = 0
tot = {"fwd": 0.0, "bwd": 0.0, "wgrad": 0.0}
w_fl = {"fwd": 0.0, "bwd": 0.0, "wgrad": 0.0}
print(f"{'layer':22s} {'fwd us':>8s} {'TF':>6s} {'cfg':>10s}  {'bwd us':>8s} {'TF':>6s}  {'wgrad us':>8s} {'TF':>6s}")
for (ci, co, H, k) in SHAPES:
    fl = 2.0 * B * M * H * H * k * k * ci * co
    f = conv[ci_:ci_ + 7]; ci_ += 7
    fd = sum(x[0] for x in f[2:]) / 5
    line = f"{ci:3d}->{co:3d} {H:2d}x{H:2d} k{k}       {fd:8.0f} {fl / fd / 1e6:6.0f} {f[-1][1]:>10s}"
    tot["fwd"] += fd; w_fl["fwd"] += fl
    if ci % 32 == 0:
        b = conv[ci_:ci_ + 7]; ci_ += 7
        bd = sum(x[0] for x in b[2:]) / 5
        line += f"  {bd:8.0f} {fl / bd / 1e6:6.0f}"
        tot["bwd"] += bd; w_fl["bwd"] += fl
    else:
        line += "         -      -"
    if wg:
        w = wg[wi_:wi_ + 7]; wi_ += 7
        wd = sum(x[0] for x in w[2:]) / 5
        line += f"  {wd:8.0f} {fl / wd / 1e6:6.0f}"
        tot["wgrad"] += wd; w_fl["wgrad"] += fl
    print(line)
for k_ in tot:
    if tot[k_] == 0.0:
        continue
    print(f"{k_}: {tot[k_] / 1e3:.2f} ms, {w_fl[k_] / tot[k_] / 1e6:.0f} TFLOP/s over the listed layers")
