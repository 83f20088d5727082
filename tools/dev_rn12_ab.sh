# dev scratch: weight gradient with fewer x-fragment reads (FUMI_RN_WHACK: 1 = one per k-step, 2 = one per dy row; wrong results, timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "FUMI_RN_WHACK=0" "FUMI_RN_WHACK=2" "FUMI_RN_WHACK=1"; do
  echo "== layers $cfg"
  ( export $cfg; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_ab; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ab -o lay -- python3 $R/tools/bench_rn12_layers.py 4 100 wgrad > /tmp/ab_events.txt 2>&1; cd $R; f=$(find /tmp/prof_ab -name "*kernel_trace.csv" | head -1); python - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
wg = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "rn_wgrad_kernel" in r["Kernel_Name"]]
SH = [(16, 64, 84, 3), (64, 64, 84, 3), (16, 64, 84, 1), (64, 160, 42, 3), (160, 160, 42, 3), (64, 160, 42, 1), (160, 320, 21, 3), (320, 320, 21, 3), (160, 320, 21, 1), (320, 640, 10, 3), (640, 640, 10, 3), (320, 640, 10, 1)]
tot = 0
for i, (ci, co, H, k) in enumerate(SH):
    w = wg[7 * i:7 * i + 7]; d = sum(w[2:]) / 5; tot += d
    print(f"{ci:3d}->{co:3d} {H:2d}x{H:2d} k{k}  {d:8.0f} us  {2.0 * 4 * 100 * H * H * k * k * ci * co / d / 1e6:6.0f} TF")
print(f"total {tot / 1e3:.2f} ms")
PY
 )
done
