#!/bin/bash
# Round-3 evidence on the GPU box: bash tools/collect_r03.sh  -> gpurun_out/r03/
# (1) rocprofv3 --kernel-trace --stats of tools/bench_resnet12.py (8 episodes of configs[4]'s per-rank shape, 1 meta-step),
# (2) the same for the default bench.py, (3) the PMC passes of tools/collect_rn12_pmc.sh.
set -o pipefail
out=$PWD/gpurun_out/r03
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rn12" -- python3 "$root/tools/bench_resnet12.py" 8 1 5 15 > "$out/rn12.log" 2>&1) || exit 1
echo "[r03] rn12 stats done"
(cd /tmp && timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -- python3 "$root/bench.py" --steps 20 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/bench.err") || exit 1
echo "[r03] bench stats done"
[ "$1" = "nopmc" ] || bash tools/collect_rn12_pmc.sh r03/pmc
