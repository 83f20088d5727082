#!/bin/bash
# Collects the round's measurements on the GPU box into gpurun_out/$1 (copy what is to be judged into profiles/rNN afterwards).
#   bash tools/collect_profiles.sh r02v3
set -e -o pipefail
tag=${1:-run}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
echo "[collect] bench.py (default)"; python bench.py > "$out/bench.json" 2> "$out/bench.err"
echo "[collect] rocprofv3 kernel stats of bench.py"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_bench" -- python3 "$OLDPWD/bench.py" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err")
echo "[collect] PMC passes (faithful leg)"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$c" -- python3 "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-phase-timing --no-as-worded > /dev/null 2> "$out/pmc_$c.err")
done
python tools/pmc_summary.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$out/pmc_traffic.json" > "$out/pmc_summary.txt"
echo "[collect] other configurations"; python tools/bench_configs.py --roofline > "$out/other_configs_roofline.jsonl" 2> /dev/null
python tools/bench_configs.py --steps 200 > "$out/other_configs.jsonl" 2> /dev/null
echo "[collect] rocprofv3 kernel stats of AM3"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_am3" -- python3 "$OLDPWD/tools/bench_configs.py" --only am3_b32 --steps 100 > /dev/null 2> "$out/prof_am3.err")
echo "[collect] AM3 + Conv4"; python tools/bench_am3_conv4.py 32 5 > "$out/am3_conv4.txt" 2>&1
python tools/bench_conv4.py 32 5 > "$out/fumi_conv4.txt" 2>&1
find "$out" -name "*kernel_stats.csv" | head
echo "[collect] done"
