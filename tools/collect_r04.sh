#!/bin/bash
# Round-4 evidence on the GPU box: bash tools/collect_r04.sh [part ...]  -> gpurun_out/r04/   (parts: bench stats pmc rn12 rn12pmc; default all)
# Every rocprofv3 run starts from /tmp with TMPDIR=/tmp; --pmc passes carry --kernel-trace only, one counter group per pass.
set -o pipefail
out=$PWD/gpurun_out/r04
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
parts=${@:-bench stats pmc rn12 rn12pmc}
HEAD="--steps 20 --warmup 5 --no-cpu-baseline --no-as-worded --no-configs4 --no-extra"
for part in $parts; do
  case $part in
    bench)   # the default line, not under a profiler
      (timeout -k 10 900 python3 bench.py > "$out/bench.json" 2> "$out/bench.err") || exit 1; echo "[r04] bench done";;
    stats)   # per-kernel table of the headline step
      (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -- python3 "$root/bench.py" $HEAD > "$out/bench_under_rocprof.json" 2> "$out/bench_stats.err") || exit 1
      cp "$(find "$out/bench_stats" -name '*kernel_stats.csv' | head -1)" "$out/bench_kernel_stats.csv"; echo "[r04] stats done";;
    pmc)     # HBM bytes per launch of the headline kernels: FETCH_SIZE and WRITE_SIZE in separate passes
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$c" -- python3 "$root/bench.py" $HEAD --no-phase-timing > /dev/null 2> "$out/pmc_$c.err") || exit 1
      done
      python3 tools/pmc_summary.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$out/pmc_traffic.json" > "$out/pmc_traffic.txt"; echo "[r04] pmc done";;
    rn12)    # configs[4]'s per-rank shape, 8 episodes, one stream: per-kernel durations that add up to the step
      (cd /tmp && FUMI_RN_SIDE=0 FUMI_RN_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rn12" -- python3 "$root/tools/bench_resnet12.py" 8 1 5 15 > "$out/rn12_step_serial.txt" 2>&1) || exit 1
      cp "$(find "$out/rn12" -name '*kernel_stats.csv' | head -1)" "$out/rn12_kernel_stats_serial.csv"; echo "[r04] rn12 done";;
    rn12pmc) # HBM bytes of the ResNet-12 kernels (bench.py's configs4.roofline.traffic reads the summary)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/rn12pmc_$c" -- python3 "$root/tools/bench_resnet12.py" 8 1 5 15 > "$out/rn12pmc_$c.log" 2>&1) || exit 1
      done
      python3 tools/pmc_summary.py "$out/rn12pmc_FETCH_SIZE" "$out/rn12pmc_WRITE_SIZE" "$out/rn12_pmc_traffic.json" \
        "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/bench_resnet12.py 8 1 5 15 = two 8-episode steps; FETCH_SIZE doubled (gfx950)" > "$out/rn12_pmc_traffic.txt"
      python3 - "$out/rn12_pmc_traffic.json" <<'PY'
import json, sys
p = sys.argv[1]; d = json.load(open(p)); d["steps"] = 2; d["episodes"] = 8; json.dump(d, open(p, "w"), indent=1)
PY
      echo "[r04] rn12pmc done";;
  esac
done
# keep the merge small: the raw counter / trace CSVs stay on the box
find "$out" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la "$out"
