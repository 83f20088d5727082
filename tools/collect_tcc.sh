#!/bin/bash
# L2 (TCC) request / hit / miss counters of the forward X-panel kernels: the pre-split form (default) and the 64 x 64 form (FUMI_XP_PS=0).
#   bash tools/collect_tcc.sh <tag>     -> gpurun_out/<tag>/tcc_{ps,sb}/..._counter_collection.csv
tag=${1:-tcc}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/tcc_ps" -- python3 "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-phase-timing --no-as-worded > /dev/null 2> "$out/tcc_ps.err"
export FUMI_XP_PS=0
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/tcc_sb" -- python3 "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-phase-timing --no-as-worded > /dev/null 2> "$out/tcc_sb.err"
find "$out" -name "*counter_collection.csv" | head
