#!/bin/bash
# ResNet-12 (configs[4]) counters on the GPU box: bash tools/collect_rn12_pmc.sh <tag>   -> gpurun_out/<tag>/
# separate --pmc passes (SQ mix; L2 hits; FETCH_SIZE; WRITE_SIZE) of tools/bench_resnet12.py with 8 episodes, as the guide prescribes
set -o pipefail
tag=${1:-rn12pmc}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
run() {   # name, counters...
  name=$1; shift
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python3 "$root/tools/bench_resnet12.py" 8 1 5 15 > "$out/$name.log" 2>&1)
  echo "[pmc] $name rc=$?"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run tcc TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
find "$out" -name "*counter_collection.csv" | head
