#!/bin/bash
# Conv4 as-worded (configs[1] wording) on the GPU box: SQ counters of every kernel (what does c1_kernel wait on?), and the
# T = 5 form (configs[2] wording) timed.   bash tools/collect_conv4_pmc.sh <tag>  -> gpurun_out/<tag>/
set -o pipefail
tag=${1:-c4pmc}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$out/sq" -- python3 "$root/tools/bench_conv4.py" 32 3 1 32 > "$out/sq.log" 2>&1)
echo "[c4] sq rc=$?"
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$out/mix" -- python3 "$root/tools/bench_conv4.py" 32 3 1 32 > "$out/mix.log" 2>&1)
echo "[c4] mix rc=$?"
timeout -k 10 300 python3 tools/bench_conv4.py 32 5 1 32 2>&1 | tail -2 > "$out/t1.txt"
timeout -k 10 300 python3 tools/bench_conv4.py 32 3 5 32 2>&1 | tail -2 > "$out/t5.txt"
cat "$out/t1.txt" "$out/t5.txt"
