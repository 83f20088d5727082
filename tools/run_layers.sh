#!/bin/bash
# kernel-only per-layer table of the ResNet-12 matrix kernels (tools/bench_rn12_layers.py under rocprofv3 --kernel-trace)
# usage: tools/run_layers.sh <tag> [B] [M]
tag=${1:-base}; B=${2:-4}; M=${3:-100}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o lay -- python3 $R/tools/bench_rn12_layers.py $B $M > $R/gpurun_out/layers_${tag}_events.txt 2>&1
cd $R
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python tools/layers_from_trace.py $f $B $M > gpurun_out/layers_${tag}.txt
cat gpurun_out/layers_${tag}.txt
