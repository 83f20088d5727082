#!/bin/bash
# End-of-round checks on the GPU box: the full -m gpu suite, then the one-stream kernel table of the ResNet-12 step.
set -o pipefail
mkdir -p gpurun_out/r03f
export TMPDIR=/tmp
root=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03f/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f/gputest.log
tail -4 gpurun_out/r03f/gputest.log
(cd /tmp && FUMI_RN_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/r03f/rn12_serial" -- python3 "$root/tools/bench_resnet12.py" 8 1 5 15 > "$root/gpurun_out/r03f/rn12_serial.log" 2>&1)
echo "[r03f] serial stats rc=$?"
