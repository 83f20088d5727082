R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 400 python -m pytest tests/test_resnet12_gpu.py -q -x 2>&1 | tail -2
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_ab; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ab -o lay -- python3 $R/tools/bench_rn12_layers.py 4 100 > /tmp/ab_events.txt 2>&1; cd $R; f=$(find /tmp/prof_ab -name "*kernel_trace.csv" | head -1); python tools/layers_from_trace.py $f 4 100 | tail -16 )
for i in 1 2 3; do timeout -k 10 300 python tools/bench_resnet12.py 16 2 5 15 2>&1 | tail -1; done
