# dev scratch: the register epilogue of the convolution (FUMI_RN_EPI=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
FUMI_RN_EPI=1 timeout -k 10 400 python -m pytest tests/test_resnet12_gpu.py -q -x 2>&1 | tail -3
for cfg in "FUMI_RN_EPI=0" "FUMI_RN_EPI=1"; do
  echo "== layers $cfg"
  ( export $cfg; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_ab; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ab -o lay -- python3 $R/tools/bench_rn12_layers.py 4 100 fwd,bwd > /tmp/ab_events.txt 2>&1; cd $R; f=$(find /tmp/prof_ab -name "*kernel_trace.csv" | head -1); python tools/layers_from_trace.py $f 4 100 | tail -15 )
done
for cfg in "FUMI_RN_EPI=0" "FUMI_RN_EPI=1" "FUMI_RN_EPI=0" "FUMI_RN_EPI=1"; do
  echo "== step $cfg"; env $cfg timeout -k 10 300 python tools/bench_resnet12.py 16 2 5 15 2>&1 | tail -1
done
