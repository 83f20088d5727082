timeout -k 10 400 python -m pytest tests/test_resnet12_gpu.py -q -x 2>&1 | tail -3
for cfg in "FUMI_RN_WSPLIT=1 FUMI_RN_SIDE=0" "FUMI_RN_WSPLIT=0 FUMI_RN_SIDE=0" "FUMI_RN_WSPLIT=1 FUMI_RN_SIDE=1" "FUMI_RN_WSPLIT=0 FUMI_RN_SIDE=1"; do
  echo "== $cfg"; env $cfg RN12_PHASES=1 timeout -k 10 200 python tools/bench_resnet12.py 8 1 5 15 2>&1 | tail -2
done
