timeout -k 10 400 python -m pytest tests/test_resnet12_gpu.py -q -x 2>&1 | tail -3
for cfg in "FUMI_RN_LANES=1" "FUMI_RN_LANES=2 FUMI_RN_LANE_THREAD=0" "FUMI_RN_LANES=2 FUMI_RN_LANE_THREAD=1"; do
  echo "== $cfg B=16"; env $cfg timeout -k 10 300 python tools/bench_resnet12.py 16 3 5 15 2>&1 | tail -1
done
