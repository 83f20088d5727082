timeout -k 10 400 python -m pytest tests/test_resnet12_gpu.py -q -x 2>&1 | tail -2
for cfg in "FUMI_RN_LANES=2" "FUMI_RN_LANES=3" "FUMI_RN_LANES=4"; do
  echo "== $cfg B=24"; env $cfg timeout -k 10 300 python tools/bench_resnet12.py 24 2 5 15 2>&1 | tail -1
done
