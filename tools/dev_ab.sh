timeout -k 10 600 python -m pytest tests/test_conv4_gpu.py -q -x 2>&1 | tail -3
for cfg in "FUMI_CV_LANES=1" "FUMI_CV_LANES=2" "FUMI_CV_LANES=3"; do
  echo "== $cfg"; env $cfg python tools/bench_am3_conv4.py 2>&1 | tail -1
done
