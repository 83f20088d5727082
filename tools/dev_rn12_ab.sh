# dev scratch: A/B of the ResNet-12 step's knobs at 8 episodes of configs[4]'s per-rank shape (phases printed from a one-stream step)
timeout -k 10 400 python -m pytest tests/test_resnet12_gpu.py -q -x 2>&1 | tail -2
for cfg in "FUMI_RN_LANES=1 FUMI_RN_SIDE=0" "FUMI_RN_LANES=1" "FUMI_RN_LANES=2"; do
  echo "== $cfg"; env $cfg RN12_PHASES=1 timeout -k 10 300 python tools/bench_resnet12.py 8 3 5 15 2>&1 | tail -2
done
