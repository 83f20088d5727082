# dev scratch: A/B of the ResNet-12 step's knobs at 16 episodes of configs[4]'s per-rank shape (two lanes of 2 chunks of 4)
for cfg in "FUMI_RN_X=0" "FUMI_RN_NF640=4" "FUMI_RN_X=0" "FUMI_RN_NF640=4"; do
  echo "== $cfg"; env $cfg timeout -k 10 300 python tools/bench_resnet12.py 16 2 5 15 2>&1 | tail -1
done
