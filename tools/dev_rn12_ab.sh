for cfg in "FUMI_RN_XCD=1"; do
  echo "== $cfg"; env $cfg RN12_PHASES=1 timeout -k 10 200 python tools/bench_resnet12.py 8 1 5 15 2>&1 | tail -2
done
for ks in 4 8 16; do
  echo "== am3 FUMI_XP_KSPLIT=$ks"; env FUMI_XP_KSPLIT=$ks timeout -k 10 200 python tools/bench_configs.py --only am3_b32 --roofline 2>&1 | tail -1 | cut -c1-600
done
