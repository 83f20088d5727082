timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "am3 or AM3" 2>&1 | tail -4
for cfg in "FUMI_XP_PS=1" "FUMI_XP_PS=0"; do
  echo "== am3 $cfg"; env $cfg timeout -k 10 200 python tools/bench_configs.py --only am3_b32 --roofline 2>&1 | tail -1 | cut -c1-520
done
