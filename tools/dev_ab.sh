for cfg in "FUMI_CV_LANES=4 GPU_MAX_HW_QUEUES=8" "FUMI_CV_LANES=3 GPU_MAX_HW_QUEUES=8" "FUMI_CV_LANES=2 GPU_MAX_HW_QUEUES=8" "FUMI_CV_LANES=4 GPU_MAX_HW_QUEUES=16"; do
  echo "== $cfg"; env $cfg python tools/bench_conv4.py 32 30 1 32 2>&1 | tail -1
done
