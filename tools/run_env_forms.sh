#!/bin/bash
# The parity suite (tests/test_hip_parity.py + tests/test_sampler.py + tests/test_lstm_finetune_gpu.py) under every alternative code path an environment knob selects.
#   bash tools/run_env_forms.sh > gpurun_out/env_forms.txt
for form in "" FUMI_EPI_FUSE=0 FUMI_XP_PS=0 FUMI_XPB_NB=1 FUMI_XP_RIDER=0 FUMI_XPB_SB=0 FUMI_XP_SB=0 FUMI_HYPER_BWD=0 FUMI_EPI_OVERLAP=0 FUMI_EPI_OVERLAP=2 FUMI_GLOVE_RIDE=0 FUMI_ADAM_FUSE=0 FUMI_RN_S16=0; do
  echo "== ${form:-default}"
  if [ -n "$form" ]; then export $form; fi
  timeout -k 10 400 python -m pytest tests/test_hip_parity.py tests/test_sampler.py tests/test_lstm_finetune_gpu.py -m gpu -q 2>&1 | tail -1
  if [ -n "$form" ]; then unset ${form%%=*}; fi
done
