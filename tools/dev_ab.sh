timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03f/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f/gputest.log
tail -6 gpurun_out/r03f/gputest.log
python tests/dev/trace_rn12_conv.py 320 320 21 8 100 2>&1 | tail -12
python tests/dev/trace_rn12_conv.py 64 64 84 8 100 2>&1 | tail -12
