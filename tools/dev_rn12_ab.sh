python tests/dev/trace_rn12_conv.py 64 64 84 8 100 2>&1 | tail -10
python tests/dev/trace_rn12_conv.py 320 320 21 8 100 2>&1 | tail -10
RN12_PHASES=1 timeout -k 10 200 python tools/bench_resnet12.py 8 1 5 15 2>&1 | tail -2
timeout -k 10 400 python -m pytest tests/test_resnet12_gpu.py -q -x 2>&1 | tail -3
