echo "== one process B=32"; python tools/bench_conv4.py 32 20 1 32 2>&1 | tail -1
echo "== one process B=16"; python tools/bench_conv4.py 16 20 1 32 2>&1 | tail -1
echo "== two processes B=16 concurrently"
(python tools/bench_conv4.py 16 400 1 32 2>&1 | tail -1 > gpurun_out/h1.txt; date +%s.%N >> gpurun_out/h1.txt) &
(python tools/bench_conv4.py 16 400 1 32 2>&1 | tail -1 > gpurun_out/h2.txt; date +%s.%N >> gpurun_out/h2.txt) &
wait
cat gpurun_out/h1.txt gpurun_out/h2.txt
