# dev scratch: end-of-round check on the GPU box (full -m gpu suite, smoke, default bench) -> gpurun_out/r03f/
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03f/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f/gputest.log
tail -4 gpurun_out/r03f/gputest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err; echo "bench rc=$?"
