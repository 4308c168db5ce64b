mkdir -p gpurun_out/r6
(time timeout 1100 python -m pytest tests -x -q -m gpu --durations=25 -p no:cacheprovider) > gpurun_out/r6/gpu_suite2.log 2>&1
echo "suite rc=$?" >> gpurun_out/r6/gpu_suite2.log
tail -4 gpurun_out/r6/gpu_suite2.log
timeout 900 python tools/debug/failalloc_loop.py 40 120 coalesced_callers > gpurun_out/r6/failalloc_loop2_coalesced.jsonl 2>&1
tail -2 gpurun_out/r6/failalloc_loop2_coalesced.jsonl
timeout 900 python tools/debug/failalloc_loop.py 3 200 > gpurun_out/r6/failalloc_loop2_all.jsonl 2>&1
tail -2 gpurun_out/r6/failalloc_loop2_all.jsonl
(time python bench.py) > gpurun_out/r6/bench2.log 2> gpurun_out/r6/bench2.err
tail -c 1500 gpurun_out/r6/bench2.log
