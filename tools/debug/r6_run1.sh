mkdir -p gpurun_out/r6
nproc > gpurun_out/r6/box1_info.txt; free -g >> gpurun_out/r6/box1_info.txt; rocm-smi --showmeminfo vram >> gpurun_out/r6/box1_info.txt 2>&1
(time timeout 1100 python -m pytest tests -x -q -m gpu --durations=40 -p no:cacheprovider) > gpurun_out/r6/gpu_suite1.log 2>&1
echo "suite rc=$?" >> gpurun_out/r6/gpu_suite1.log
tail -5 gpurun_out/r6/gpu_suite1.log
timeout 1500 python tools/debug/failalloc_loop.py 6 200 > gpurun_out/r6/failalloc_loop1.jsonl 2>&1
tail -3 gpurun_out/r6/failalloc_loop1.jsonl
