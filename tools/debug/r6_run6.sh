mkdir -p gpurun_out/r6
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_deadlines.py -x -q -m gpu -p no:cacheprovider) > gpurun_out/r6/gpu_deadlines.log 2>&1
tail -15 gpurun_out/r6/gpu_deadlines.log
(time timeout 1100 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider) > gpurun_out/r6/gpu_suite6.log 2>&1
echo "suite rc=$?" >> gpurun_out/r6/gpu_suite6.log
tail -4 gpurun_out/r6/gpu_suite6.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke6.log 2>&1; tail -2 gpurun_out/r6/smoke6.log
