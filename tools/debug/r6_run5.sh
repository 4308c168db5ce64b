mkdir -p gpurun_out/r6
export TMPDIR=/tmp
bash tools/run_sanitized.sh gpu > gpurun_out/r6/san_gpu.out 2>&1; echo "san gpu rc=$?" >> gpurun_out/r6/san_gpu.out; tail -3 gpurun_out/r6/san_gpu.out
bash tools/run_sanitized.sh tsan > gpurun_out/r6/san_tsan.out 2>&1; echo "tsan rc=$?" >> gpurun_out/r6/san_tsan.out; tail -3 gpurun_out/r6/san_tsan.out
# soak: the stress drivers of rounds 3-5 on the final code
timeout 300 python tools/stress_gpu.py 40 8 1 > gpurun_out/r6/stress_churn.json 2> gpurun_out/r6/stress_churn.err; tail -c 400 gpurun_out/r6/stress_churn.json
timeout 900 python tools/debug/failalloc_loop.py 12 150 coalesced_callers > gpurun_out/r6/failalloc_loop5.jsonl 2>&1; tail -1 gpurun_out/r6/failalloc_loop5.jsonl
