mkdir -p gpurun_out/r6
export TMPDIR=/tmp
tools/ubench/exec_skip > gpurun_out/r6/exec_skip.txt 2>&1
cat gpurun_out/r6/exec_skip.txt
for spec in "0 1" "32 1" "32 0" "0 1" "32 1" "32 0"; do
  set -- $spec
  echo "== CKZG_HIP_SYNC_STEP_DIV=$1 CKZG_HIP_SYNC_PREDICT=$2" >> gpurun_out/r6/sync_ab2.txt
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so CKZG_HIP_SYNC_STEP_DIV=$1 CKZG_HIP_SYNC_PREDICT=$2 timeout 300 python bench.py --steps 30 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','value_host_pointer')}, d['roofline']['kernel_ms'])" >> gpurun_out/r6/sync_ab2.txt 2>&1
done
cat gpurun_out/r6/sync_ab2.txt
(time timeout 1100 python -m pytest tests -x -q -m gpu --durations=12 -p no:cacheprovider) > gpurun_out/r6/gpu_suite4.log 2>&1
echo "suite rc=$?" >> gpurun_out/r6/gpu_suite4.log
tail -4 gpurun_out/r6/gpu_suite4.log
