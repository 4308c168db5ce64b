mkdir -p gpurun_out/r6
export TMPDIR=/tmp
(time timeout 1100 python -m pytest tests -x -q -m gpu --durations=15 -p no:cacheprovider) > gpurun_out/r6/gpu_suite3.log 2>&1
echo "suite rc=$?" >> gpurun_out/r6/gpu_suite3.log
tail -4 gpurun_out/r6/gpu_suite3.log
# polling step of the bounded device waits: never sleep (floor) vs 1/16, 1/32 (product), 1/64 of the time waited
for div in 0 32 16 64 0 32; do
  echo "== CKZG_HIP_SYNC_STEP_DIV=$div" >> gpurun_out/r6/sync_ab.txt
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so CKZG_HIP_SYNC_STEP_DIV=$div timeout 300 python bench.py --steps 30 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','value_host_pointer')}, d['roofline']['kernel_ms'])" >> gpurun_out/r6/sync_ab.txt 2>&1
done
cat gpurun_out/r6/sync_ab.txt
PMC_MIN_SHARE=0.0003 bash tools/pmc_rows.sh cells_small_default cells_small_wide > gpurun_out/r6/pmc_small.log 2>&1
bash tools/pmc_rows.sh cells_wide verify_wide > gpurun_out/r6/pmc_wide.log 2>&1
tail -5 gpurun_out/r6/pmc_small.log
