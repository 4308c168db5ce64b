mkdir -p gpurun_out/r6
export TMPDIR=/tmp
# bounded pipeline waits (product) against the round-5 spins (the ab library was built before the change), alternating
for rep in 1 2; do
  for lib in libckzg_hip_ab.so libckzg_hip.so; do
    echo "== $lib" >> gpurun_out/r6/pipe_wait_ab.txt
    CKZG_HIP_SO=c-kzg-4844_amd/$lib timeout 200 python tools/bench_small_batches.py --ops cells --sizes 2,4,8,12,16,32 >> gpurun_out/r6/pipe_wait_ab.txt 2>&1
  done
done
cat gpurun_out/r6/pipe_wait_ab.txt
(time timeout 1100 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider) > gpurun_out/r6/gpu_suite7.log 2>&1
echo "suite rc=$?" >> gpurun_out/r6/gpu_suite7.log
tail -4 gpurun_out/r6/gpu_suite7.log
