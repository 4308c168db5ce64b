#!/bin/bash
# (sweeps tuning constants: needs the A/B build -- bash tools/build_variant.sh ab -DCKZG_AB; export CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so)
# A/B of the lanes-per-vector forms of k_msm_small (CKZG_HIP_SMALL_LPV = 4 | 8 | 16) inside one gpurun call:
# correctness of the forced forms on the cells tests, then the 2048-blob batch row from the wide tables.
export TMPDIR=/tmp
rm -f gpurun_out/r2_lpv.log
for v in 4 8; do
  echo "== tests with CKZG_HIP_SMALL_LPV=$v" >> gpurun_out/r2_lpv.log
  CKZG_HIP_SMALL_LPV=$v timeout 200 python -m pytest tests/test_gpu_cells.py tests/test_gpu_recover_batch.py -m gpu -x -q --timeout=150 2>&1 | tail -1 >> gpurun_out/r2_lpv.log
done
for v in 16 8 4; do
  echo "== CKZG_HIP_SMALL_LPV=$v" >> gpurun_out/r2_lpv.log
  CKZG_HIP_SMALL_LPV=$v timeout 240 python tools/row_driver.py cells wide 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d['batch_2048']
print('batch2048', b['blobs_per_s'], 'host', b['host_pointer_blobs_per_s'], 'k_msm_small ms', b['k_msm_small_ms'], 'g1_fft ms', b.get('g1_fft_ms'))" >> gpurun_out/r2_lpv.log 2>&1
done
cat gpurun_out/r2_lpv.log
