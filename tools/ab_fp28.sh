#!/bin/bash
# A/B of Montgomery-product variants on the headline kernel and on the one-blob proof path, inside one gpurun
# call: the default build vs builds made by tools/build_variant.sh (named on the command line, e.g. "_asmmsm").
export TMPDIR=/tmp
rm -f gpurun_out/r2_ab.log
for v in "" "$@"; do
  [ -f c-kzg-4844_amd/libckzg_hip$v.so ] || continue
  echo "== variant libckzg_hip$v.so" >> gpurun_out/r2_ab.log
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 240 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('commit', d['value'], d['roofline']['kernel_ms'], d['roofline_valu']['frac'])" >> gpurun_out/r2_ab.log 2>&1
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 240 python tools/row_driver.py cells wide 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cells 1 blob ms', d['one_blob']['ms_per_call'], 'kernel', d['one_blob']['roofline']['kernel_ms'], 'batch2048', d['batch_2048']['blobs_per_s'], 'k_msm_small ms', d['batch_2048']['k_msm_small_ms'], 'g1_fft ms', d['batch_2048'].get('g1_fft_ms'))" >> gpurun_out/r2_ab.log 2>&1
done
cat gpurun_out/r2_ab.log
