#!/bin/bash
# A/B of the Montgomery product variants on the headline kernel, inside one gpurun call:
# default build (column-wise, inline-asm mads) vs tools/build_variant.sh builds.  Every step is bounded.
export TMPDIR=/tmp
rm -f gpurun_out/r2_ab.log
for v in "" _rowwise _noasm; do
  [ -f c-kzg-4844_amd/libckzg_hip$v.so ] || continue
  echo "== variant libckzg_hip$v.so" >> gpurun_out/r2_ab.log
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 240 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel_ms'], d['roofline_valu']['frac'])" >> gpurun_out/r2_ab.log 2>&1
done
cat gpurun_out/r2_ab.log
