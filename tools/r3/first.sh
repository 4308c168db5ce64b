#!/bin/bash
# round 3, GPU call 1: full GPU suite (new wide-table / precompute tests included), residency experiments on the
# headline kernel (work-item size A/B + per-wave trace), baseline bench line.
export TMPDIR=/tmp
O=gpurun_out/r3_first
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
export CKZG_HIP_PROOF_WBITS=0 CKZG_HIP_FK20_WBITS=8     # only the 16-bit commitment table for the kernel A/Bs
for ppb in 0 32768 16384 8192; do
  [ $ppb = 0 ] && unset CKZG_HIP_PPB || export CKZG_HIP_PPB=$ppb
  timeout 300 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-pcie 2> $O/ppb_$ppb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ppb $ppb', d['value'], d['roofline']['kernel_ms'], d['roofline_valu']['frac'])" >> $O/ppb_ab.log 2>&1
done
cat $O/ppb_ab.log
for ppb in 0 32768 8192; do
  [ $ppb = 0 ] && unset CKZG_HIP_PPB || export CKZG_HIP_PPB=$ppb
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_trace.so CKZG_HIP_MSM_TRACE_FILE=$O/trace_$ppb.bin timeout 300 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-pcie > $O/trace_$ppb.json 2> $O/trace_$ppb.err
  python tools/msm_trace.py $O/trace_$ppb.bin > $O/trace_$ppb.summary.json 2>> $O/trace_$ppb.err
  cat $O/trace_$ppb.summary.json
  gzip -f $O/trace_$ppb.bin
done
unset CKZG_HIP_PPB CKZG_HIP_PROOF_WBITS CKZG_HIP_FK20_WBITS
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json | head -c 3000
