#!/bin/bash
# round 3, GPU call 2: full GPU suite, fair-priority time slices on the fixed-base MSM kernels (A/B + trace),
# the three forms of the 4096-blob verification.
export TMPDIR=/tmp
O=gpurun_out/r3_second
rm -rf $O && mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
export CKZG_HIP_PROOF_WBITS=0 CKZG_HIP_FK20_WBITS=8     # only the 16-bit commitment table for the kernel A/Bs
for bit in 0 9 11 13 15; do
  CKZG_HIP_MSM_PRIO_BIT=$bit timeout 300 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2> $O/prio_$bit.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio_bit $bit', d['value'], d['roofline']['kernel_ms'], d['roofline_valu']['frac'], 'host', d['host_pointer']['value'])" >> $O/prio_ab.log 2>&1
done
cat $O/prio_ab.log
for bit in 11 13; do
  CKZG_HIP_MSM_PRIO_BIT=$bit CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_trace.so CKZG_HIP_MSM_TRACE_FILE=$O/trace_prio$bit.bin timeout 300 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-pcie > $O/trace_prio$bit.json 2> $O/trace_prio$bit.err
  python tools/msm_trace.py $O/trace_prio$bit.bin > $O/trace_prio$bit.summary.json 2>> $O/trace_prio$bit.err
  cat $O/trace_prio$bit.summary.json
  gzip -f $O/trace_prio$bit.bin
done
unset CKZG_HIP_PROOF_WBITS CKZG_HIP_FK20_WBITS
# cells rows (one blob on the direct path = one round of k_msm_accumulate; 2048-blob FK20 = k_msm_small)
for cfg in "0 0" "13 0" "13 13" "11 11"; do
  set -- $cfg
  CKZG_HIP_MSM_PRIO_BIT=$1 CKZG_HIP_SMALL_PRIO_BIT=$2 timeout 300 python tools/row_driver.py cells wide 2> $O/cells_$1_$2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio $1 small $2: 1 blob ms', d['one_blob']['ms_per_call'], 'kernel', d['one_blob']['roofline']['kernel_ms'], 'batch2048', d['batch_2048']['blobs_per_s'], 'k_msm_small ms', d['batch_2048']['k_msm_small_ms'], 'g1_fft ms', d['batch_2048'].get('g1_fft_ms'))" >> $O/cells_ab.log 2>&1
done
cat $O/cells_ab.log
CKZG_HIP_TRACE=1 timeout 300 python tools/bench_verify_forms.py 4096 5 > $O/verify_forms.json 2> $O/verify_forms.err
cat $O/verify_forms.json; grep "trace\|--" $O/verify_forms.err | tail -40
