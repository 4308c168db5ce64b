#!/bin/bash
# round 3, GPU call 3: full GPU suite (fuzz + bucket split included), sanitizer pass on the GPU, prio-slice
# confirmation, default bench line with the new fields.
export TMPDIR=/tmp
O=gpurun_out/r3_third
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 1500 bash tools/run_sanitized.sh gpu > $O/sanitize.out 2>&1; echo "sanitize rc=$?" >> $O/sanitize.out
tail -30 $O/sanitize.out; cp gpurun_out/sanitize_gpu.log $O/ 2>/dev/null
export CKZG_HIP_PROOF_WBITS=0 CKZG_HIP_FK20_WBITS=8
for bit in 0 15 17 0 15 19; do
  CKZG_HIP_MSM_PRIO_BIT=$bit timeout 300 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-pcie 2> $O/prio_$bit.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio_bit $bit', d['value'], d['roofline']['kernel_ms'], d['roofline_valu']['frac'])" >> $O/prio_ab.log 2>&1
done
cat $O/prio_ab.log
unset CKZG_HIP_PROOF_WBITS CKZG_HIP_FK20_WBITS
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 6000 $O/bench_default.json; tail -5 $O/bench_default.err
