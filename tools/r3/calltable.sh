#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_calltable
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for w in 0 5 6 7 8; do
  echo "== CKZG_HIP_VERIFY_TABLE_WBITS=$w" >> $O/forms.log
  CKZG_HIP_VERIFY_TABLE_WBITS=$w timeout 300 python tools/bench_verify_forms.py 4096 7 2>/dev/null >> $O/forms.log
done
for w in 0 6; do
  echo "== n=1024 CKZG_HIP_VERIFY_TABLE_WBITS=$w" >> $O/forms.log
  CKZG_HIP_VERIFY_TABLE_WBITS=$w timeout 300 python tools/bench_verify_forms.py 1024 7 2>/dev/null >> $O/forms.log
done
cat $O/forms.log
CKZG_HIP_TRACE=1 timeout 300 python tools/bench_verify_forms.py 4096 3 2>&1 >/dev/null | grep -A8 -- "-- pinned" | head -10
