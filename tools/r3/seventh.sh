#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_seventh
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for q in 8192 16384 32768; do
  echo "== CKZG_HIP_QUAD_MAX=$q" >> $O/verify_forms.log
  CKZG_HIP_QUAD_MAX=$q CKZG_HIP_TRACE=1 timeout 300 python tools/bench_verify_forms.py 4096 7 >> $O/verify_forms.log 2> $O/verify_forms_$q.err
done
cat $O/verify_forms.log
grep -A8 -- "-- pinned" $O/verify_forms_8192.err | head -12
timeout 300 python tools/bench_load.py 16 16 13 async >> $O/load.log 2>> $O/load.err; cat $O/load.log | cut -c1-700
