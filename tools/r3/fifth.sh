#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_fifth
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for b in new old; do
  [ $b = old ] && export CKZG_HIP_TABLE_BUILDER=old || unset CKZG_HIP_TABLE_BUILDER
  timeout 300 python tools/bench_load.py 16 16 13 >> $O/load_ab.log 2>> $O/load_ab.err
  timeout 300 python tools/bench_load.py 10 8 8 >> $O/load_ab.log 2>> $O/load_ab.err
done
unset CKZG_HIP_TABLE_BUILDER
timeout 300 python tools/bench_load.py 16 16 13 async >> $O/load_ab.log 2>> $O/load_ab.err
cat $O/load_ab.log
timeout 900 bash tools/r3/san_probe.sh > $O/san_probe.log 2>&1
head -40 $O/san_probe.log | cut -c1-300
timeout 1500 bash tools/run_sanitized.sh gpu > $O/sanitize.out 2>&1; echo "sanitize rc=$?" >> $O/sanitize.out
tail -25 $O/sanitize.out | cut -c1-300; cp gpurun_out/sanitize_gpu.log $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 800 $O/bench_default.json; tail -3 $O/bench_default.err
