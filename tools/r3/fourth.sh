#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_fourth
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 900 bash tools/r3/san_probe.sh > $O/san_probe.log 2>&1
tail -60 $O/san_probe.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -5 $O/bench_default.err
CKZG_HIP_TRACE=1 timeout 300 python tools/bench_verify_forms.py 4096 5 > $O/verify_forms.json 2> $O/verify_forms.err
cat $O/verify_forms.json
for cfg in "16 0" "16 16"; do
  set -- $cfg
  CKZG_HIP_MSM_PRIO_BIT=$1 CKZG_HIP_SMALL_PRIO_BIT=$2 timeout 300 python tools/row_driver.py cells wide 2> $O/cells_$1_$2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio $1 small $2: 1 blob ms', d['one_blob']['ms_per_call'], 'kernel', d['one_blob']['roofline']['kernel_ms'], 'batch2048', d['batch_2048']['blobs_per_s'], 'k_msm_small ms', d['batch_2048']['k_msm_small_ms'], 'g1_fft ms', d['batch_2048'].get('g1_fft_ms'))" >> $O/cells_ab.log 2>&1
done
cat $O/cells_ab.log
