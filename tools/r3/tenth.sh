#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_tenth
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fan_out" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
export CKZG_HIP_PROOF_WBITS=0 CKZG_HIP_FK20_WBITS=8
CKZG_HIP_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $O/hp.json 2> $O/hp.err
grep "commit_batch" $O/hp.err | tail -12
python -c "
import json; d=json.loads(open('$O/hp.json').read().strip().splitlines()[-1]); print(d['value'], d['value_host_pointer'], d['host_pointer'])"
for ch in 256 512 1024; do
  CKZG_HIP_COMMIT_CHUNK=$ch timeout 300 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ch', d['value'], d['value_host_pointer'], d['host_pointer']['pinned_caller_memory_blobs_per_s_this_rank'])"
done
