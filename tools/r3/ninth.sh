#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_ninth
rm -rf $O && mkdir -p $O
run() { echo "== $*" >> $O/sweep.log; env "$@" timeout 300 python tools/bench_verify_forms.py 4096 7 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pageable', d['pageable_ms'], 'pinned', d['pinned_ms'], 'resident', d['resident_ms'])" >> $O/sweep.log; }
run A=0
run CKZG_HIP_VERIFY_CHUNK=512
run CKZG_HIP_VERIFY_CHUNK=1024
run CKZG_HIP_VERIFY_CHUNK=128
run CKZG_HIP_HASH_THREADS=16
run CKZG_HIP_HASH_THREADS=64
run CKZG_HIP_HASH_THREADS=24
run HSA_ENABLE_SDMA=0
run CKZG_HIP_VERIFY_CHUNK=512 CKZG_HIP_HASH_THREADS=64
cat $O/sweep.log
