#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3_eleventh
rm -rf $O && mkdir -p $O
export CKZG_HIP_PROOF_WBITS=0 CKZG_HIP_FK20_WBITS=8
for rep in 1 2; do
for ch in 128 192 256 384 512; do
  CKZG_HIP_COMMIT_CHUNK=$ch timeout 300 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ch resident', d['value'], 'pageable', d['value_host_pointer'], 'pinned', d['host_pointer']['pinned_caller_memory_blobs_per_s_this_rank'])" >> $O/chunk.log
done
done
cat $O/chunk.log
