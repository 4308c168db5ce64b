#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/icache
rm -rf $O && mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT[A-Z_]*" $O/counters.txt | sort -u | tr '\n' ' ' | head -c 3000; echo
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace --output-format csv -d $O/pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-pcie > /dev/null 2> $O/pmc.err
tail -3 $O/pmc.err
python tools/summarize_pmc.py k_msm_accumulate $O/pmc | grep -A8 counter_mean
