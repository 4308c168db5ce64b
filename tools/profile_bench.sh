#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: default bench line, kernel-trace stats of the
# timed-only run, and separate --pmc passes (never combined with --stats or other trace domains).
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/timed -- python bench.py --no-cpu-baseline --no-secondary --no-pcie > $O/bench_timed.json 2> $O/timed.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-pcie > /dev/null 2> $O/pmc_$c.err
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_SQ -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-pcie > /dev/null 2> $O/pmc_SQ.err
GRID=262144 python tools/summarize_pmc.py k_msm_accumulate $O/timed $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ > $O/summary.json
cat $O/bench_default.json; cat $O/bench_timed.json; head -c 1500 $O/summary.json
