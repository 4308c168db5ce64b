#!/bin/bash
# rocprofv3 evidence for the dominant kernels of the SECONDARY rows (compute_cells_and_kzg_proofs one blob and batch,
# resident batch verification, recovery), run on the GPU box via gpurun from the repo root:
#   bash tools/pmc_rows.sh [rows...]      default rows: cells_wide verify_wide
# Per row: one --kernel-trace --stats pass, then SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, an SQ set), never
# combined with --stats or other trace domains.  tools/summarize_pmc_rows.py turns the directories into
# profiles/<prefix>_pmc_<row>.json.
export TMPDIR=/tmp
O=gpurun_out/pmc_rows
ROWS=${@:-cells_wide verify_wide}
mkdir -p $O
for row in $ROWS; do
  rm -rf $O/$row && mkdir -p $O/$row
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$row/stats -- python tools/pmc_workload.py $row > $O/$row/stats.json 2> $O/$row/stats.err
  cat $O/$row/stats.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$row/pmc_$c -- python tools/pmc_workload.py $row > /dev/null 2> $O/$row/pmc_$c.err
  done
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/$row/pmc_SQ -- python tools/pmc_workload.py $row > /dev/null 2> $O/$row/pmc_SQ.err
  # the raw per-dispatch csv files are large: keep the summaries only
  python tools/summarize_pmc_rows.py $O/$row > $O/$row.json
  rm -rf $O/$row/pmc_*/*/*_counter_collection.csv $O/$row/*/*/*kernel_trace.csv $O/$row/*/*/*agent_info.csv
  head -c 600 $O/$row.json; echo
done
