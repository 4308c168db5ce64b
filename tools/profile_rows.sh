#!/bin/bash
# Kernel-level breakdown of the secondary rows (compute_cells_and_kzg_proofs, verify_*_batch, recover),
# one rocprofv3 --kernel-trace --stats run per row; run on the GPU box via gpurun from the repo root.
# The kernel_stats.csv of each run is copied to profiles/<prefix>_<row>_kernel_stats.csv by
# tools/collect_rows.py.
export TMPDIR=/tmp
O=gpurun_out/prof_rows
rm -rf $O && mkdir -p $O
for spec in "cells wide" "cells default" "verify default" "verify wide"; do
  set -- $spec
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1_$2 -- python tools/row_driver.py $1 $2 > $O/$1_$2.json 2> $O/$1_$2.err
  tail -c 600 $O/$1_$2.json
done
