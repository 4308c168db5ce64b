#!/bin/bash
# (sweeps tuning constants: needs the A/B build -- bash tools/build_variant.sh ab -DCKZG_AB; export CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so)
# A/B of the two variable-base sum paths inside verify_cell_kzg_proof_batch, one process each (the choice is
# read once per process): CKZG_HIP_LINCOMB=1 per-term GLV ladders (verify.hip), =2 bucket kernels
# (pippenger.hip).  PROFILE=1 adds a rocprofv3 kernel trace of the n=8192 and n=65536 cases.
# (the bucket kernels are in the product since round 6; the A/B build carries them as well)
# Run on the GPU box via gpurun from the repo root; output in gpurun_out/lincomb/.
export TMPDIR=/tmp
export CKZG_HIP_SO=${CKZG_HIP_SO:-c-kzg-4844_amd/libckzg_hip_ab.so}
O=gpurun_out/lincomb
rm -rf $O && mkdir -p $O
for algo in 1 2; do
  echo "== algo $algo" >> $O/summary.txt
  CKZG_HIP_LINCOMB=$algo timeout 200 python tools/bench_verify_cells.py 128 1024 8192 65536 >> $O/summary.txt 2>&1
  if [ -n "$PROFILE" ]; then
    CKZG_HIP_LINCOMB=$algo timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$algo -- python tools/bench_verify_cells.py 8192 65536 > /dev/null 2> $O/prof$algo.err
    f=$(find $O/prof$algo -name "*kernel_stats.csv" | head -1)
    grep -E "k_lincomb|k_pip|k_subgroup|k_validate" "$f" | cut -c1-200 >> $O/summary.txt
  fi
done
cat $O/summary.txt
