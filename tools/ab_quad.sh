#!/bin/bash
# (sweeps tuning constants: needs the A/B build -- bash tools/build_variant.sh ab -DCKZG_AB; export CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so)
# A/B of the one-lane and the four-lane (DPP quad, g1_quad.hpp) forms of the latency-bound G1 work:
# verify_cell_kzg_proof_batch (ladders + subgroup test; CKZG_HIP_QUAD_MAX=0 disables the quad forms) and the
# small-batch FK20 path (G1 FFT twiddles; CKZG_HIP_QUAD_FFT_MAX=0 disables).  Bounded steps.
export TMPDIR=/tmp
for qm in 0 8192; do
  echo "== CKZG_HIP_QUAD_MAX=$qm"
  CKZG_HIP_QUAD_MAX=$qm timeout 120 python tools/bench_verify_cells.py 16 128 1024 2048 4096 2>&1 | grep verify_cell
  CKZG_HIP_QUAD_MAX=$qm NVERIFY=512 timeout 120 python tools/bench_verify.py 2>&1 | grep -i "verify" | head -8
done
for qf in 0 256; do
  echo "== CKZG_HIP_QUAD_FFT_MAX=$qf"
  for tw in "8 8" "16 13"; do
    CKZG_HIP_QUAD_FFT_MAX=$qf timeout 200 python tools/bench_direct_vs_fk20.py $tw 2>&1 | grep direct_max
  done
done
