#!/bin/bash
# (sweeps tuning constants: needs the A/B build -- bash tools/build_variant.sh ab -DCKZG_AB; export CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so)
# verify_cell_kzg_proof_batch / verify_blob_kzg_proof_batch latencies for the default build and variants named on
# the command line, inside one gpurun call (default tables).
export TMPDIR=/tmp
rm -f gpurun_out/r2_verify_ab.log
for v in "" "$@"; do
  [ -f c-kzg-4844_amd/libckzg_hip$v.so ] || continue
  echo "== libckzg_hip$v.so" >> gpurun_out/r2_verify_ab.log
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 200 python tools/bench_verify_cells.py 16 64 128 200 1024 2>/dev/null | tr '\n' ' ' >> gpurun_out/r2_verify_ab.log
  echo >> gpurun_out/r2_verify_ab.log
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 200 python tools/bench_verify_small.py 9 16 40 64 512 2>/dev/null | tr '\n' ' ' >> gpurun_out/r2_verify_ab.log
  echo >> gpurun_out/r2_verify_ab.log
done
cat gpurun_out/r2_verify_ab.log
