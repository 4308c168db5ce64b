#!/bin/bash
# (sweeps tuning constants: needs the A/B build -- bash tools/build_variant.sh ab -DCKZG_AB; export CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so)
# Small-batch FK20 latency (compute_cells_and_kzg_proofs through FK20, 1..64 blobs, default 8-bit table) for the
# default build and variants named on the command line, inside one gpurun call.
export TMPDIR=/tmp
rm -f gpurun_out/r2_small_ab.log
for v in "" "$@"; do
  [ -f c-kzg-4844_amd/libckzg_hip$v.so ] || continue
  echo "== libckzg_hip$v.so" >> gpurun_out/r2_small_ab.log
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 200 python tools/bench_fk20_sizes.py 8 1 4 8 16 24 32 48 64 128 2>/dev/null | tail -1 >> gpurun_out/r2_small_ab.log
done
cat gpurun_out/r2_small_ab.log
