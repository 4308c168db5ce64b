#!/bin/bash
# Build a second copy of the library with extra compiler flags, for A/B runs inside one gpurun call:
#   bash tools/build_variant.sh ab -DCKZG_AB      ->  c-kzg-4844_amd/libckzg_hip_ab.so: the tuning constants of the
#                                                     product (device.hpp: ab_knob) are read from CKZG_HIP_* variables
#   bash tools/build_variant.sh trace -DCKZG_MSM_TRACE   per-wave trace of k_msm_accumulate (tools/msm_trace.py)
# then   CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip_ab.so CKZG_HIP_SMALL_LPV=8 python tools/row_driver.py cells wide ...
# The product itself reads no tuning variable (tools/ab_*.sh, tools/bench_lincomb.sh, tools/bench_fk20_sizes.py need
# the ab build).
set -e
name=$1; shift
cd "$(dirname "$0")/../c-kzg-4844_amd"
mkdir -p build_$name
for f in csrc/ckzg_api.hip csrc/device_ctx.hip csrc/msm.hip csrc/ntt.hip csrc/fk20.hip csrc/verify.hip csrc/pippenger.hip csrc/ckzg_api2.hip; do
  o=build_$name/$(basename $f .hip).o
  # ONLY=<file stem> restricts the extra flags to that translation unit (e.g. ONLY=msm)
  # (a space-separated list is accepted: ONLY="msm fk20")
  stem=$(basename $f .hip)
  if [ -z "$ONLY" ] || [[ " $ONLY " == *" $stem "* ]]; then extra=("$@"); else extra=(); fi
  # the product's Makefile compiles the throughput units with the asm-block Montgomery product; so do variants
  # (NOASM=1 leaves it out, for A/Bs of the product forms themselves)
  if [ -z "$NOASM" ] && { [ "$stem" = msm ] || [ "$stem" = fk20 ]; }; then extra+=(-DCKZG_F28_ASM_BLOCKS); fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed "${extra[@]}" -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -Wl,--version-script=exports.map -o libckzg_hip_$name.so build_$name/*.o
ls -la libckzg_hip_$name.so
