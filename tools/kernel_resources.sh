#!/bin/bash
# Register / LDS / scratch budget of every kernel in the product, read from the gfx950 code objects
# themselves (the .note metadata the loader uses), so that occupancy claims do not depend on how a
# profiler labels its columns.  On gfx950 .vgpr_count is the unified total (architectural + accumulation registers, of which .agpr_count are the latter): waves/SIMD = floor(512 / roundup(vgpr_count, 8)), max 8.
# Usage: bash tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt   (no GPU needed)
set -e
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
cd "$(dirname "$0")/../c-kzg-4844_amd"
printf "%-14s %-60s %6s %6s %6s %8s %8s %6s\n" file kernel vgpr agpr sgpr lds_B scratch waves
for f in csrc/msm.hip csrc/ntt.hip csrc/fk20.hip csrc/verify.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Wno-pass-failed -c $f -o $T/$b.o 2>/dev/null
  $LLVM/clang-offload-bundler --unbundle --type=o --input=$T/$b.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$b.co
  $LLVM/llvm-readelf --notes $T/$b.co | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|group_segment_fixed|private_segment_fixed" \
    | paste - - - - - - | while read -r line; do
      name=$(echo "$line" | sed -E 's/.*\.name:[ ]+([^ \t]+).*/\1/' | c++filt | sed -E 's/\(.*//; s/^void //; s/ckzg::dev:://')
      v=$(echo "$line" | sed -E 's/.*\.vgpr_count:[ ]+([0-9]+).*/\1/'); a=$(echo "$line" | sed -E 's/.*\.agpr_count:[ ]+([0-9]+).*/\1/')
      s=$(echo "$line" | sed -E 's/.*\.sgpr_count:[ ]+([0-9]+).*/\1/'); l=$(echo "$line" | sed -E 's/.*group_segment_fixed_size:[ ]+([0-9]+).*/\1/')
      p=$(echo "$line" | sed -E 's/.*private_segment_fixed_size:[ ]+([0-9]+).*/\1/')
      t=$(( (v + 7) / 8 * 8 )); [ $t -lt 8 ] && t=8; w=$(( 512 / t )); [ $w -gt 8 ] && w=8
      printf "%-14s %-60s %6s %6s %6s %8s %8s %6s\n" $b "${name:0:60}" $v $a $s $l $p $w
    done
done
rm -rf $T
