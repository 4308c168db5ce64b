#!/bin/bash
# Kernel-level A/B of the polynomial-evaluation kernels (verify.hip: k_eval_barycentric / k_eval_tree) between the
# product and a build of the previous commit (c-kzg-4844_amd/libckzg_hip_base.so), inside one gpurun call:
# rocprofv3 --kernel-trace --stats over the verify rows of bench.py (tools/row_driver.py verify default), twice each,
# alternating.  Output: gpurun_out/r6/eval_ab/<lib>_<pass>_kernel_stats.csv and the rows' own JSON.
export TMPDIR=/tmp
O=gpurun_out/r6/eval_ab
rm -rf $O && mkdir -p $O
for pass in 1 2; do
  for v in _base ""; do
    [ -f c-kzg-4844_amd/libckzg_hip$v.so ] || continue
    d=$O/run${v}_$pass
    CKZG_HIP_SO=$PWD/c-kzg-4844_amd/libckzg_hip$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python tools/row_driver.py verify default > $O/rows${v}_$pass.json 2> $O/rows${v}_$pass.err
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $O/stats${v}_$pass.csv
    rm -rf $d
    echo "== libckzg_hip$v.so pass $pass"
    grep -E 'k_eval|k_sha256' $O/stats${v}_$pass.csv | sed -E 's/\(.*\)"/"/' | cut -d, -f1-4
  done
done
