for n in 8 16 32; do
  echo "== blobs n=$n table off"; CKZG_HIP_VERIFY_TABLE_WBITS=0 timeout 100 python tools/bench_verify_forms.py $n 9 2>/dev/null | cut -c1-120
  echo "== blobs n=$n table on (min 4)"; CKZG_HIP_VERIFY_TABLE_MIN=4 timeout 100 python tools/bench_verify_forms.py $n 9 2>/dev/null | cut -c1-120
done
echo "== cells min 128"; CKZG_HIP_VERIFY_CELL_TABLE_MIN=128 timeout 100 python tools/bench_verify_cells.py 128 256 384 | tail -3
echo "== cells default"; timeout 100 python tools/bench_verify_cells.py 128 256 384 512 | tail -4
