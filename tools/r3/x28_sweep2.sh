timeout 900 python -m pytest tests -x -q -m gpu -k "verify or cell or fuzz" 2>&1 | tail -2
for n in 64 256 512 768; do
  echo "== blobs n=$n table off"; CKZG_HIP_VERIFY_TABLE_WBITS=0 timeout 100 python tools/bench_verify_forms.py $n 7 2>/dev/null | cut -c1-120
  echo "== blobs n=$n table on (min 64)"; CKZG_HIP_VERIFY_TABLE_MIN=64 timeout 100 python tools/bench_verify_forms.py $n 7 2>/dev/null | cut -c1-120
done
echo "== cells default"; timeout 100 python tools/bench_verify_cells.py 128 512 768 1024 8192 | tail -5
echo "== cells min 512"; CKZG_HIP_VERIFY_CELL_TABLE_MIN=512 timeout 100 python tools/bench_verify_cells.py 512 768 | tail -2
