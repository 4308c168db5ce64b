# (historical: CKZG_HIP_VERIFY_TABLE_RESIDENT was an A/B knob of commits b2ebe19..81b6cba; the table is now the default in the resident form)
for n in 1024 1536 2048 4096; do
  echo "== blobs n=$n table off"; CKZG_HIP_VERIFY_TABLE_WBITS=0 timeout 100 python tools/bench_verify_forms.py $n 7 2>/dev/null | cut -c1-230
  echo "== blobs n=$n table on (min 1024, resident too)"; CKZG_HIP_VERIFY_TABLE_MIN=1024 CKZG_HIP_VERIFY_TABLE_RESIDENT=1 timeout 100 python tools/bench_verify_forms.py $n 7 2>/dev/null | cut -c1-230
done
echo "== cells table off"; CKZG_HIP_VERIFY_TABLE_WBITS=0 timeout 100 python tools/bench_verify_cells.py 1024 2048 3072 4096 6144 | tail -5
echo "== cells table on (min 1024)"; CKZG_HIP_VERIFY_CELL_TABLE_MIN=1024 timeout 100 python tools/bench_verify_cells.py 1024 2048 3072 4096 6144 | tail -5
