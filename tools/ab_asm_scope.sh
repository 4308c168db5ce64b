export TMPDIR=/tmp
for v in "" _asm2 _asm4; do
  echo "== libckzg_hip$v.so"
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 240 python tools/row_driver.py cells wide 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d['batch_2048']
print('cells 1 blob', d['one_blob']['ms_per_call'], 'batch2048', b['blobs_per_s'], 'k_msm_small', b['k_msm_small_ms'], 'g1_fft', b['g1_fft_ms'])"
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 240 python tools/row_driver.py verify default 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(v.get('ms')) for k,v in d.items() if isinstance(v,dict) and 'ms' in v})"
  CKZG_HIP_SO=c-kzg-4844_amd/libckzg_hip$v.so timeout 100 python tools/bench_direct_vs_fk20.py 8 8 2>&1 | grep "direct_max=0"
done
