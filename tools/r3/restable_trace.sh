# (historical: CKZG_HIP_VERIFY_TABLE_RESIDENT was an A/B knob of commits b2ebe19..81b6cba; the table is now the default in the resident form)
export TMPDIR=/tmp; O=gpurun_out/prof_restable; rm -rf $O; mkdir -p $O
CKZG_HIP_VERIFY_TABLE_RESIDENT=1 timeout 250 rocprofv3 --kernel-trace --output-format csv -d $O -- python tools/bench_verify_forms.py 4096 2 > $O/out.txt 2>$O/err.txt
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_restable/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
# last k_sha256 launch = last resident call
idx=[i for i,r in enumerate(rows) if 'k_sha256' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sha=[r for r in rows if 'k_sha256' in r['Kernel_Name']][-1]
t0=int(sha['Start_Timestamp'])-3_000_000
sel=[r for r in rows if t0 <= int(r['Start_Timestamp']) <= int(sha['End_Timestamp'])+8_000_000]
for r in sel:
    print("%8.3f %8.3f q%-3s %s grid=%s" % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, r.get('Queue_Id','?'), r['Kernel_Name'][:50], r.get('Grid_Size_X', r.get('Grid_Size','?'))))
PY
