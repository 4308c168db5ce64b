export TMPDIR=/tmp; O=gpurun_out/prof_vcells; rm -rf $O; mkdir -p $O
timeout 200 python tools/bench_verify_cells.py 128 1024 8192 2>&1 | tail -3
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python tools/bench_verify_cells.py 8192 > $O/out.txt 2>$O/err.txt
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_vcells/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=4: print(r['Name'][:70].ljust(70), r['Calls'], "%.1f us avg"%(float(r['AverageNs'])/1e3))
PY
