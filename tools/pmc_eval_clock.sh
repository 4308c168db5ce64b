#!/bin/bash
# wave cycles, VALU instructions and GRBM cycles of k_eval_tree together with the dispatch's own duration (same pass)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for N in "$@"; do
  rm -rf /tmp/pmc_e
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_e -- python $R/tools/trace_verify_resident.py $N > /dev/null 2>&1
  f=$(find /tmp/pmc_e -name '*counter_collection.csv' | head -1)
  python3 - "$f" $N <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("N =", sys.argv[2], "columns:", [c for c in rows[0].keys() if "imest" in c or "Dispatch" in c][:6])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "k_eval_tree" not in k and "k_sha256" not in k:
        continue
    name = k.split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r:
        acc[name]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for name, cs in acc.items():
    print(name)
    for c, v in cs.items():
        print("   %-28s n=%d mean=%.1f" % (c, len(v), sum(v) / len(v)))
PY
done
