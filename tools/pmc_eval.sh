#!/bin/bash
# Counters of the evaluation kernel (verify.hip: k_eval_tree) in a resident 4096-blob verification; separate --pmc passes,
# never combined with --stats.  Output: gpurun_out/r6/pmc_eval/set<i>.txt (mean per launch, per kernel and grid)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6/pmc_eval
rm -rf $O && mkdir -p $O
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_e
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_e -- python $R/tools/trace_verify_resident.py 4096 > /dev/null 2> $O/set$i.err
  f=$(find /tmp/pmc_e -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/set$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_eval_tree" not in k and "k_sha256" not in k:
        continue
    name = k.split("(")[0] + " grid=" + r["Grid_Size"]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in cs.items():
        print("   %-28s n=%d mean=%.1f" % (c, len(v), sum(v) / len(v)))
PY
  cat $O/set$i.txt
done
