#!/bin/bash
# Memory-side counters of k_eval_tree at a batch past the Infinity Cache (resident verification, N blobs; default 8192)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-8192}
O=$R/gpurun_out/r6/pmc_eval_mem
rm -rf $O && mkdir -p $O
cd /tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum" \
           "TCC_EA_RDREQ_DRAM_sum TCC_EA_RDREQ_LEVEL_sum TCC_BUBBLE_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_e
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_e -- python $R/tools/trace_verify_resident.py $N > /dev/null 2> $O/set$i.err
  f=$(find /tmp/pmc_e -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" > $O/set$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_eval_tree" not in k:
        continue
    name = k.split("(")[0] + " grid=" + r["Grid_Size"]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in cs.items():
        print("   %-40s n=%d mean=%.1f" % (c, len(v), sum(v) / len(v)))
PY
  cat $O/set$i.txt
  else tail -3 $O/set$i.err; fi
done
