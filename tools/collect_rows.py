"""Copy the per-row kernel statistics tools/profile_rows.sh left under gpurun_out/prof_rows into profiles/.
usage: python tools/collect_rows.py <prefix>      e.g. r02"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_rows")
pre = sys.argv[1]
for d in sorted(glob.glob(os.path.join(src, "*/"))):
    row = os.path.basename(os.path.normpath(d))
    # gpurun merges into the local directory: an earlier call's files may still be there -- take the newest
    st = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (pre, row)))
    js = os.path.join(src, row + ".json")
    if os.path.exists(js):
        shutil.copy(js, os.path.join(ROOT, "profiles", "%s_%s_row.json" % (pre, row)))
print("ok")
