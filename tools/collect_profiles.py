"""Copy the summaries tools/profile_bench.sh left under gpurun_out/prof into profiles/ (tracked).
usage: python tools/collect_profiles.py <prefix>      e.g. r01_c16"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
pre = sys.argv[1]
s = json.load(open(os.path.join(src, "summary.json")))
json.dump({k: s[k] for k in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE")},
          open(os.path.join(dst, pre + "_pmc_k_msm_accumulate.json"), "w"), indent=1)
json.dump({"pmc_SQ": s["pmc_SQ"],
           "timed_kernel_trace": {k: s["timed"][k] for k in ("kernel_ms", "kernel_ms_mean", "launches")}},
          open(os.path.join(dst, pre + "_pmc_sq_k_msm_accumulate.json"), "w"), indent=1)
last = lambda f: json.loads(open(os.path.join(src, f)).read().strip().splitlines()[-1])
json.dump({"default_command_python_bench_py": last("bench_default.json"),
           "profiled_timed_only_run": last("bench_timed.json")},
          open(os.path.join(dst, pre + "_bench_lines.json"), "w"), indent=1)
shutil.copy(sorted(glob.glob(os.path.join(src, "timed", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1],
            os.path.join(dst, pre + "_bench_timed_only_kernel_stats.csv"))
print("ok")
