#!/usr/bin/env python3
"""Per-kernel summary of one row directory of tools/pmc_rows.sh: for every (kernel, grid size) that takes at least
0.5 % of the row's kernel time, the launches, mean duration (from the --kernel-trace --stats pass) and the mean of
every counter collected in the separate --pmc passes, per launch.  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3
reports them (see MI355X_MICROARCH.md, "HBM": on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced
streaming read -- bench.py doubles it for the streaming kernels and says so).
usage: python tools/summarize_pmc_rows.py gpurun_out/pmc_rows/<row> > <row>.json"""
import csv
import glob
import json
import os
import re
import sys


def rows(d, suffix):
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("ckzg::dev::", "").replace("ckzg::", "")
    return name.strip()


def main():
    d = sys.argv[1]
    kern = {}
    for r in rows(os.path.join(d, "stats"), "kernel_trace.csv"):
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        kern.setdefault(key, {"ms": []})["ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    total = sum(sum(v["ms"]) for v in kern.values()) or 1.0
    counters = {}
    for p in glob.glob(os.path.join(d, "pmc_*")):
        for r in rows(p, "counter_collection.csv"):
            key = (short(r["Kernel_Name"]), r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
            c = counters.setdefault(key, {})
            c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k_src, k_dst in (("VGPR_Count", "vgpr_alloc_units"), ("LDS_Block_Size", "lds_bytes"), ("Scratch_Size", "scratch")):
                if k_src in r:
                    c.setdefault("_" + k_dst, r[k_src])
    out = {"row": os.path.basename(os.path.normpath(d)), "total_kernel_ms_in_trace": round(total, 3), "kernels": []}
    try:   # what the workload printed in its --stats pass: the tables it ran on and its wall-clock numbers
        w = json.load(open(os.path.join(d, "stats.json")))
        out["tables"] = w.get("tables")
        out["workload"] = {k: v for k, v in w.items() if k not in ("tables", "row")}
    except Exception:  # noqa: BLE001
        pass
    for (name, grid, wg), v in sorted(kern.items(), key=lambda kv: -sum(kv[1]["ms"])):
        share = sum(v["ms"]) / total
        if share < float(os.environ.get("PMC_MIN_SHARE", "0.005")):
            continue
        e = {"kernel": name, "grid": grid, "workgroup": wg, "launches": len(v["ms"]),
             "mean_ms": round(sum(v["ms"]) / len(v["ms"]), 4), "share_of_trace": round(share, 4)}
        c = counters.get((name, grid, wg))
        if c:
            e["counters_per_launch"] = {k: round(sum(x) / len(x), 2) for k, x in c.items() if not k.startswith("_")}
            e.update({k[1:]: x for k, x in c.items() if k.startswith("_")})
        out["kernels"].append(e)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
