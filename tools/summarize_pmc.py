"""Summarise rocprofv3 output directories (csv format) for one kernel: per-launch counter values from
*counter_collection.csv and per-launch durations from *kernel_trace.csv.
usage: python tools/summarize_pmc.py <kernel-substring> <dir> [<dir> ...] > summary.json
GRID=<threads> keeps only the launches of that grid size (the bench's 1024-blob steps are 262144 threads; the
one-blob commitments around the load use the same kernel with another shape)."""
import csv
import glob
import json
import os
import sys


def rows(d, suffix):
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


GRID = os.environ.get("GRID")


def main():
    key = sys.argv[1]
    out = {}
    for d in sys.argv[2:]:
        entry = {"counters": {}, "kernel_ms": []}
        for r in rows(d, "counter_collection.csv"):
            if key not in r.get("Kernel_Name", ""):
                continue
            if GRID and r.get("Grid_Size") != GRID:
                continue
            c = entry["counters"].setdefault(r["Counter_Name"], [])
            c.append(float(r["Counter_Value"]))
            for k_src, k_dst in (("VGPR_Count", "vgpr_count"), ("Accum_VGPR_Count", "agpr_count"),
                                 ("LDS_Block_Size", "lds_block_size"), ("Scratch_Size", "scratch_size"),
                                 ("Grid_Size", "grid"), ("Workgroup_Size", "workgroup")):
                if k_src in r:
                    entry[k_dst] = r[k_src]
        for r in rows(d, "kernel_trace.csv"):
            if GRID and r.get("Grid_Size_X", r.get("Grid_Size")) != GRID:
                continue
            if key in r.get("Kernel_Name", ""):
                entry["kernel_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        entry["counter_mean"] = {k: sum(v) / len(v) for k, v in entry["counters"].items() if v}
        if entry["kernel_ms"]:
            entry["kernel_ms_mean"] = sum(entry["kernel_ms"]) / len(entry["kernel_ms"])
            entry["launches"] = len(entry["kernel_ms"])
        out[os.path.basename(os.path.normpath(d))] = entry
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
