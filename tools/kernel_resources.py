#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel in the PRODUCT, read from the gfx950 code objects inside the
object files `make` produced (c-kzg-4844_amd/csrc/*.o) -- the same bytes that are linked into libckzg_hip.so, so
the per-file flags of the Makefile (e.g. -DCKZG_F28_ASM_BLOCKS for msm.o / fk20.o) are in by construction.

On gfx950 .vgpr_count is the unified total (architectural + accumulation registers, of which .agpr_count are the
latter): waves/SIMD = floor(512 / roundup(vgpr_count, 8)), max 8.

    python tools/kernel_resources.py            > profiles/rNN_kernel_resources.txt     (no GPU needed)
    python tools/kernel_resources.py --json     machine-readable (tests/test_kernel_resources.py)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "c-kzg-4844_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FILES = ["msm", "ntt", "fk20", "verify", "pippenger", "ckzg_api", "ckzg_api2", "device_ctx"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    res = []
    for n in out.splitlines():
        n = re.sub(r"^void ", "", n)
        n = n.replace("ckzg::dev::", "").replace("ckzg::", "")
        # keep template arguments, drop the parameter list
        depth, cut = 0, len(n)
        for i, c in enumerate(n):
            if c == "<":
                depth += 1
            elif c == ">":
                depth -= 1
            elif c == "(" and depth == 0:
                cut = i
                break
        res.append(n[:cut])
    return res


def kernels_of(obj):
    """[(kernel, vgpr, agpr, sgpr, lds, scratch)] of every gfx950 code object in a host object's (or shared library's)
    .hip_fatbin section -- a library holds one offload bundle per translation unit, back to back."""
    import yaml
    out = []
    with tempfile.TemporaryDirectory() as t:
        fat = os.path.join(t, "k.fatbin")
        # (an explicit output operand: without one llvm-objcopy rewrites `obj` in place, and the product's object files
        # came out newer than libckzg_hip.so -- the next `make` relinked for nothing)
        r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(t, "discard.o")],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, a in enumerate(starts):
            part, co = os.path.join(t, "p%d.bundle" % i), os.path.join(t, "p%d.co" % i)
            open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True, check=True)
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            if "amdhsa.kernels" not in notes:
                continue
            doc = notes[notes.index("---"):]
            doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
            rows = yaml.safe_load(doc).get("amdhsa.kernels") or []
            names = demangle([r[".name"] for r in rows])
            out += [(n, r[".vgpr_count"], r.get(".agpr_count", 0), r.get(".sgpr_count", 0),
                     r.get(".group_segment_fixed_size", 0), r.get(".private_segment_fixed_size", 0)) for n, r in zip(names, rows)]
    return out


def waves(v):
    t = max(8, (v + 7) // 8 * 8)
    return min(8, 512 // t)


def collect():
    table = {}
    for b in FILES:
        obj = os.path.join(CSRC, b + ".o")
        if not os.path.exists(obj):
            raise SystemExit("kernel_resources: %s missing -- run `make -C c-kzg-4844_amd` first" % obj)
        for (n, v, a, s, l, p) in kernels_of(obj):
            table["%s:%s" % (b, n)] = {"vgpr": v, "agpr": a, "sgpr": s, "lds": l, "scratch": p, "waves": waves(v)}
    return table


def main():
    t = collect()
    if "--json" in sys.argv:
        print(json.dumps(t, indent=1, sort_keys=True))
        return
    print("# from the product's object files (c-kzg-4844_amd/csrc/*.o as built by the Makefile)")
    print("%-12s %-64s %6s %6s %6s %8s %8s %6s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "lds_B", "scratch", "waves"))
    for k in sorted(t):
        f, n = k.split(":", 1)
        r = t[k]
        print("%-12s %-64s %6d %6d %6d %8d %8d %6d" % (f, n[:64], r["vgpr"], r["agpr"], r["sgpr"], r["lds"], r["scratch"], r["waves"]))


if __name__ == "__main__":
    main()
