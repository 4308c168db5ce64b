"""Drop-in check of the boundary against the reference's OWN binding glue: bindings/python/ckzg_wrap.c and
bindings/csharp/ckzg_wrap.c are compiled UNCHANGED, from where they lie under /root/reference, against
include/ckzg.h and linked to libckzg_hip.so -- what a maintainer who swaps `ckzg.c + libblst` for this library does
(setup.py:48-56, bindings/csharp/Makefile).  Zero undefined symbols, and the Python extension imports with its 11
methods.  Nothing of the reference is copied or shipped: outputs go to a temporary directory, and the test is
skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys
import sysconfig

import pytest

from conftest import HIP_SO, ROOT

REF = "/root/reference"
PY_WRAP = os.path.join(REF, "bindings", "python", "ckzg_wrap.c")
CS_WRAP = os.path.join(REF, "bindings", "csharp", "ckzg_wrap.c")
PKG = os.path.dirname(HIP_SO)

pytestmark = pytest.mark.skipif(not (os.path.exists(PY_WRAP) and os.path.exists(CS_WRAP)),
                                reason="the reference tree is only present in the build container")


def _link(tmp_path, src, out, extra, no_undefined=True):
    so = str(tmp_path / out)
    cmd = ["gcc", "-O1", "-shared", "-fPIC"] + (["-Wl,--no-undefined"] if no_undefined else []) + \
          ["-I", os.path.join(ROOT, "include")] + extra + \
          ["-o", so, src, "-L", PKG, "-l:" + os.path.basename(HIP_SO), "-Wl,-rpath," + PKG]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    nm = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    lib = subprocess.run(["nm", "-D", "--defined-only", HIP_SO], capture_output=True, text=True).stdout
    ours = {line.split()[-1].split("@")[0] for line in lib.splitlines() if line.strip()}
    # (symbols the C library provides carry a version tag: name@GLIBC_x.y)
    wanted = {line.split()[-1].split("@")[0] for line in nm.splitlines()
              if line.strip() and "@GLIBC" not in line and " w " not in line}
    return so, wanted, ours


def test_csharp_glue_links_unchanged(tmp_path):
    so, wanted, ours = _link(tmp_path, CS_WRAP, "libckzg_cs.so", ["-I", os.path.dirname(CS_WRAP)])
    from_lib = {s for s in wanted if s in ours}
    # the glue calls the public API only (bindings/csharp/ckzg_wrap.c); every such symbol comes from this library
    assert {"load_trusted_setup_file", "free_trusted_setup"} <= from_lib, from_lib


def test_python_extension_links_and_imports(tmp_path):
    inc = sysconfig.get_paths()["include"]
    if not os.path.exists(os.path.join(inc, "Python.h")):
        pytest.skip("Python.h not installed")
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    # (an extension module leaves the interpreter's own symbols open, as setup.py's build does)
    so, wanted, ours = _link(tmp_path, PY_WRAP, "ckzg" + suffix, ["-I", inc], no_undefined=False)
    stray = {s for s in wanted if s not in ours and not s.startswith(("Py", "_Py"))}
    assert not stray, "symbols neither this library nor the interpreter provides: %s" % sorted(stray)
    api = {"blob_to_kzg_commitment", "compute_kzg_proof", "compute_blob_kzg_proof", "verify_kzg_proof",
           "verify_blob_kzg_proof", "verify_blob_kzg_proof_batch", "compute_cells_and_kzg_proofs",
           "recover_cells_and_kzg_proofs", "verify_cell_kzg_proof_batch", "load_trusted_setup_file", "free_trusted_setup"}
    assert api <= (wanted & ours), api - (wanted & ours)
    code = ("import sys; sys.path.insert(0, %r); import ckzg; "
            "names = [n for n in dir(ckzg) if not n.startswith('_')]; print(len(names), ' '.join(sorted(names)))" % str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    count, names = r.stdout.split(None, 1)
    for fn in ("blob_to_kzg_commitment", "compute_cells_and_kzg_proofs", "verify_cell_kzg_proof_batch", "load_trusted_setup"):
        assert fn in names.split(), names
    assert int(count) >= 11
    # without a GPU the load must fail cleanly (an exception of the binding), never fall back to a CPU path
    import torch
    if not torch.cuda.is_available():
        code = ("import sys; sys.path.insert(0, %r); import ckzg\n"
                "try:\n    ckzg.load_trusted_setup(%r, 0)\n    print('LOADED')\nexcept Exception as e:\n    print('REFUSED', type(e).__name__)\n"
                % (str(tmp_path), os.path.join(PKG, "data", "trusted_setup.txt")))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert "REFUSED" in r.stdout, (r.stdout, r.stderr[-500:])
