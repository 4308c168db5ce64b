"""Kernel-level timeline of ONE compute_cells_and_kzg_proofs call (run under rocprofv3 --kernel-trace)."""
import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mod = ge.load_package()
hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 10, "proof_wbits": 13, "fk20_wbits": 8})
b = b"".join(b"\x00" + hashlib.sha256(b"o%d" % j).digest()[:31] for j in range(4096))
for _ in range(3):
    hip.compute_cells_and_kzg_proofs(b)
time.sleep(0.2)
t = time.perf_counter()
hip.compute_cells_and_kzg_proofs(b)
print("one call: %.3f ms" % ((time.perf_counter() - t) * 1e3))
