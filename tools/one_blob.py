import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mod = ge.load_package()
hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 13, "proof_wbits": 13, "fk20_wbits": 8})
b = b"".join(b"\x00" + hashlib.sha256(b"o%d" % j).digest()[:31] for j in range(4096))
for _ in range(3):
    hip.compute_cells_and_kzg_proofs(b)
    hip.blob_to_kzg_commitment(b)
t = time.perf_counter()
for _ in range(20): hip.blob_to_kzg_commitment(b)
print("blob_to_kzg_commitment 1 blob: %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
