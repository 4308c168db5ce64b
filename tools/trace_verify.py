import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mod = ge.load_package()
hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 8, "proof_wbits": 0})
b = b"".join(b"\x00" + hashlib.sha256(b"t%d" % j).digest()[:31] for j in range(4096))
c = hip.blob_to_kzg_commitment(b); p = hip.compute_blob_kzg_proof(b, c)
z = bytes(31) + b"\x05"
pr, y = hip.compute_kzg_proof(b, z)
for i in range(6):
    t = time.perf_counter(); ok = hip.verify_kzg_proof(c, z, y, pr); print("verify_kzg_proof (host only) %.2f ms %s" % ((time.perf_counter() - t) * 1e3, ok))
os.environ["CKZG_HIP_TRACE"] = "1"
for i in range(3):
    print(hip.verify_blob_kzg_proof(b, c, p))
