import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mod = ge.load_package()
t = time.perf_counter()
hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 8, "proof_wbits": 0})
print("load (commit 8, fk20 8, no proof table): %.2f s" % (time.perf_counter() - t))
blobs = [b"".join(b"\x00" + hashlib.sha256(b"t%d|%d" % (i, j)).digest()[:31] for j in range(4096)) for i in range(4)]
cs = [hip.blob_to_kzg_commitment(b) for b in blobs]
ps = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cs)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = [blobs[i % 4] for i in range(n)]; C_ = [cs[i % 4] for i in range(n)]; P = [ps[i % 4] for i in range(n)]
hip.verify_blob_kzg_proof_batch(B, C_, P)
os.environ["CKZG_HIP_TRACE"] = "1"
t = time.perf_counter(); ok = hip.verify_blob_kzg_proof_batch(B, C_, P); print("total %.1f ms ok=%s (includes python-side b''.join of %d MB)" % ((time.perf_counter() - t) * 1e3, ok, n * 131072 >> 20))
