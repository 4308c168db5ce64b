"""compute_blob_kzg_proof: the single-call entry point vs the batch entry point with n = 1."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from test_gpu_commitment import rand_blob
mod = ge.load_package()
hip = mod.Kzg(mod.HIP_SO)
b = rand_blob(95, 0)
c = hip.blob_to_kzg_commitment(b)
p0 = hip.compute_blob_kzg_proof(b, c)
f1 = hip.lib.compute_blob_kzg_proof; f1.restype = C.c_int
fb = hip.lib.ckzg_hip_compute_blob_kzg_proof_batch; fb.restype = C.c_int
out = C.create_string_buffer(48); st = C.create_string_buffer(1)
for name, call in (("single", lambda: f1(out, b, c, hip.sp)), ("batch n=1", lambda: fb(out, st, b, c, C.c_uint64(1), hip.sp))):
    call(); best = 1e9
    for _ in range(10):
        t = time.perf_counter(); rc = call(); best = min(best, time.perf_counter() - t)
    print("%s: %.3f ms rc=%d same=%s" % (name, best * 1e3, rc, out.raw == p0))
