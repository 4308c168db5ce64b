"""verify_blob_kzg_proof_batch latency for small n (host-side scalar multiplications up to
SMALL_VERIFY_N, GPU validation/lincomb above), timed at the C-ABI."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from test_gpu_commitment import rand_blob  # noqa: E402

mod = ge.load_package()
hip = mod.Kzg(mod.HIP_SO)
uniq = [rand_blob(80, i) for i in range(4)]
cs = [hip.blob_to_kzg_commitment(b) for b in uniq]
ps = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(uniq, cs)]
fv = hip.lib.verify_blob_kzg_proof_batch
fv.restype = C.c_int
fv.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
for n in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 9, 12, 16, 32]:
    bb = b"".join(uniq[i % 4] for i in range(n))
    cc = b"".join(cs[i % 4] for i in range(n))
    pp = b"".join(ps[i % 4] for i in range(n))
    ok = C.c_bool(False)
    fv(C.byref(ok), bb, cc, pp, n, C.addressof(hip.s))
    best = 1e9
    for _ in range(3 if n > 256 else 5):
        t = time.perf_counter()
        rc = fv(C.byref(ok), bb, cc, pp, n, C.addressof(hip.s))
        best = min(best, time.perf_counter() - t)
    print("n=%d: %.2f ms (rc=%d ok=%s)" % (n, best * 1e3, rc, ok.value))
