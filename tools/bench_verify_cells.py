"""verify_cell_kzg_proof_batch timing at the C-ABI for n cells drawn from a few blobs."""
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
nb = 8
blobs = [rand_blob(90, i) for i in range(nb)]
cs = [hip.blob_to_kzg_commitment(b) for b in blobs]
cp = [hip.compute_cells_and_kzg_proofs(b) for b in blobs]
fv = hip.lib.verify_cell_kzg_proof_batch
fv.restype = C.c_int
fv.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
for n in [int(x) for x in sys.argv[1:]] or [16, 128, 1024]:
    rows = [(i // 128) % nb for i in range(n)]
    cols = [i % 128 for i in range(n)]
    cc = b"".join(cs[r] for r in rows)
    idx = (C.c_uint64 * n)(*cols)
    cells = b"".join(cp[r][0][c] for r, c in zip(rows, cols))
    proofs = b"".join(cp[r][1][c] for r, c in zip(rows, cols))
    ok = C.c_bool(False)
    fv(C.byref(ok), cc, idx, cells, proofs, n, C.addressof(hip.s))
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        rc = fv(C.byref(ok), cc, idx, cells, proofs, n, C.addressof(hip.s))
        best = min(best, time.perf_counter() - t)
    print("verify_cell_kzg_proof_batch n=%d: %.2f ms -> %.0f cells/s (rc=%d ok=%s)" % (n, best * 1e3, n / best, rc, ok.value))
