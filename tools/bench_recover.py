"""Time recover_cells_and_kzg_proofs: single call vs the batch entry point (same missing columns
for every row).  Host buffers in and out (the C-ABI boundary), so PCIe is included."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from test_gpu_commitment import rand_blob  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
api = pkg.Kzg(options={"fk20_wbits": 12, "proof_wbits": 13})
base = [rand_blob(70, i) for i in range(8)]
full = [api.compute_cells_and_kzg_proofs(b) for b in base]
keep = list(range(0, 128, 2))
rows = [[full[b % 8][0][i] for i in keep] for b in range(nb)]
t = time.perf_counter()
for _ in range(5):
    api.recover_cells_and_kzg_proofs(keep, rows[0])
single = (time.perf_counter() - t) / 5
f = api.lib.ckzg_hip_recover_cells_and_kzg_proofs_batch
f.restype = C.c_int
data = b"".join(b"".join(r) for r in rows)
idx = (C.c_uint64 * len(keep))(*keep)
rc = C.create_string_buffer(nb * 128 * 2048)
rp = C.create_string_buffer(nb * 128 * 48)
for want_cells in (True, False):
    f(rc if want_cells else None, rp, None, idx, data, C.c_uint64(len(keep)), C.c_uint64(nb), api.sp)
    t = time.perf_counter()
    assert f(rc if want_cells else None, rp, None, idx, data, C.c_uint64(len(keep)), C.c_uint64(nb), api.sp) == 0
    dt = time.perf_counter() - t
    print("batch %d rows, cells_out=%s: %.2f ms total, %.3f ms/row, %.0f rows/s" %
          (nb, want_cells, dt * 1e3, dt * 1e3 / nb, nb / dt))
assert rp.raw[:128 * 48] == b"".join(full[0][1])
print("single call: %.3f ms" % (single * 1e3))
