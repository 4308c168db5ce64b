"""Where the low-latency proof path (128 fixed-base MSMs) hands over to FK20: cells+proofs for n blobs with
direct_max = 0 (always FK20) and direct_max = 4096 (always direct), default table widths unless given."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from test_gpu_commitment import rand_blob
mod = ge.load_package()
pw = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
blobs = b"".join(rand_blob(97, i) for i in range(48))
for dm in (0, 4096):
    k = mod.Kzg(mod.HIP_SO, options={"direct_max": dm, "proof_wbits": pw, "fk20_wbits": fw})
    f = k.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    out = []
    for n in [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8, 12, 16, 24, 32, 48]:
        cells = C.create_string_buffer(n * 128 * 2048); proofs = C.create_string_buffer(n * 128 * 48); st = C.create_string_buffer(n)
        f(cells, proofs, st, blobs, C.c_uint64(n), k.sp)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); rc = f(cells, proofs, st, blobs, C.c_uint64(n), k.sp); best = min(best, time.perf_counter() - t)
        out.append("%d:%.1f" % (n, best * 1e3))
    print("direct_max=%d proof_wbits=%d fk20_wbits=%d  n:ms  %s" % (dm, pw, fw, "  ".join(out)))
    k.close()
