import sys, os, hashlib, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mod = ge.load_package()
hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 8, "proof_wbits": 0, "fk20_wbits": int(sys.argv[1]) if len(sys.argv) > 1 else 12})
n = 2048
b = b"".join(b"\x00" + hashlib.sha256(b"f%d" % j).digest()[:31] for j in range(4096))
f = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
cells = C.create_string_buffer(n * 128 * 2048); proofs = C.create_string_buffer(n * 128 * 48); st = C.create_string_buffer(n)
blobs = b * n
f(cells, proofs, st, blobs, n, C.addressof(hip.s))
t = time.perf_counter(); f(cells, proofs, st, blobs, n, C.addressof(hip.s)); print("batch %d: %.1f ms" % (n, (time.perf_counter() - t) * 1e3))
