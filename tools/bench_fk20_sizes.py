"""compute_cells_and_kzg_proofs through FK20 (direct_max = 0) for a range of batch sizes, ms per call, host pointers:
where the small-batch forms of the G1 FFT (radix-4 pairs, four-lane ladders) hand over to the throughput forms.
usage: [CKZG_HIP_R4_FFT_MAX=..] [CKZG_HIP_QUAD_FFT_MAX=..] python tools/bench_fk20_sizes.py [fk20_wbits] n1 n2 ..."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from test_gpu_commitment import rand_blob
mod = ge.load_package()
fw = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sizes = [int(x) for x in sys.argv[2:]] or [64, 128, 256, 512, 1024]
k = mod.Kzg(mod.HIP_SO, options={"direct_max": 0, "proof_wbits": 0, "fk20_wbits": fw, "commit_wbits": 8})
f = k.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
f.restype = C.c_int
base = [rand_blob(97, i) for i in range(8)]
out = []
for n in sizes:
    blobs = b"".join(base[i % 8] for i in range(n))
    proofs = C.create_string_buffer(n * 128 * 48); st = C.create_string_buffer(n)
    f(None, proofs, st, blobs, C.c_uint64(n), k.sp)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); rc = f(None, proofs, st, blobs, C.c_uint64(n), k.sp); best = min(best, time.perf_counter() - t)
    out.append("%d:%.1f" % (n, best * 1e3))
print("R4_MAX=%s QUAD_MAX=%s fk20_wbits=%d proofs only  n:ms  %s" % (os.environ.get("CKZG_HIP_R4_FFT_MAX", "dflt"),
      os.environ.get("CKZG_HIP_QUAD_FFT_MAX", "dflt"), fw, "  ".join(out)))
k.close()
