"""Latency of ONE blob_to_kzg_commitment call (BASELINE configs[0]'s shape): wall clock per call over many calls, for
rocprofv3 --kernel-trace --stats to attribute.  usage: python tools/one_commit.py [commit_wbits=10] [calls=300]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

wbits = int(sys.argv[1]) if len(sys.argv) > 1 else 10
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mod = ge.load_package()
hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": wbits})
hip.lib.ckzg_hip_set_option(b"commit_wbits", 10)
b = b"".join(b"\x00" + hashlib.sha256(b"o%d" % j).digest()[:31] for j in range(4096))
for _ in range(10):
    hip.blob_to_kzg_commitment(b)
best = 1e9
for rep in range(3):
    t = time.perf_counter()
    for _ in range(calls):
        hip.blob_to_kzg_commitment(b)
    best = min(best, (time.perf_counter() - t) / calls)
print("blob_to_kzg_commitment, 1 blob, %d-bit table: %.1f us per call" % (wbits, best * 1e6))
import ctypes as C  # noqa: E402
f = hip.lib.ckzg_hip_last_kernel_ms
f.restype = C.c_double
f.argtypes = [C.c_void_p, C.c_int]
hip.blob_to_kzg_commitment(b)
t = [f(C.addressof(hip.s), i) for i in range(4)]
if min(t) < 0:
    print("  (the call runs as a captured hipGraph: its kernels are timed by rocprofv3 --kernel-trace, not by events)")
else:
    print("  device time of the last call: digits %.1f us, accumulate %.1f us, reduce + finalize %.1f us, total %.1f us" %
          tuple(1e3 * x for x in t))
