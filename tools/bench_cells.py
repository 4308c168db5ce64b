#!/usr/bin/env python3
"""Secondary measurement: compute_cells_and_kzg_proofs latency (1 blob) and batch throughput."""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    mod = ge.load_package()
    fk = int(os.environ.get("FK20_WBITS", "8"))
    t0 = time.perf_counter()
    pw = int(os.environ.get("PROOF_WBITS", "13"))
    hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 8, "fk20_wbits": fk, "proof_wbits": pw,
                                       "direct_max": int(os.environ.get("DIRECT_MAX", "32"))})
    print("load_trusted_setup_file: %.2f s (fk20_wbits=%d)" % (time.perf_counter() - t0, fk))
    blob = b"".join(b"\x00" + hashlib.sha256(b"c%d" % j).digest()[:31] for j in range(4096))
    hip.compute_cells_and_kzg_proofs(blob)
    for what, wc, wp in (("cells+proofs", True, True), ("cells only", True, False), ("proofs only", False, True)):
        ts = []
        for _ in range(10):
            t = time.perf_counter()
            hip.compute_cells_and_kzg_proofs(blob, wc, wp)
            ts.append(time.perf_counter() - t)
        ts.sort()
        print("1 blob %-13s median %.2f ms  min %.2f ms" % (what, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
    f = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    for n in (4, 16, 64, 512, 2048, 4096):
        blobs = blob * n
        cells = C.create_string_buffer(n * 128 * 2048)
        proofs = C.create_string_buffer(n * 128 * 48)
        st = C.create_string_buffer(n)
        f(cells, proofs, st, blobs, n, C.addressof(hip.s))
        t = time.perf_counter()
        rc = f(cells, proofs, st, blobs, n, C.addressof(hip.s))
        dt = time.perf_counter() - t
        print("batch %4d: %.1f ms  -> %.1f blobs/s (host pointers, rc=%d)" % (n, dt * 1e3, n / dt, rc))
    hip.close()


if __name__ == "__main__":
    main()
