#!/usr/bin/env python3
"""verify_blob_kzg_proof_batch over n blobs (BASELINE configs[3]: n = 4096) in its three forms:
pageable host pointers, page-locked host pointers (DMA'd in place), inputs resident in HBM
(ckzg_hip_verify_blob_kzg_proof_batch_device).  Prints one JSON line; CKZG_HIP_TRACE=1 adds the per-phase
wall clock of one call of each form on stderr.   usage: python tools/bench_verify_forms.py [n=4096] [runs=5]"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import __graft_entry__ as ge  # noqa: E402


def blob(i):
    return b"".join(b"\x00" + hashlib.sha256(b"v%d|%d" % (i, j)).digest()[:31] for j in range(4096))


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    mod = ge.load_package()
    hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 10, "fk20_wbits": 8, "proof_wbits": 0})
    sp = C.addressof(hip.s)
    uniq = [blob(i) for i in range(8)]
    cm = [hip.blob_to_kzg_commitment(b) for b in uniq]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(uniq, cm)]
    bb = b"".join(uniq[i % 8] for i in range(n))
    cc = b"".join(cm[i % 8] for i in range(n))
    pp = b"".join(pr[i % 8] for i in range(n))
    fv = hip.lib.verify_blob_kzg_proof_batch
    fv.restype = C.c_int
    fv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    fd = hip.lib.ckzg_hip_verify_blob_kzg_proof_batch_device
    fd.restype = C.c_int
    fd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    kms = hip.lib.ckzg_hip_last_kernel_ms
    kms.restype = C.c_double
    kms.argtypes = [C.c_void_p, C.c_int]
    rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    rt.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    ok = C.c_bool(False)

    def timed(f, *a):
        f(C.byref(ok), *a)     # warm-up: arenas, pinned staging
        ts = []
        for _ in range(runs):
            t = time.perf_counter()
            rc = f(C.byref(ok), *a)
            ts.append(time.perf_counter() - t)
            assert rc == 0 and ok.value, (rc, ok.value)
        return median(ts) * 1e3

    out = {"n": n, "runs": runs, "algorithmic_bytes": n * (131072 + 48 + 48)}
    trace = os.environ.pop("CKZG_HIP_TRACE", None)
    out["pageable_ms"] = round(timed(fv, C.cast(C.c_char_p(bb), C.c_void_p), C.cast(C.c_char_p(cc), C.c_void_p),
                                     C.cast(C.c_char_p(pp), C.c_void_p), n, sp), 3)
    pin = C.c_void_p()
    assert rt.hipHostMalloc(C.byref(pin), len(bb), 0) == 0
    C.memmove(pin, bb, len(bb))
    out["pinned_ms"] = round(timed(fv, pin, C.cast(C.c_char_p(cc), C.c_void_p), C.cast(C.c_char_p(pp), C.c_void_p), n, sp), 3)
    d = [C.c_void_p() for _ in range(3)]
    for q, src in zip(d, (bb, cc, pp)):
        assert rt.hipMalloc(C.byref(q), len(src)) == 0
        assert rt.hipMemcpy(q, C.cast(C.c_char_p(src), C.c_void_p), len(src), 1) == 0
    out["resident_ms"] = round(timed(fd, d[0], d[1], d[2], n, sp), 3)
    out["resident_kernel_ms"] = {"total": round(kms(sp, 3), 3), "validate_convert_hash_evaluate": round(kms(sp, 0), 3),
                                 "sums": round(kms(sp, 2), 3)}
    for k in ("pageable", "pinned", "resident"):
        out[k + "_GBps"] = round(out["algorithmic_bytes"] / out[k + "_ms"] / 1e6, 2)
    # a wrong proof must turn every form false
    bad = pp[:48 * 7] + pr[0] + pp[48 * 8:]
    rc = fv(C.byref(ok), pin, C.cast(C.c_char_p(cc), C.c_void_p), C.cast(C.c_char_p(bad), C.c_void_p), n, sp)
    assert rc == 0 and not ok.value
    if trace:
        os.environ["CKZG_HIP_TRACE"] = trace
        sys.stderr.write("-- pageable\n")
        fv(C.byref(ok), C.cast(C.c_char_p(bb), C.c_void_p), C.cast(C.c_char_p(cc), C.c_void_p), C.cast(C.c_char_p(pp), C.c_void_p), n, sp)
        sys.stderr.write("-- pinned\n")
        fv(C.byref(ok), pin, C.cast(C.c_char_p(cc), C.c_void_p), C.cast(C.c_char_p(pp), C.c_void_p), n, sp)
        sys.stderr.write("-- resident\n")
        fd(C.byref(ok), d[0], d[1], d[2], n, sp)
    print(json.dumps(out))
    hip.close()


if __name__ == "__main__":
    main()
