#!/usr/bin/env python3
"""Fill the HBM for real, leaving LEAVE_MB free, and run verify_blob_kzg_proof_batch (n = 512) on a fresh
KZGSettings: one JSON line per level {leave_mb, free_mb, rc, ok, rc_bad, ok_bad}.  Run in a process of its own
(tests/test_gpu_round4.py does): a device with NO memory left makes the HSA runtime abort the process when a queue
needs scratch, which no library can catch.
  python tools/oom_fallback_probe.py 900 440 400 360"""
import ctypes as C
import hashlib
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def blob(i):
    return b"".join(b"\x00" + hashlib.sha256(b"oom%d|%d" % (i, j)).digest()[:31] for j in range(4096))


def main():
    import torch
    mod = ge.load_package()
    levels = [int(x) for x in sys.argv[1:]] or [900, 440, 400, 360]
    n = 512
    small = {"commit_wbits": 6, "proof_wbits": 0, "fk20_wbits": 4}
    k = mod.Kzg(mod.HIP_SO, options=small)
    blobs = [blob(i) for i in range(4)]
    cm = [k.blob_to_kzg_commitment(b) for b in blobs]
    pr = [k.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    k.close()
    bb = b"".join(blobs[i % 4] for i in range(n))
    cc = b"".join(cm[i % 4] for i in range(n))
    pp = b"".join(pr[i % 4] for i in range(n))
    bad_pp = pp[:48 * 77] + pr[(77 + 1) % 4] + pp[48 * 78:]   # a proof of another blob at position 77
    for leave_mb in levels:
        k = mod.Kzg(mod.HIP_SO, options=small)
        f = k.lib.verify_blob_kzg_proof_batch
        f.restype = C.c_int
        ok = C.c_bool(False)
        hog = []
        for _ in range(8):   # memory released a moment ago comes back in stages (the driver scrubs it first)
            torch.cuda.synchronize()
            spare = torch.cuda.mem_get_info()[0] - leave_mb * (1 << 20)
            if spare < (8 << 20):
                break
            hog.append(torch.empty((spare,), dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        free_mb = torch.cuda.mem_get_info()[0] >> 20
        print(json.dumps({"leave_mb": leave_mb, "free_mb": free_mb, "stage": "filled"}), flush=True)
        rc = f(C.byref(ok), bb, cc, pp, C.c_uint64(n), k.sp)
        verdict = bool(ok.value)
        rc_bad = f(C.byref(ok), bb, cc, bad_pp, C.c_uint64(n), k.sp)
        print(json.dumps({"leave_mb": leave_mb, "free_mb": free_mb, "rc": rc, "ok": verdict, "rc_bad": rc_bad,
                          "ok_bad": bool(ok.value), "free_after_mb": torch.cuda.mem_get_info()[0] >> 20}), flush=True)
        del hog
        torch.cuda.empty_cache()
        k.close()


if __name__ == "__main__":
    main()
