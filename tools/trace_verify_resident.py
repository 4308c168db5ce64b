#!/usr/bin/env python3
"""Phase trace (CKZG_HIP_TRACE) of the resident blob-batch verification, ckzg_hip_verify_blob_kzg_proof_batch_device:
    python tools/trace_verify_resident.py [n ...]      default 512 4096; default tables
Prints the library's own phase marks (stderr) for the third call of each size and the wall time of five calls."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 4096]
    import torch
    import __graft_entry__ as ge
    mod = ge.load_package()
    hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 10, "proof_wbits": 8, "fk20_wbits": 0})
    L = bench.Lib(hip.lib)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0xC4B64844)
    blobs = torch.randint(0, 256, (8, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
    blobs[:, :, 0] = 0
    ub = [bytes(b) for b in blobs.cpu().numpy().reshape(8, -1)]
    cm = [hip.blob_to_kzg_commitment(b) for b in ub]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(ub, cm)]
    sp = C.addressof(hip.s)
    hb = bench.HipBuffers(torch, dev)
    ok = C.c_bool(False)
    for n in sizes:
        bb = b"".join(ub[i % 8] for i in range(n))
        cc = b"".join(cm[i % 8] for i in range(n))
        pp = b"".join(pr[i % 8] for i in range(n))
        dev_t = [hb.device(x) for x in (bb, cc, pp)]
        dptr = [C.c_void_p(t.data_ptr()) for t in dev_t]
        for _ in range(2):
            L.verify_blobs_dev(C.byref(ok), dptr[0], dptr[1], dptr[2], n, sp)
        os.environ["CKZG_HIP_TRACE"] = "1"
        sys.stderr.write("== n = %d\n" % n)
        rc = L.verify_blobs_dev(C.byref(ok), dptr[0], dptr[1], dptr[2], n, sp)
        del os.environ["CKZG_HIP_TRACE"]
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            rc = L.verify_blobs_dev(C.byref(ok), dptr[0], dptr[1], dptr[2], n, sp)
            ts.append((time.perf_counter() - t) * 1e3)
        print("n=%d rc=%d ok=%s wall ms %s kernel_ms total %.3f first %.3f sums %.3f" %
              (n, rc, ok.value, " ".join("%.3f" % t for t in ts), L.kms(sp, 3), L.kms(sp, 0), L.kms(sp, 2)))
        sys.stdout.flush()
    hip.close()


if __name__ == "__main__":
    main()
