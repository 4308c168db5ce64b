#!/usr/bin/env python3
"""Does the work on the side streams (point validation, call-time table) slow the SHA-256 chain of a resident
verification?  Times ckzg_hip_verify_blob_kzg_proof_batch_device at n blobs with the call-time table on and off
(option verify_call_table) and with the compute-unit partition on and off (option verify_cu_partition) and prints the library's device-time split (first = validation + challenges + evaluation).
    python tools/ubench/sha_contention_probe.py [n ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]
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
        dev_t = [hb.device(b"".join(x[i % 8] for i in range(n))) for x in (ub, cm, pr)]
        dptr = [C.c_void_p(t.data_ptr()) for t in dev_t]
        import time
        for part, table in ((0, 1), (1, 1), (0, 0), (1, 0), (0, 1), (1, 1)):
            hip.lib.ckzg_hip_set_option(b"verify_call_table", table)
            hip.lib.ckzg_hip_set_option(b"verify_cu_partition", part)
            first, wall = [], []
            for _ in range(6):
                t = time.perf_counter()
                rc = L.verify_blobs_dev(C.byref(ok), dptr[0], dptr[1], dptr[2], n, sp)
                wall.append((time.perf_counter() - t) * 1e3)
                first.append(L.kms(sp, 0))
            print("n=%d CU partition %s call-time table %s: rc=%d ok=%s first (validation + challenges + evaluation) ms: %s | wall %s" %
                  (n, "on " if part else "off", "on " if table else "off", rc, ok.value, " ".join("%.3f" % x for x in first[1:]),
                   " ".join("%.3f" % x for x in wall[1:])), flush=True)
    hip.close()


if __name__ == "__main__":
    main()
