#!/usr/bin/env python3
"""Latency of the host-pointer batch entry points at the batch sizes the combiner launches (1..256 units), from
page-locked memory (what the combiner hands them), with the kernel-family times of ckzg_hip_last_kernel_ms.
  python tools/bench_small_batches.py [--wide] [--ops commit,cells]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wide", action="store_true")
    ap.add_argument("--ops", default="commit,cells")
    ap.add_argument("--sizes", default="1,2,4,8,16,32,64,128,256")
    args = ap.parse_args()
    import torch
    mod = ge.load_package()
    tables = {"commit_wbits": 16, "proof_wbits": 16, "fk20_wbits": 13} if args.wide else {}
    k = mod.Kzg(mod.HIP_SO, options=tables)
    sp = C.addressof(k.s)
    lib = k.lib
    kms = lib.ckzg_hip_last_kernel_ms
    kms.restype = C.c_double
    kms.argtypes = [C.c_void_p, C.c_int]
    nmax = 256
    g = torch.Generator()
    g.manual_seed(7)
    blobs = torch.randint(0, 256, (nmax, 4096, 32), dtype=torch.uint8, generator=g)
    blobs[:, :, 0] = 0
    blobs = blobs.pin_memory()
    out = torch.empty((nmax * (128 * 2048 + 128 * 48 + 1),), dtype=torch.uint8).pin_memory()
    st = torch.empty((nmax,), dtype=torch.uint8)
    p = C.c_void_p
    for op in args.ops.split(","):
        for n in [int(x) for x in args.sizes.split(",")]:
            if op == "commit":
                f = lib.ckzg_hip_blob_to_kzg_commitment_batch
                f.restype = C.c_int
                f.argtypes = [p, p, p, C.c_uint64, p]
                call = lambda: f(out.data_ptr(), st.data_ptr(), blobs.data_ptr(), n, sp)  # noqa: E731
            else:
                f = lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
                f.restype = C.c_int
                f.argtypes = [p, p, p, p, C.c_uint64, p]
                call = lambda: f(out.data_ptr(), out.data_ptr() + n * 128 * 2048, st.data_ptr(), blobs.data_ptr(), n, sp)  # noqa: E731
            assert call() == 0
            ts = []
            reps = 20 if op == "commit" else 6
            for _ in range(reps):
                t = time.perf_counter()
                rc = call()
                ts.append(time.perf_counter() - t)
            assert rc == 0
            ts.sort()
            med = ts[len(ts) // 2]
            print(json.dumps({"op": op, "wide": args.wide, "n": n, "ms": round(med * 1e3, 3), "units_per_s": round(n / med, 1),
                              "kernel_ms": [round(kms(sp, w), 3) for w in range(5)]}), flush=True)
    k.close()


if __name__ == "__main__":
    main()
