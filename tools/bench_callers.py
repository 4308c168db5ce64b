#!/usr/bin/env python3
"""Concurrent one-blob callers of the unchanged ckzg.h API (fanout.py: native threads) at several thread counts,
with coalescing off / on and several numbers of launches in flight.  One JSON line per configuration.
  python tools/bench_callers.py [--wide] [--ops commit,cells] [--threads 1,8,32,128,256] [--active 0,1,2,3]
(active 0 = coalescing off)"""
import argparse
import hashlib
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def blob(i):
    seed = hashlib.sha256(b"callers%d" % i).digest()
    out = bytearray()
    for j in range(0, 4096, 8):   # 8 field elements per hash call (speed): 31 bytes each from a 256-byte stream
        s = b"".join(hashlib.sha256(seed + bytes([k]) + j.to_bytes(2, "big")).digest() for k in range(8))
        for k in range(8):
            out += b"\x00" + s[31 * k:31 * k + 31]
    return bytes(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wide", action="store_true")
    ap.add_argument("--ops", default="commit,cells")
    ap.add_argument("--threads", default="1,8,32,128,256")
    ap.add_argument("--active", default="0,2")
    ap.add_argument("--seconds", type=float, default=0.5)
    args = ap.parse_args()
    mod = ge.load_package()
    fo = mod.fanout
    blobs = [blob(i) for i in range(32)]
    tables = {"commit_wbits": 16, "proof_wbits": 16, "fk20_wbits": 13} if args.wide else {}
    for act in [int(x) for x in args.active.split(",")]:
        opts = dict(tables, coalesce=0 if act == 0 else 1, coalesce_active=max(act, 1))
        k = mod.Kzg(mod.HIP_SO, options=opts)
        try:
            for op_name in args.ops.split(","):
                op, idx = {"commit": (fo.OP_COMMIT, 0), "cells": (fo.OP_CELLS_PROOFS, 3), "proof": (fo.OP_BLOB_PROOF, 4),
                           "verify": (fo.OP_VERIFY_BLOB, 6), "recover": (fo.OP_RECOVER, 5)}[op_name]
                aux, aux_n, rec_in = None, 0, None
                if op == fo.OP_RECOVER:   # every caller holds the even columns of its blob (the same index set: one key)
                    import struct
                    keep = list(range(0, 128, 2))
                    rec_in = [b"".join(k.compute_cells(b)[i] for i in keep) for b in blobs[:8]]
                    aux, aux_n = struct.pack("<64Q", *keep), 64
                for nt in [int(x) for x in args.threads.split(",")]:
                    ins = [blobs[t % 32] for t in range(nt)] if rec_in is None else [rec_in[t % 8] for t in range(nt)]
                    if op in (fo.OP_BLOB_PROOF, fo.OP_VERIFY_BLOB):
                        cm = [k.blob_to_kzg_commitment(b) for b in blobs]
                        aux = [cm[t % 32] for t in range(nt)]
                    if op == fo.OP_VERIFY_BLOB:
                        pr = [k.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
                        aux = [cm[t % 32] + pr[t % 32] for t in range(nt)]
                    fo.run(k, mod.HIP_SO, op, ins, seconds=0.15, aux=aux, aux_n=aux_n)   # warm-up: arenas, batch buffers
                    before = fo.coalesce_stats(k, idx)
                    st, rets, _ = fo.run(k, mod.HIP_SO, op, ins, seconds=args.seconds, aux=aux, aux_n=aux_n)
                    after = fo.coalesce_stats(k, idx)
                    row = {"op": op_name, "wide": args.wide, "coalesce_active": act, "threads": nt,
                           "calls_per_s": round(st["calls_per_s"], 1), "mean_call_ms": round(st["mean_call_ms"], 3),
                           "p50_call_ms": round(st["p50_call_ms"], 3), "p99_call_ms": round(st["p99_call_ms"], 3),
                           "p999_call_ms": round(st["p999_call_ms"], 3),
                           "worst_call_ms": round(st["worst_call_ms"], 3), "not_ok": st["not_ok"]}
                    if after:
                        d = {n: after[n] - before[n] for n in ("calls", "solo", "batches", "batched", "run_us", "retried")}
                        row["retried"] = d["retried"]
                        row["mean_launch_ms"] = round(d["run_us"] / d["batches"] / 1e3, 3) if d["batches"] else None
                        row["launches"] = d["solo"] + d["batches"]
                        row["mean_batch"] = round(d["batched"] / d["batches"], 1) if d["batches"] else None
                        row["largest_batch"] = after["largest"]
                    print(json.dumps(row), flush=True)
        finally:
            k.close()


if __name__ == "__main__":
    main()
