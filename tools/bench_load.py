#!/usr/bin/env python3
"""load_trusted_setup wall clock and its phases (ckzg_hip_load_times) for a given set of table widths.
usage: python tools/bench_load.py [commit_wbits proof_wbits fk20_wbits] [async]     (default 16 16 13, synchronous)
(The round-1/2 table builder this was A/B'd against is gone; profiles/r03_load_ab.jsonl keeps the comparison.)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def main():
    w = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [16, 16, 13]
    use_async = "async" in sys.argv
    mod = ge.load_package()
    t0 = time.perf_counter()
    hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": w[0], "proof_wbits": w[1], "fk20_wbits": w[2], "async_tables": int(use_async)})
    t_ret = time.perf_counter() - t0
    L = bench.Lib(hip.lib)
    sp = C.addressof(hip.s)
    c = hip.blob_to_kzg_commitment(bytes(131072))
    t_first = time.perf_counter() - t0
    L.wait_tables(sp)
    t_all = time.perf_counter() - t0
    assert c == b"\xc0" + bytes(47)
    print(json.dumps({"builder": "affine-chains", "async": use_async, "tables": bench.tables_of(L, hip),
                      "load_call_returned_after_s": round(t_ret, 3), "time_to_first_commitment_s": round(t_first, 3),
                      "all_tables_ready_s": round(t_all, 3), "phases_s": bench.load_phases(L, hip)}))
    hip.close()


if __name__ == "__main__":
    main()
