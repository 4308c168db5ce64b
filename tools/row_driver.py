#!/usr/bin/env python3
"""Run ONE secondary row of bench.py, for kernel-level profiling of that row alone:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tools/row_driver.py <row> [wide|default]
rows: cells (1-blob latency + 2048-blob batch of compute_cells_and_kzg_proofs), verify (blob batches of 512 /
4096, cell batches of 128 / 8192, 256-row recover), concurrent (1 vs 8 threads of single-blob commitments)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (also sets GPU_MAX_HW_QUEUES before HIP starts)


def main():
    row = sys.argv[1]
    wide = len(sys.argv) > 2 and sys.argv[2] == "wide"
    import torch
    import __graft_entry__ as ge
    mod = ge.load_package()
    opts = dict(bench.WIDE) if wide else {"commit_wbits": 10, "proof_wbits": 8, "fk20_wbits": 0}
    hip = mod.Kzg(mod.HIP_SO, options=opts)
    L = bench.Lib(hip.lib)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0xC4B64844)
    blobs = torch.randint(0, 256, (1024, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
    blobs[:, :, 0] = 0
    torch.cuda.synchronize()
    if row == "cells":
        out = bench.cells_rows(L, hip, torch, dev, blobs, "wide" if wide else "default")
    elif row == "verify":
        out = bench.verify_and_recover_rows(L, hip, blobs[:8].cpu().numpy())
    elif row == "concurrent":
        out = bench.concurrency_row(hip, blobs[0].cpu().numpy().tobytes())
    else:
        raise SystemExit("unknown row " + row)
    out["tables"] = bench.tables_of(L, hip)
    print(json.dumps(out))
    hip.close()


if __name__ == "__main__":
    main()
