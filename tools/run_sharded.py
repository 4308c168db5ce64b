#!/usr/bin/env python3
"""BASELINE configs[3] and configs[4] in their multi-GPU form: one process per GPU (torchrun), blobs /
rows sharded contiguously across ranks, no collective on the data path; the verdict of the verify batch
is an AND over shards and the recovered rows are all-gathered (c-kzg-4844_amd/multi_gpu.py).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_sharded.py

On a one-GPU box the control flow can be exercised with CKZG_ONE_GPU=1 CKZG_BACKEND=gloo and 2 ranks."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

import __graft_entry__ as ge
from test_gpu_commitment import rand_blob


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("CKZG_ONE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    backend = os.environ.get("CKZG_BACKEND", "nccl")
    dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    red_dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    mod = ge.load_package()
    hip = mod.Kzg(mod.HIP_SO, options={"device": local})
    import importlib
    mg = importlib.import_module("ckzg_4844_amd.multi_gpu")
    n_verify = int(os.environ.get("N_VERIFY", "4096"))
    n_rows = int(os.environ.get("N_ROWS", "256"))
    base = [rand_blob(99, i) for i in range(8)]
    cm = [hip.blob_to_kzg_commitment(b) for b in base]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(base, cm)]
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in base]
    fv = hip.lib.verify_blob_kzg_proof_batch
    fv.restype = C.c_int

    def verify(lo, hi):
        n = hi - lo
        bb = b"".join(base[i % 8] for i in range(lo, hi))
        cc = b"".join(cm[i % 8] for i in range(lo, hi))
        pp = b"".join(pr[i % 8] for i in range(lo, hi))
        ok = C.c_bool(False)
        rc = fv(C.byref(ok), bb, cc, pp, C.c_uint64(n), hip.sp)
        return rc, ok.value

    dist.barrier()
    t = time.perf_counter()
    rc, ok = mg.sharded_verify(verify, n_verify, red_dev)
    dt = time.perf_counter() - t
    keep = list(range(0, 128, 2))

    def recover(lo, hi):
        rows = [[cp[b % 8][0][i] for i in keep] for b in range(lo, hi)]
        _, rp = hip.recover_cells_and_kzg_proofs_batch(keep, rows, want_cells=False)
        flat = b"".join(b"".join(r) for r in rp)
        return torch.frombuffer(bytearray(flat), dtype=torch.uint8).reshape(hi - lo, 128 * 48)

    dist.barrier()
    t = time.perf_counter()
    proofs = mg.sharded_map(recover, n_rows, 128 * 48, red_dev)
    dr = time.perf_counter() - t
    good = all(bytes(proofs[b].cpu().numpy().tobytes()) == b"".join(cp[b % 8][1]) for b in (0, n_rows // 2, n_rows - 1))
    if rank == 0:
        print("verify_blob_kzg_proof_batch: %d blobs over %d ranks: rc=%d ok=%s, %.1f ms (includes building the host buffers)"
              % (n_verify, world, rc, ok, dt * 1e3))
        print("recover_cells_and_kzg_proofs: %d rows over %d ranks: gathered proofs correct=%s, %.1f ms" % (n_rows, world, good, dr * 1e3))
    hip.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
