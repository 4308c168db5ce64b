#!/usr/bin/env python3
"""One secondary row of the path, a few calls only, for rocprofv3 passes (tools/pmc_rows.sh):
  cells_wide    compute_cells_and_kzg_proofs: 1 blob x 5 (low-latency path: k_msm_accumulate over the 16-bit monomial
                table) and a 2048-blob batch x 2 (FK20: k_msm_small<8>, the G1-FFT ladders, k_ntt_tile), 16/16/13-bit tables
  cells_default the same on the library's default tables
  verify_wide   ckzg_hip_verify_blob_kzg_proof_batch_device, 4096 blobs x 3 (k_sha256_challenges, k_eval_tree,
                validation, call-time table) + recover_cells_and_kzg_proofs batch of 256 rows x 2, 16/16/13-bit tables
  verify_default the same on the library's default tables
  cells_small_default / cells_small_wide   compute_cells_and_kzg_proofs batches of 2, 8, 16, 32, 64 and 128 blobs x 6 each
                (resident): every form the two G1 transforms of FK20 take for small batches (fk20.hip: three-wave /
                two-wave pipelines, one-wave radix-8, radix-4 pairs) and k_msm_small's one-wave-per-vector form
Prints one JSON line with the wall-clock of what it ran."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (also sets GPU_MAX_HW_QUEUES before HIP starts)


def main():
    row = sys.argv[1]
    import torch
    import __graft_entry__ as ge
    mod = ge.load_package()
    opts = dict(bench.WIDE) if row.endswith("_wide") else {}
    if row.startswith("cells_small"):
        opts["direct_max"] = 0   # every size through FK20 (the low-latency path would take 1-2 blobs)
    hip = mod.Kzg(mod.HIP_SO, options=opts)
    L = bench.Lib(hip.lib)
    sp = C.addressof(hip.s)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0xC4B64844)
    out = {"row": row, "tables": bench.tables_of(L, hip)}
    if row.startswith("cells_small"):
        nb = 128
        blobs = torch.randint(0, 256, (nb, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
        blobs[:, :, 0] = 0
        status = torch.empty((nb,), dtype=torch.uint8, device=dev)
        cells = torch.empty((nb, 128, 2048), dtype=torch.uint8, device=dev)
        proofs = torch.empty((nb, 128, 48), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for n in (2, 8, 16, 32, 64, 128):
            ts = []
            for _ in range(7):
                t = time.perf_counter()
                rc = L.cells_dev(cells.data_ptr(), proofs.data_ptr(), status.data_ptr(), blobs.data_ptr(), n, sp)
                ts.append(time.perf_counter() - t)
                assert rc == 0
            out["cells_and_proofs_n%d_ms" % n] = round(min(ts[1:]) * 1e3, 3)
    elif row.startswith("cells"):
        nb = 2048
        blobs = torch.randint(0, 256, (nb, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
        blobs[:, :, 0] = 0
        status = torch.empty((nb,), dtype=torch.uint8, device=dev)
        cells = torch.empty((nb, 128, 2048), dtype=torch.uint8, device=dev)
        proofs = torch.empty((nb, 128, 48), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for n, reps in ((1, 5), (nb, 2)):
            ts = []
            for _ in range(reps + 1):
                t = time.perf_counter()
                rc = L.cells_dev(cells.data_ptr(), proofs.data_ptr(), status.data_ptr(), blobs.data_ptr(), n, sp)
                ts.append(time.perf_counter() - t)
                assert rc == 0
            out["cells_and_proofs_n%d_ms" % n] = round(min(ts[1:]) * 1e3, 3)
    else:
        ub = [bench_blob(i) for i in range(8)]
        cm = [hip.blob_to_kzg_commitment(b) for b in ub]
        pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(ub, cm)]
        n = 4096
        hb = bench.HipBuffers(torch, dev)
        d = [hb.device(b"".join(x[i % 8] for i in range(n))) for x in (ub, cm, pr)]
        ok = C.c_bool(False)
        ts = []
        for _ in range(4):
            t = time.perf_counter()
            rc = L.verify_blobs_dev(C.byref(ok), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, sp)
            ts.append(time.perf_counter() - t)
            assert rc == 0 and ok.value
        out["verify_resident_n4096_ms"] = round(min(ts[1:]) * 1e3, 3)
        cp = [hip.compute_cells_and_kzg_proofs(b) for b in ub]
        nb = 256
        keep = list(range(0, 128, 2))
        data = b"".join(b"".join(cp[b % 8][0][i] for i in keep) for b in range(nb))
        kidx = (C.c_uint64 * len(keep))(*keep)
        rc_buf = C.create_string_buffer(nb * 128 * 2048)
        rp_buf = C.create_string_buffer(nb * 128 * 48)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            rc = L.recover(rc_buf, rp_buf, None, kidx, data, C.c_uint64(len(keep)), C.c_uint64(nb), C.c_void_p(sp))
            ts.append(time.perf_counter() - t)
            assert rc == 0
        out["recover_batch256_ms"] = round(min(ts[1:]) * 1e3, 3)
    print(json.dumps(out))
    hip.close()


def bench_blob(i):
    import hashlib
    return b"".join(b"\x00" + hashlib.sha256(b"pmc%d|%d" % (i, j)).digest()[:31] for j in range(4096))


if __name__ == "__main__":
    main()
