#!/usr/bin/env python3
"""bench.py -- headline measurement: blobs/s of blob_to_kzg_commitment on a 1024-blob batch per GPU
(BASELINE.json configs[1]); weak scaling over N GPUs = N independent shards, no data-path collective
(SURVEY.md section 8e).  One JSON line on rank 0.

A step = one call of ckzg_hip_blob_to_kzg_commitment_batch_device over 1024 synthetic blobs that
are already resident in HBM (device pointers; the PCIe-inclusive host-pointer rate is reported
separately as pcie_inclusive_blobs_per_s and is never `value`).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOBS_PER_STEP = 1024
ALGO_BYTES_PER_BLOB = 131072 + 48  # scalars in + commitment out
HBM_PEAK_GBS = 8000.0
# HBM bytes per k_msm_accumulate launch (1024 blobs) from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE,
# KB units), see profiles/README.md; keyed by table window width.  The traffic is the table gathers
# themselves (nwin*4096 x 96 B per blob), not re-reads of the algorithmic bytes.
PMC_TRAFFIC_BYTES = {13: (7658816 + 960) * 1024, 15: (6796321 + 960) * 1024, 16: (6316804 + 192) * 1024}
# v_mad_u64_u32 per mixed addition (g1_28.hpp: xyzz28_madd_alt): 6 products x 392, 2 squares x 301,
# one fused two-product reduction x 588
MADS_PER_ADDITION = 6 * 392 + 2 * 301 + 588
# SQ_INSTS_VALU per 1024-blob launch (profiles/r01_c16_pmc_sq_k_msm_accumulate.json)
PMC_VALU_INSTS = {16: 4.98e9}


def cpu_baseline(seconds_budget=12.0):
    """Oracle ('port' of the reference algorithm, oracle/okzg.c) timed on this host's cores."""
    import hashlib
    import __graft_entry__ as ge
    mod = ge.load_package()
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        return None
    orc = mod.Kzg(so, "okzg_")
    blob = b"".join(b"\x00" + hashlib.sha256(b"cpu%d" % j).digest()[:31] for j in range(4096))
    orc.blob_to_kzg_commitment(blob)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds_budget and n < 512:
        orc.blob_to_kzg_commitment(blob)
        n += 1
    dt = time.perf_counter() - t0
    # the same port on many cores at once, one blob per thread (the shape of the reference's
    # ComputeCellsAndKZGProofsParallel benchmark, bindings/go/main_test.go:953-971); ctypes
    # releases the GIL during the C call
    import threading
    nthreads = min(64, os.cpu_count() or 1)
    per_thread = 6
    counts = [0] * nthreads

    def work(i):
        for _ in range(per_thread):
            orc.blob_to_kzg_commitment(blob)
            counts[i] += 1

    ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t1 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt_mt = time.perf_counter() - t1
    orc.close()
    return {"value": round(n / dt, 3), "unit": "blobs/s", "cores": 1, "kind": "port",
            "sample": "%d x blob_to_kzg_commitment on one 4096-element blob, oracle/liboracle.so "
                      "(portable C, Pippenger), single thread; host has %d logical CPUs"
                      % (n, os.cpu_count() or 0),
            "all_cores": {"value": round(sum(counts) / dt_mt, 2), "unit": "blobs/s", "cores": nthreads,
                          "sample": "%d threads x %d commitments" % (nthreads, per_thread)}}


def verify_and_recover_rows(hip, lib, base):
    """verify_blob_kzg_proof_batch over 4096 blobs (this GPU's view of configs[3]: the whole batch on one
    GPU), verify_cell_kzg_proof_batch over 8192 cells, and a 256-row recover batch (configs[4]),
    timed at the C-ABI with host buffers; inputs are 8 distinct valid blobs repeated."""
    ub = [base[i].tobytes() for i in range(8)]
    cm = [hip.blob_to_kzg_commitment(b) for b in ub]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(ub, cm)]
    out = {}
    sp = C.addressof(hip.s)
    n = 4096
    bb = b"".join(ub[i % 8] for i in range(n))
    cc = b"".join(cm[i % 8] for i in range(n))
    pp = b"".join(pr[i % 8] for i in range(n))
    fv = lib.verify_blob_kzg_proof_batch
    fv.restype = C.c_int
    fv.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
    ok = C.c_bool(False)
    for k in (512, n):
        fv(C.byref(ok), bb, cc, pp, k, sp)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            rc = fv(C.byref(ok), bb, cc, pp, k, sp)
            best = min(best, time.perf_counter() - t)
        if rc != 0 or not ok.value:
            raise RuntimeError("verify_blob_kzg_proof_batch rc=%d ok=%s" % (rc, ok.value))
        out["verify_blob_kzg_proof_batch_n%d_blobs_per_s" % k] = round(k / best, 1)
    del bb
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in ub]
    n = 8192
    rows = [(i // 128) % 8 for i in range(n)]
    cols = [i % 128 for i in range(n)]
    ccm = b"".join(cm[r] for r in rows)
    idx = (C.c_uint64 * n)(*cols)
    cells = b"".join(cp[r][0][c] for r, c in zip(rows, cols))
    cprf = b"".join(cp[r][1][c] for r, c in zip(rows, cols))
    fc = lib.verify_cell_kzg_proof_batch
    fc.restype = C.c_int
    fc.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
    for k in (128, n):
        fc(C.byref(ok), ccm, idx, cells, cprf, k, sp)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            rc = fc(C.byref(ok), ccm, idx, cells, cprf, k, sp)
            best = min(best, time.perf_counter() - t)
        if rc != 0 or not ok.value:
            raise RuntimeError("verify_cell_kzg_proof_batch rc=%d ok=%s" % (rc, ok.value))
        out["verify_cell_kzg_proof_batch_n%d_ms" % k] = round(best * 1e3, 3)
    nb = 256
    keep = list(range(0, 128, 2))
    fr = lib.ckzg_hip_recover_cells_and_kzg_proofs_batch
    fr.restype = C.c_int
    data = b"".join(b"".join(cp[b % 8][0][i] for i in keep) for b in range(nb))
    kidx = (C.c_uint64 * len(keep))(*keep)
    rc_buf = C.create_string_buffer(nb * 128 * 2048)
    rp_buf = C.create_string_buffer(nb * 128 * 48)
    args = (rc_buf, rp_buf, None, kidx, data, C.c_uint64(len(keep)), C.c_uint64(nb), C.c_void_p(sp))
    fr(*args)
    t = time.perf_counter()
    rc = fr(*args)
    dt = time.perf_counter() - t
    if rc != 0 or rp_buf.raw[:128 * 48] != b"".join(cp[0][1]):
        raise RuntimeError("recover batch rc=%d or wrong proofs" % rc)
    out["recover_cells_and_kzg_proofs_batch256_rows_per_s"] = round(nb / dt, 1)
    out["recover_note"] = "64 of 128 cells per row, same columns in every row; cells and proofs out"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--wbits", type=int, default=int(os.environ.get("CKZG_BENCH_WBITS", "16")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the untimed host-pointer (PCIe-inclusive) leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  CKZG_BENCH_BACKEND=gloo + CKZG_BENCH_ONE_GPU=1 exist only to exercise
        # this file's multi-rank control flow on a one-GPU box (all ranks share device 0).
        dist.init_process_group(os.environ.get("CKZG_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if os.environ.get("CKZG_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")

    import __graft_entry__ as ge
    mod = ge.load_package()
    # headline leg: the widest commitment table that fits (16-bit windows = 206 GB of the 288 GB);
    # the cell-proof tables stay at their small defaults here and are widened for the secondary leg
    hip = None
    for w in range(args.wbits, 9, -1):
        try:
            hip = mod.Kzg(mod.HIP_SO, options={"device": local_rank, "commit_wbits": w, "proof_wbits": 8,
                                               "fk20_wbits": 8})
            break
        except Exception as e:  # the library already narrows to the free HBM; this is the belt to its braces
            sys.stderr.write("bench: load with commit_wbits=%d failed (%s), trying %d\n" % (w, e, w - 1))
    if hip is None:
        raise SystemExit("bench: load_trusted_setup failed for every table width")
    lib = hip.lib
    lib.ckzg_hip_table_wbits.restype = C.c_int
    lib.ckzg_hip_table_wbits.argtypes = [C.c_void_p, C.c_int]
    wbits = int(lib.ckzg_hip_table_wbits(C.addressof(hip.s), 0))  # what was actually built
    fn = lib.ckzg_hip_blob_to_kzg_commitment_batch_device
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    kms = lib.ckzg_hip_last_kernel_ms
    kms.restype = C.c_double
    kms.argtypes = [C.c_void_p, C.c_int]
    lib.ckzg_hip_table_bytes.restype = C.c_uint64
    lib.ckzg_hip_table_bytes.argtypes = [C.c_void_p]

    # synthetic blobs: 31 random bytes per field element, top byte 0 => canonical
    # (same distribution as bindings/go/main_test.go:31-51), fixed seed per rank
    g = torch.Generator(device=dev)
    g.manual_seed(0xC4B64844 + rank)
    blobs = torch.randint(0, 256, (BLOBS_PER_STEP, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
    blobs[:, :, 0] = 0
    out = torch.empty((BLOBS_PER_STEP, 48), dtype=torch.uint8, device=dev)
    status = torch.empty((BLOBS_PER_STEP,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        rc = fn(out.data_ptr(), status.data_ptr(), blobs.data_ptr(), BLOBS_PER_STEP, C.addressof(hip.s))
        if rc != 0:
            raise RuntimeError("commit batch failed rc=%d" % rc)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        step()
        kern_ms.append(kms(C.addressof(hip.s), 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # PCIe-inclusive rate (host pointers), one untimed-for-`value` call on rank 0
    pcie_rate = None
    if rank == 0 and not args.no_pcie:
        try:
            hb = blobs.cpu().numpy().tobytes()
            ho = C.create_string_buffer(48 * BLOBS_PER_STEP)
            hs = C.create_string_buffer(BLOBS_PER_STEP)
            f2 = lib.ckzg_hip_blob_to_kzg_commitment_batch
            f2.restype = C.c_int
            f2.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
            f2(ho, hs, hb, C.c_uint64(BLOBS_PER_STEP), C.addressof(hip.s))  # warm-up: pinned staging, buffers
            t1 = time.perf_counter()
            rc = f2(ho, hs, hb, C.c_uint64(BLOBS_PER_STEP), C.addressof(hip.s))
            t2 = time.perf_counter()
            if rc == 0:
                pcie_rate = BLOBS_PER_STEP / (t2 - t1)
                assert ho.raw == out.cpu().numpy().tobytes(), "host-pointer and device-pointer paths disagree"
        except AssertionError:
            raise
        except Exception as e:  # reported as null, never fatal for the headline
            sys.stderr.write("bench: PCIe-inclusive leg failed: %s\n" % e)

    # spot-check the timed kernel's output against the CPU oracle (checker only, untimed)
    parity = None
    if rank == 0:
        try:
            orc = mod.Kzg(os.path.join(ROOT, "oracle", "liboracle.so"), "okzg_")
            hb_all = blobs.cpu().numpy()
            ho_all = out.cpu().numpy()
            parity = True
            for i in (0, 1, 511, BLOBS_PER_STEP - 1):
                parity &= orc.blob_to_kzg_commitment(hb_all[i].tobytes()) == ho_all[i].tobytes()
            orc.close()
        except Exception as e:
            parity = "oracle unavailable: %s" % e
        if parity is False:
            raise SystemExit("bench: GPU commitments differ from the oracle -- number would be invalid")

    table_bytes = int(lib.ckzg_hip_table_bytes(C.addressof(hip.s)))
    # secondary metric of BASELINE.json: compute_cells_and_kzg_proofs (configs[2]), rank 0 only, on two
    # further loads: a latency configuration (16-bit table over the monomial points for the
    # low-latency proof path) and a throughput configuration (15-bit FK20 table)
    # secondary rows: never allowed to take the headline line down with them
    secondary = None

    def secondary_rows():
        nonlocal hip
        secondary = None
        fc = lib.ckzg_hip_compute_cells_and_kzg_proofs_batch_device
        fc.restype = C.c_int
        fc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        nb = 2048
        blobs2 = blobs.repeat(2, 1, 1)
        status2 = torch.empty((nb,), dtype=torch.uint8, device=dev)
        cells = torch.empty((nb, 128, 2048), dtype=torch.uint8, device=dev)
        proofs = torch.empty((nb, 128, 48), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def run(n):
            rc = fc(cells.data_ptr(), proofs.data_ptr(), status2.data_ptr(), blobs2.data_ptr(), n, C.addressof(hip.s))
            if rc != 0:
                raise RuntimeError("cells+proofs failed rc=%d" % rc)

        def tables():
            return {"fk20_wbits": int(lib.ckzg_hip_table_wbits(C.addressof(hip.s), 1)),
                    "proof_wbits": int(lib.ckzg_hip_table_wbits(C.addressof(hip.s), 2)),
                    "bytes": int(lib.ckzg_hip_table_bytes(C.addressof(hip.s)))}

        hip.close()
        hip = mod.Kzg(mod.HIP_SO, options={"device": local_rank, "commit_wbits": 10, "proof_wbits": 16,
                                           "fk20_wbits": 8})
        run(1)
        ts = []
        for _ in range(20):
            t1 = time.perf_counter()
            run(1)
            ts.append(time.perf_counter() - t1)
        ts.sort()
        secondary = {"compute_cells_and_kzg_proofs_ms_per_call_1blob": round(ts[len(ts) // 2] * 1e3, 3),
                     "tables_1blob": tables()}
        proofs_1 = proofs[0].clone()
        hip.close()
        hip = mod.Kzg(mod.HIP_SO, options={"device": local_rank, "commit_wbits": 10, "proof_wbits": 8,
                                           "fk20_wbits": 15})
        run(nb)
        t1 = time.perf_counter()
        run(nb)
        tb = time.perf_counter() - t1
        if not torch.equal(proofs[0], proofs_1):
            raise SystemExit("bench: low-latency and FK20 proof paths disagree")
        secondary.update({"compute_cells_and_kzg_proofs_batch2048_blobs_per_s": round(nb / tb, 1),
                          "tables_batch": tables(),
                          "note": "1 blob: low-latency path (128 fixed-base MSMs, no G1 FFT); batch: FK20 path; "
                                  "inputs/outputs resident in HBM; the two paths' proofs are compared"})
        # the other two rows of the path (BASELINE configs 4 and 5), host pointers at the C-ABI
        try:
            secondary.update(verify_and_recover_rows(hip, lib, blobs[:8].cpu().numpy()))
        except Exception as e:  # reported, never fatal for the headline
            secondary["verify_recover_error"] = str(e)
        return secondary

    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            secondary = secondary_rows()
        except BaseException as e:  # noqa: BLE001 -- report, keep the headline
            if isinstance(e, SystemExit) and 'disagree' in str(e):
                raise
            secondary = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        total_blobs = BLOBS_PER_STEP * args.steps * world
        value = total_blobs / dt
        avg_k = sum(kern_ms) / len(kern_ms) * 1e-3
        achieved = ALGO_BYTES_PER_BLOB * BLOBS_PER_STEP / avg_k / 1e9
        line = {
            "metric": "blob_to_kzg_commitment throughput",
            "value": round(value, 2), "unit": "blobs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "blob_to_kzg_commitment batch of 1024 blobs per GPU (4096-point G1 MSM per blob), "
                                   "inputs resident in HBM", "blobs_per_step_per_gpu": BLOBS_PER_STEP,
                       "table_wbits": wbits, "table_bytes": table_bytes,
                       "parallelism": "independent blob shards per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": PMC_TRAFFIC_BYTES.get(wbits),
                         "kernel": "k_msm_accumulate", "kernel_ms": round(avg_k * 1e3, 3),
                         "note": "integer-VALU-bound kernel (v_mad_u64_u32 chains); HBM fraction is small by nature"},
            # the physical bound of this kernel: integer multiply-add issue rate.  peak = measured
            # v_mad_u64_u32 rate of the chip (tools/ubench/instr_rates.hip: 32.9e12 lane-ops/s);
            # achieved counts only the multiply-adds of the field products of each table addition
            # (MADS_PER_ADDITION; nwin*4096 additions per blob), not the ~25 % of other instructions.
            "roofline_valu": {"bound": "v_mad_u64_u32 issue", "unit": "T lane-mad/s", "peak": 32.9,
                              "achieved": round(BLOBS_PER_STEP * (255 // wbits + 1) * 4096 * MADS_PER_ADDITION / avg_k / 1e12, 3),
                              "frac": round(BLOBS_PER_STEP * (255 // wbits + 1) * 4096 * MADS_PER_ADDITION / avg_k / 32.9e12, 4),
                              "valu_wave_insts_per_launch": PMC_VALU_INSTS.get(wbits),
                              "pmc": "profiles/r01_c16_pmc_sq_k_msm_accumulate.json (SQ_INSTS_VALU, GRBM_GUI_ACTIVE): "
                                     "~95 % of the VALU issue slots at the sustained ~2.1 GHz clock"},
            "pcie_inclusive_blobs_per_s": None if pcie_rate is None else round(pcie_rate, 2),
            "parity_spot_check_vs_oracle": parity,
            "secondary": secondary,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the oracle is only a reported baseline
                line["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(line))
    hip.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
