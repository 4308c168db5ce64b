#!/usr/bin/env python3
"""bench.py -- headline measurement: blobs/s of blob_to_kzg_commitment on a 1024-blob batch per GPU
(BASELINE.json configs[1]); weak scaling over N GPUs = N independent shards, no data-path collective
(SURVEY.md section 8e).  One JSON line on rank 0.

A step = one call of ckzg_hip_blob_to_kzg_commitment_batch_device over 1024 synthetic blobs that are
already resident in HBM (the task's measurement contract: `value` is the HBM-resident rate).  The same
1024 blobs through the reference-shaped host-pointer call (H2D + D2H inside the timed region, pageable
host memory) are timed over the same number of steps and reported next to it as `host_pointer`.

ONE KZGSettings (commit 16-bit + proof 16-bit + FK20 13-bit GLV tables, ~238 GB) serves every wide-table
row of the line; a second, co-resident KZGSettings with the library's default tables (~7 GB) gives the
default-footprint figures.  Nothing is reloaded between rows.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself through
torch.distributed.run with N ranks (one per GPU); under the driver's own torchrun launch it reads
RANK / LOCAL_RANK / WORLD_SIZE as usual.
"""
import argparse
import ctypes as C
import json
import os
import socket
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before HIP initialises: concurrent callers need the queues

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOBS_PER_STEP = 1024
ALGO_BYTES_PER_BLOB = 131072 + 48          # scalars in + commitment out
ALGO_BYTES_CELLS_PROOFS = 131072 + 262144 + 6144   # blob in, cells + proofs out
HBM_PEAK_GBS = 8000.0
# v_mad_u64_u32 per mixed addition (g1_28.hpp: xyzz28_madd_alt): 6 products x 392, 2 squares x 301,
# one fused two-product reduction x 588
MADS_PER_ADDITION = 6 * 392 + 2 * 301 + 588
ALGO_BYTES_VERIFY_BLOB = 131072 + 48 + 48          # blob + commitment + proof per blob
ALGO_BYTES_RECOVER_ROW = 131072 + 262144 + 6144    # 64 cells in, 128 cells + 128 proofs out
PCIE_PEAK_GBS = 63.0                               # PCIe Gen5 x16, one direction (spec; the measured H2D rate is printed next to it)
WIDE = {"commit_wbits": 16, "proof_wbits": 16, "fk20_wbits": 13}


def respawn(n):
    """`python bench.py --gpus N` without a launcher: become N ranks through torch.distributed.run."""
    import torch
    have = torch.cuda.device_count()
    if have < n and not (os.environ.get("CKZG_BENCH_ONE_GPU") and have >= 1):  # (control-flow test: ranks share device 0)
        raise SystemExit("bench: --gpus %d but only %d HIP device(s) visible" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def analytic_traffic(wbits, blobs):
    """HBM bytes one k_msm_accumulate launch has to move, from the table geometry alone: every one of the
    nwin * 4096 (window, point) pairs of a blob gathers one 96-byte table entry and reads one int16 digit; one
    192-byte partial sum is written per workgroup.  (Zero digits skip their gather: 2^-wbits of the pairs.)"""
    nwin = 2 * (127 // wbits + 1)
    pairs = nwin * 4096
    return int(blobs * (pairs * (96 * (1.0 - 2.0 ** -wbits) + 2) + 192))


def pmc_cross_check():
    """The newest committed PMC summaries of the headline kernel (separate rocprofv3 --pmc passes, tools/profile_bench.sh):
    a cross-check of the analytic traffic and the instruction count, named with its file, never a constant of this script."""
    import glob
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_pmc_k_msm_accumulate.json")))[-1:]:
        try:
            d = json.load(open(f))
            kb = d["pmc_FETCH_SIZE"]["counter_mean"]["FETCH_SIZE"] + d["pmc_WRITE_SIZE"]["counter_mean"]["WRITE_SIZE"]
            out["traffic_bytes_per_launch"] = int(kb * 1024)
            out["traffic_file"] = os.path.relpath(f, ROOT)
        except Exception as e:  # noqa: BLE001
            out["traffic_error"] = str(e)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_pmc_sq_k_msm_accumulate.json")))[-1:]:
        try:
            d = json.load(open(f))["pmc_SQ"]["counter_mean"]
            out["valu_wave_insts_per_launch"] = d["SQ_INSTS_VALU"]
            out["sq_file"] = os.path.relpath(f, ROOT)
        except Exception as e:  # noqa: BLE001
            out["sq_error"] = str(e)
    return out or None


VALU_PEAK_TMADS = 32.9   # measured v_mad_u64_u32 rate of the chip at 2.4 GHz, T lane-mad/s (tools/ubench/instr_rates.hip)
LADDER_PRODUCTS = 1510   # field products of one twiddle multiplication of the G1 FFT, one-lane co-Z form
MADS_PER_PRODUCT = 392


def valu_roofline(lane_mads, kernel_ms, what):
    """Achieved v_mad_u64_u32 rate of a kernel against the chip's measured rate: the honest bound of the curve work."""
    if not kernel_ms or kernel_ms <= 0:
        return None
    ach = lane_mads / (kernel_ms * 1e-3) / 1e12
    return {"bound": "v_mad_u64_u32 issue", "unit": "T lane-mad/s", "peak": VALU_PEAK_TMADS, "achieved": round(ach, 3),
            "frac": round(ach / VALU_PEAK_TMADS, 4), "counted": what}


_PMC_CACHE = {}


def pmc_rows(row):
    """Newest committed per-kernel counter summary of a secondary row (tools/pmc_rows.sh -> profiles/rNN_pmc_<row>.json):
    separate rocprofv3 --pmc passes, per launch."""
    import glob
    if row not in _PMC_CACHE:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % row)))
        d = None
        if files:
            try:
                d = json.load(open(files[-1]))
                d["_file"] = os.path.relpath(files[-1], ROOT)
            except Exception:  # noqa: BLE001
                d = None
        _PMC_CACHE[row] = d
    return _PMC_CACHE[row]


def pmc_kernel(row, name, pick="longest", streaming=False):
    """Counters of kernel `name` in that summary (of its several launch shapes: the one with the longest mean duration,
    or the largest grid).  streaming: the kernel reads with wide coalesced loads, for which gfx950's FETCH_SIZE tallies
    64 B per 128-B request (MI355X_MICROARCH.md, HBM section) -- doubled here, and the entry says so."""
    d = pmc_rows(row)
    if not d:
        return None
    ks = [k for k in d["kernels"] if k["kernel"].startswith(name) and "counters_per_launch" in k]
    if not ks:
        return None
    k = max(ks, key=(lambda x: x["mean_ms"]) if pick == "longest" else (lambda x: int(x["grid"] or 0)))
    c = k["counters_per_launch"]
    fetch = c.get("FETCH_SIZE", 0.0) * 1024 * (2 if streaming else 1)
    out = {"kernel": k["kernel"], "grid": k["grid"], "launches_profiled": k["launches"], "mean_ms": k["mean_ms"],
           "fetch_bytes": int(fetch), "write_bytes": int(c.get("WRITE_SIZE", 0.0) * 1024),
           "traffic_bytes_per_launch": int(fetch + c.get("WRITE_SIZE", 0.0) * 1024),
           "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 64 B per 128-B streaming request)" if streaming else "none (96-byte gathers: matches the analytic count)",
           "valu_wave_insts": c.get("SQ_INSTS_VALU"), "waves": c.get("SQ_WAVES"), "wave_quad_cycles": c.get("SQ_WAVE_CYCLES"),
           "file": d["_file"], "tables_profiled": d.get("tables")}
    return out


def cpu_baseline(seconds_budget=12.0):
    """Oracle ('port' of the reference algorithm, oracle/okzg.c) timed on this host's cores."""
    import hashlib
    import threading
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
    # ComputeCellsAndKZGProofsParallel benchmark, bindings/go/main_test.go:953-971) -- NATIVE pthreads inside the
    # oracle library (oracle/obench.c), so the figure does not depend on Python threads or ctypes.
    # Every logical CPU the process may use is tried (N = nproc); containers often grant fewer cores than
    # they show, so the sample is repeated at smaller counts and the best rate is reported with its thread count
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    fn = orc.lib.okzg_bench_commit_threads
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    expect = orc.blob_to_kzg_commitment(blob)
    tried = {}
    for nthreads in sorted({usable, min(usable, 64), min(usable, 32), min(usable, 16)}, reverse=True):
        first = C.create_string_buffer(48)
        rate = fn(C.addressof(orc.s), blob, nthreads, 2 if nthreads > 64 else 4, first)
        if rate < 0 or first.raw != expect:
            raise RuntimeError("oracle pthread driver failed (%r) at %d threads" % (rate, nthreads))
        tried[nthreads] = round(rate, 2)
    best_threads = max(tried, key=lambda k: tried[k])
    # the other half of the metric: compute_cells_and_kzg_proofs, single thread: one warm-up, median of three
    orc.compute_cells_and_kzg_proofs(blob)
    cell_ts = []
    for _ in range(3):
        t2 = time.perf_counter()
        orc.compute_cells_and_kzg_proofs(blob)
        cell_ts.append(time.perf_counter() - t2)
    dt_cells = median(cell_ts)
    orc.close()
    return {"value": round(n / dt, 3), "unit": "blobs/s", "cores": 1, "kind": "port",
            "sample": "%d x blob_to_kzg_commitment on one 4096-element blob, oracle/liboracle.so "
                      "(portable C, Pippenger), single thread; host has %d logical CPUs"
                      % (n, os.cpu_count() or 0),
            "all_cores": {"value": tried[best_threads], "unit": "blobs/s", "cores": best_threads,
                          "logical_cpus": logical, "usable_cpus": usable,
                          "blobs_per_s_by_thread_count": {str(k): v for k, v in sorted(tried.items())},
                          "driver": "pthreads (oracle/obench.c)",
                          "sample": "one commitment per call, N native threads on their own blob copies, best of the thread counts tried"},
            "compute_cells_and_kzg_proofs_ms_per_call": round(dt_cells * 1e3, 1),
            "compute_cells_and_kzg_proofs_sample": "median of 3 calls after one warm-up, single thread"}


class Lib:
    """ctypes prototypes of the additive entry points the bench uses."""

    def __init__(self, lib):
        self.lib = lib
        p, u64 = C.c_void_p, C.c_uint64
        self.commit_dev = self._fn("ckzg_hip_blob_to_kzg_commitment_batch_device", [p, p, p, u64, p])
        self.commit_host = self._fn("ckzg_hip_blob_to_kzg_commitment_batch", [p, p, C.c_char_p, u64, p])
        self.cells_dev = self._fn("ckzg_hip_compute_cells_and_kzg_proofs_batch_device", [p, p, p, p, u64, p])
        self.cells_host = self._fn("ckzg_hip_compute_cells_and_kzg_proofs_batch", [p, p, p, C.c_char_p, u64, p])
        self.verify_blobs = self._fn("verify_blob_kzg_proof_batch", [p, C.c_char_p, C.c_char_p, C.c_char_p, u64, p])
        self.verify_cells = self._fn("verify_cell_kzg_proof_batch", [p, C.c_char_p, p, C.c_char_p, C.c_char_p, u64, p])
        self.recover = self._fn("ckzg_hip_recover_cells_and_kzg_proofs_batch", None)
        self.kms = self._fn("ckzg_hip_last_kernel_ms", [p, C.c_int], C.c_double)
        self.table_bytes = self._fn("ckzg_hip_table_bytes", [p], C.c_uint64)
        self.table_wbits = self._fn("ckzg_hip_table_wbits", [p, C.c_int])
        self.num_devices = self._fn("ckzg_hip_num_devices", [p])
        self.verify_blobs_dev = self._fn("ckzg_hip_verify_blob_kzg_proof_batch_device", [p, p, p, p, u64, p])
        self.load_times = self._fn("ckzg_hip_load_times", [p, C.POINTER(C.c_double), C.c_int])
        self.wait_tables = self._fn("ckzg_hip_wait_tables", [p])

    def _fn(self, name, argtypes, restype=C.c_int):
        f = getattr(self.lib, name)
        f.restype = restype
        if argtypes is not None:
            f.argtypes = argtypes
        return f


def tables_of(L, hip):
    sp = C.addressof(hip.s)
    return {"commit_wbits": int(L.table_wbits(sp, 0)), "fk20_wbits": int(L.table_wbits(sp, 1)),
            "proof_wbits": int(L.table_wbits(sp, 2)), "bytes": int(L.table_bytes(sp))}


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cells_rows(L, hip, torch, dev, blobs, label):
    """compute_cells_and_kzg_proofs on one loaded KZGSettings: 1-blob latency (BASELINE configs[2]) and the
    2048-blob batch, device pointers and host pointers, each with the dominant kernel's own time."""
    sp = C.addressof(hip.s)
    nb = 2048
    blobs2 = blobs.repeat(2, 1, 1)
    status = torch.empty((nb,), dtype=torch.uint8, device=dev)
    cells = torch.empty((nb, 128, 2048), dtype=torch.uint8, device=dev)
    proofs = torch.empty((nb, 128, 48), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def run(n):
        rc = L.cells_dev(cells.data_ptr(), proofs.data_ptr(), status.data_ptr(), blobs2.data_ptr(), n, sp)
        if rc != 0:
            raise RuntimeError("cells+proofs failed rc=%d" % rc)

    run(1)
    ts, ks = [], []
    for _ in range(20):
        t1 = time.perf_counter()
        run(1)
        ts.append(time.perf_counter() - t1)
        ks.append(L.kms(sp, 1))
    one_ms, one_k = median(ts) * 1e3, median(ks)
    proofs_1 = proofs[0].clone()
    cells_1 = cells[0].clone()
    # the reference-shaped one-blob call, host pointers in and out
    hb1 = blobs2[0].cpu().numpy().tobytes()
    hc1 = C.create_string_buffer(128 * 2048)
    hp1 = C.create_string_buffer(128 * 48)
    fn1 = L._fn("compute_cells_and_kzg_proofs", [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p])
    fn1(hc1, hp1, hb1, sp)
    th = []
    for _ in range(20):
        t1 = time.perf_counter()
        rc = fn1(hc1, hp1, hb1, sp)
        th.append(time.perf_counter() - t1)
    if rc != 0 or hp1.raw != proofs_1.cpu().numpy().tobytes() or hc1.raw != cells_1.cpu().numpy().tobytes():
        raise SystemExit("bench: host-pointer compute_cells_and_kzg_proofs differs from the device-pointer call")
    tb = tables_of(L, hip)
    pw, fw = tb["proof_wbits"], tb["fk20_wbits"]
    nwin_p = 2 * (127 // pw + 1) if pw else 0
    nwin_f = 2 * (127 // fw + 1)
    row = "cells_wide" if tb["commit_wbits"] >= 16 else "cells_default"
    same = lambda k: k if (k and k["tables_profiled"] and k["tables_profiled"].get("proof_wbits") == pw   # noqa: E731
                           and k["tables_profiled"].get("fk20_wbits") == fw) else None
    out = {"tables": tb,
           "one_blob": {"ms_per_call": round(median(th) * 1e3, 3), "calls_per_s": round(1.0 / median(th), 1),
                        "ms_per_call_device_pointers": round(one_ms, 3),
                        "path": "low-latency (128 fixed-base MSMs over the monomial table, no G1 FFT)",
                        "roofline": dict(roofline(ALGO_BYTES_CELLS_PROOFS, one_k, "k_msm_accumulate",
                                                  analytic_traffic(pw, 128) if pw else None),
                                         traffic_source="analytic: 128 x nwin x 4096 table gathers of 96 B + int16 digits (bench.py: analytic_traffic)",
                                         pmc_cross_check=same(pmc_kernel(row, "k_msm_accumulate<256>"))),
                        "roofline_valu": valu_roofline(128 * nwin_p * 4096 * MADS_PER_ADDITION, one_k,
                                                       "128 MSMs x %d windows x 4096 mixed additions x %d multiply-adds" % (nwin_p, MADS_PER_ADDITION))}}
    run(nb)
    tb, kb, fb = [], [], []
    for _ in range(3):
        t1 = time.perf_counter()
        run(nb)
        tb.append(time.perf_counter() - t1)
        kb.append(L.kms(sp, 1))
        fb.append(L.kms(sp, 4))
    if not torch.equal(proofs[0], proofs_1):
        raise SystemExit("bench: low-latency and FK20 proof paths disagree")
    t_b, k_b, f_b = median(tb), median(kb), median(fb)
    dom, dom_name = (k_b, "k_msm_small") if k_b >= f_b else (f_b, "k_g1_fft_twiddle+k_g1_fft_addsub (2 G1 FFTs)")
    small_traffic = int(nb * (128 * 64 * nwin_f * (96 * (1.0 - 2.0 ** -fw) + 2) + 128 * 192))
    fft_pmc = same(pmc_kernel(row, "k_g1_fft_twiddle", pick="grid"))
    mads_small = nb * 128 * 64 * nwin_f * MADS_PER_ADDITION
    mads_fft = nb * 642 * LADDER_PRODUCTS * MADS_PER_PRODUCT
    out["batch_2048"] = {"blobs_per_s": round(nb / t_b, 1), "ms_per_blob": round(t_b / nb * 1e3, 4),
                         "path": "FK20", "k_msm_small_ms": round(k_b, 3), "g1_fft_ms": round(f_b, 3),
                         "roofline": dict(roofline(ALGO_BYTES_CELLS_PROOFS * nb, dom, dom_name,
                                                   small_traffic if k_b >= f_b else (fft_pmc["traffic_bytes_per_launch"] * 12 if fft_pmc else None)),
                                          traffic_source="analytic: 128 x 64 x nwin gathers of 96 B + digits per blob (k_msm_small)" if k_b >= f_b
                                          else "PMC: 12 launches of k_g1_fft_twiddle per batch, FETCH_SIZE + WRITE_SIZE per launch",
                                          pmc_cross_check={"k_msm_small": same(pmc_kernel(row, "k_msm_small")), "k_g1_fft_twiddle": fft_pmc}),
                         "roofline_valu": {"k_msm_small": valu_roofline(mads_small, k_b, "128 x 64 x %d mixed additions x %d multiply-adds per blob" % (nwin_f, MADS_PER_ADDITION)),
                                           "g1_fft": valu_roofline(mads_fft, f_b, "642 twiddle ladders x %d products x %d multiply-adds per blob" % (LADDER_PRODUCTS, MADS_PER_PRODUCT)),
                                           "both_kernels": valu_roofline(mads_small + mads_fft, k_b + f_b, "k_msm_small + the two G1 FFTs: the curve work of compute_fk20_cell_proofs (src/eip7594/fk20.c:139-286)")}}
    # host pointers: pageable input, pageable outputs (268 KB per blob back over PCIe)
    hb = blobs2.cpu().numpy().tobytes()
    hc = C.create_string_buffer(nb * 128 * 2048)
    hp = C.create_string_buffer(nb * 128 * 48)
    hs = C.create_string_buffer(nb)
    L.cells_host(hc, hp, hs, hb, nb, sp)
    t1 = time.perf_counter()
    rc = L.cells_host(hc, hp, hs, hb, nb, sp)
    t_h = time.perf_counter() - t1
    if rc != 0 or hp.raw[:128 * 48] != proofs_1.cpu().numpy().tobytes():
        raise SystemExit("bench: host-pointer cells+proofs batch failed or disagrees")
    out["batch_2048"]["host_pointer_blobs_per_s"] = round(nb / t_h, 1)
    out["label"] = label
    return out


def roofline(algo_bytes, kernel_ms, kernel, traffic=None, bound="hbm", peak=HBM_PEAK_GBS):
    ach = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms and kernel_ms > 0 else None
    return {"bound": bound, "achieved": None if ach is None else round(ach, 3), "peak": peak, "unit": "GB/s",
            "frac": None if ach is None else round(ach / peak, 6), "traffic": traffic, "kernel": kernel,
            "kernel_ms": None if kernel_ms is None else round(kernel_ms, 3), "algorithmic_bytes": int(algo_bytes)}


LOAD_PHASES = ["host_parse_hex", "host_decompress_points_and_pairing_check", "hip_init_and_code_load", "small_tables_and_subgroup_check",
               "commit_table_malloc", "commit_table_build", "fk20_x_ext_fft_columns", "fk20_table_malloc", "fk20_table_build",
               "proof_table_malloc", "proof_table_build", "slots_and_host_mirror"]


def load_phases(L, hip):
    """Where the wall clock of load_trusted_setup went (ckzg_hip_load_times), seconds."""
    buf = (C.c_double * len(LOAD_PHASES))()
    k = L.load_times(C.addressof(hip.s), buf, len(LOAD_PHASES))
    return {LOAD_PHASES[i]: round(buf[i] / 1e3, 3) for i in range(k)}


class HipBuffers:
    """Page-locked and device copies of the verification rows' inputs, through torch (the process's one HIP runtime:
    loading /opt/rocm/lib/libamdhip64.so next to the copy torch ships clashes at symbol resolution)."""

    def __init__(self, torch, dev):
        self.torch, self.dev, self.keep = torch, dev, []

    def pinned(self, data):
        t = self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8).pin_memory()
        self.keep.append(t)
        return t

    def device(self, data):
        t = self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8).to(self.dev)
        self.keep.append(t)
        return t

    def h2d_rate(self, pinned_t, dev_t):
        ts = []
        for _ in range(3):
            self.torch.cuda.synchronize()
            t = time.perf_counter()
            dev_t.copy_(pinned_t, non_blocking=True)
            self.torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        return pinned_t.numel() / median(ts) / 1e9


def verify_and_recover_rows(L, hip, base):
    """verify_blob_kzg_proof_batch over 4096 blobs (this GPU's view of configs[3]: the whole batch on one
    GPU), verify_cell_kzg_proof_batch over 8192 cells, and a 256-row recover batch (configs[4]),
    timed at the C-ABI with host buffers; inputs are 8 distinct valid blobs repeated.  Kernel-level
    breakdowns of these rows: profiles/r02_*_kernel_stats.csv (tools/profile_rows.sh)."""
    ub = [base[i].tobytes() for i in range(8)]
    cm = [hip.blob_to_kzg_commitment(b) for b in ub]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(ub, cm)]
    out = {}
    sp = C.addressof(hip.s)
    n = 4096
    bb = b"".join(ub[i % 8] for i in range(n))
    cc = b"".join(cm[i % 8] for i in range(n))
    pp = b"".join(pr[i % 8] for i in range(n))
    ok = C.c_bool(False)
    for k in (512, n):
        L.verify_blobs(C.byref(ok), bb, cc, pp, k, sp)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            rc = L.verify_blobs(C.byref(ok), bb, cc, pp, k, sp)
            ts.append(time.perf_counter() - t)
        if rc != 0 or not ok.value:
            raise RuntimeError("verify_blob_kzg_proof_batch rc=%d ok=%s" % (rc, ok.value))
        out["verify_blob_kzg_proof_batch_n%d" % k] = {"blobs_per_s": round(k / median(ts), 1),
                                                      "ms": round(median(ts) * 1e3, 3), "runs": 5}
    # configs[3] in its three forms: pageable host pointers (above), page-locked host pointers (DMA'd in place,
    # chunk by chunk under the evaluation kernels), inputs resident in HBM (kernel-only time)
    row = out["verify_blob_kzg_proof_batch_n%d" % n]
    row["roofline"] = dict(roofline(ALGO_BYTES_VERIFY_BLOB * n, row["ms"], "whole call, pageable host pointers (PCIe H2D of the blobs)",
                                    ALGO_BYTES_VERIFY_BLOB * n, bound="pcie", peak=PCIE_PEAK_GBS),
                           traffic_source="by construction: every input byte crosses the link exactly once (bytes over PCIe, not HBM)")
    try:
        import torch
        hb = HipBuffers(torch, torch.device("cuda", torch.cuda.current_device()))
        pin_t = hb.pinned(bb)
        dev_t = [hb.device(x) for x in (bb, cc, pp)]
        pin = C.c_void_p(pin_t.data_ptr())
        dptr = [C.c_void_p(t.data_ptr()) for t in dev_t]
        row["pcie_h2d_pinned_GBps_measured"] = round(hb.h2d_rate(pin_t, dev_t[0]), 2)
        vb = L._fn("verify_blob_kzg_proof_batch", [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p])
        vb(C.byref(ok), pin, cc, pp, n, sp)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            rc = vb(C.byref(ok), pin, cc, pp, n, sp)
            ts.append(time.perf_counter() - t)
        if rc != 0 or not ok.value:
            raise RuntimeError("pinned verify rc=%d ok=%s" % (rc, ok.value))
        row["pinned_caller_memory"] = {"ms": round(median(ts) * 1e3, 3), "blobs_per_s": round(n / median(ts), 1),
                                       "roofline": dict(roofline(ALGO_BYTES_VERIFY_BLOB * n, median(ts) * 1e3,
                                                                 "whole call, page-locked host pointers", ALGO_BYTES_VERIFY_BLOB * n,
                                                                 bound="pcie", peak=PCIE_PEAK_GBS),
                                                        traffic_source="by construction: every input byte crosses the link exactly once")}
        L.verify_blobs_dev(C.byref(ok), dptr[0], dptr[1], dptr[2], n, sp)
        ts, ks, k0, k2 = [], [], [], []
        for _ in range(5):
            t = time.perf_counter()
            rc = L.verify_blobs_dev(C.byref(ok), dptr[0], dptr[1], dptr[2], n, sp)
            ts.append(time.perf_counter() - t)
            ks.append(L.kms(sp, 3)); k0.append(L.kms(sp, 0)); k2.append(L.kms(sp, 2))
        if rc != 0 or not ok.value:
            raise RuntimeError("resident verify rc=%d ok=%s" % (rc, ok.value))
        prow = "verify_wide" if tables_of(L, hip)["commit_wbits"] >= 16 else "verify_default"
        parts = {"k_sha256_challenges": pmc_kernel(prow, "k_sha256_challenges", streaming=True),
                 "k_eval_tree": pmc_kernel(prow, "k_eval_tree<6, 512>", streaming=True),
                 "k_table_chain_x28": pmc_kernel(prow, "k_table_chain_x28"),
                 "k_msm_accumulate_x28": pmc_kernel(prow, "k_msm_accumulate_x28")}
        have = [v for v in parts.values() if v]
        sha = parts["k_sha256_challenges"]
        row["resident_inputs"] = {"ms": round(median(ts) * 1e3, 3), "blobs_per_s": round(n / median(ts), 1),
                                  "kernel_ms": {"total": round(median(ks), 3), "validate_convert_hash_evaluate": round(median(k0), 3),
                                                "sums": round(median(k2), 3)},
                                  "roofline": dict(roofline(ALGO_BYTES_VERIFY_BLOB * n, median(ks),
                                                            "k_sha256_challenges + k_eval_tree (from the blobs' bytes) + k_validate_g1 + k_msm_accumulate_x28 over the call-time table (device time of the call)",
                                                            sum(v["traffic_bytes_per_launch"] for v in have) if have else None),
                                                   traffic_source="PMC: FETCH_SIZE (x2 for the streaming kernels) + WRITE_SIZE of the four kernels that move the call's bytes, one launch each" if have else None,
                                                   pmc_per_kernel=parts),
                                  # the call's longest kernel is a dependent chain of 2,050 SHA-256 compressions per blob on ONE wave
                                  # per 64 blobs: its bound is the issue rate of a lone wave (one VALU instruction per 4 cycles)
                                  "roofline_valu": None if not (sha and sha["valu_wave_insts"] and sha["waves"]) else {
                                      "bound": "VALU issue of one wave per SIMD (sequential chain)", "unit": "G wave-instructions/s per wave",
                                      "peak": 0.6, "kernel": "k_sha256_challenges", "kernel_ms": sha["mean_ms"],
                                      "achieved": round(sha["valu_wave_insts"] / sha["waves"] / (sha["mean_ms"] * 1e-3) / 1e9, 4),
                                      "frac": round(sha["valu_wave_insts"] / sha["waves"] / (sha["mean_ms"] * 1e-3) / 0.6e9, 4),
                                      "note": "mean over the producer (message schedule) and consumer (rounds) wave of each workgroup; peak = 2.4 GHz / 4 cycles; from " + sha["file"]}}
        del hb, pin_t, dev_t
    except Exception as e:  # noqa: BLE001 -- reported, the pageable row stands
        row["forms_error"] = str(e)
    # One rank's share of configs[3] at N = 2, 4, 8, MEASURED on this GPU: the ranks of an N-GPU job are independent and each
    # runs exactly this call -- a shard of 4096/N blobs with the host threads its share of the machine leaves it
    # ("host_threads" = cpus // N).  What the emulation cannot see is N ranks contending for host memory bandwidth.
    shares = {}
    budget = int(L._fn("ckzg_hip_host_thread_budget", [])())
    if EFFECTIVE_CORES:
        budget = min(budget, EFFECTIVE_CORES)   # what the host really delivers (parallel speed-up of this run's cpu_baseline)
    try:
        for nr in (2, 4, 8):
            hip.lib.ckzg_hip_set_option(b"host_threads", max(1, budget // nr))
            k = n // nr
            L.verify_blobs(C.byref(ok), bb, cc, pp, k, sp)
            ts = []
            for _ in range(3):
                t = time.perf_counter()
                rc = L.verify_blobs(C.byref(ok), bb, cc, pp, k, sp)
                ts.append(time.perf_counter() - t)
            if rc == 0 and ok.value:
                shares[str(nr)] = {"verify_blobs": k, "verify_ms": round(median(ts) * 1e3, 3), "host_threads": max(1, budget // nr)}
    finally:
        hip.lib.ckzg_hip_set_option(b"host_threads", 0)
    del bb
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in ub]
    n = 8192
    rows = [(i // 128) % 8 for i in range(n)]
    cols = [i % 128 for i in range(n)]
    ccm = b"".join(cm[r] for r in rows)
    idx = (C.c_uint64 * n)(*cols)
    cells = b"".join(cp[r][0][c] for r, c in zip(rows, cols))
    cprf = b"".join(cp[r][1][c] for r, c in zip(rows, cols))
    for k in (128, n):
        L.verify_cells(C.byref(ok), ccm, idx, cells, cprf, k, sp)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            rc = L.verify_cells(C.byref(ok), ccm, idx, cells, cprf, k, sp)
            ts.append(time.perf_counter() - t)
        if rc != 0 or not ok.value:
            raise RuntimeError("verify_cell_kzg_proof_batch rc=%d ok=%s" % (rc, ok.value))
        out["verify_cell_kzg_proof_batch_n%d" % k] = {"ms": round(median(ts) * 1e3, 3), "runs": 5}
    nb = 256
    keep = list(range(0, 128, 2))
    data = b"".join(b"".join(cp[b % 8][0][i] for i in keep) for b in range(nb))
    kidx = (C.c_uint64 * len(keep))(*keep)
    rc_buf = C.create_string_buffer(nb * 128 * 2048)
    rp_buf = C.create_string_buffer(nb * 128 * 48)
    args = (rc_buf, rp_buf, None, kidx, data, C.c_uint64(len(keep)), C.c_uint64(nb), C.c_void_p(sp))
    L.recover(*args)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        rc = L.recover(*args)
        ts.append(time.perf_counter() - t)
    if rc != 0 or rp_buf.raw[:128 * 48] != b"".join(cp[0][1]):
        raise RuntimeError("recover batch rc=%d or wrong proofs" % rc)
    k_small, k_fft, k_dev = L.kms(sp, 1), L.kms(sp, 4), L.kms(sp, 3)
    dom, dom_name = (k_small, "k_msm_small") if k_small >= k_fft else (k_fft, "k_g1_fft_twiddle* + k_g1_fft_addsub (2 G1 FFTs)")
    tb = tables_of(L, hip)
    fw = tb["fk20_wbits"]
    nwin_f = 2 * (127 // fw + 1)
    prow = "verify_wide" if tb["commit_wbits"] >= 16 else "verify_default"
    small_traffic = int(nb * (128 * 64 * nwin_f * (96 * (1.0 - 2.0 ** -fw) + 2) + 128 * 192))
    fftq = pmc_kernel(prow, "k_g1_fft_twiddle_quad", pick="grid")
    out["recover_cells_and_kzg_proofs_batch256"] = {
        "rows_per_s": round(nb / median(ts), 1), "ms": round(median(ts) * 1e3, 3), "runs": 3,
        "kernel_ms": {"device_section": round(k_dev, 3), "k_msm_small": round(k_small, 3), "g1_fft": round(k_fft, 3)},
        "roofline": dict(roofline(ALGO_BYTES_RECOVER_ROW * nb, dom, dom_name,
                                  small_traffic if k_small >= k_fft else (fftq["traffic_bytes_per_launch"] * 12 if fftq else None)),
                         traffic_source="analytic: 128 x 64 x nwin gathers of 96 B + digits per row (k_msm_small)" if k_small >= k_fft
                         else "PMC: 12 ladder launches of the two G1 FFTs, FETCH_SIZE + WRITE_SIZE per launch",
                         pmc_cross_check={"k_msm_small": pmc_kernel(prow, "k_msm_small"), "k_g1_fft_twiddle_quad": fftq}),
        "roofline_valu": valu_roofline(nb * 128 * 64 * nwin_f * MADS_PER_ADDITION, k_small,
                                       "k_msm_small: 128 x 64 x %d mixed additions x %d multiply-adds per row (the G1 FFT of a 256-row batch "
                                       "runs its ladders four lanes per point: latency-bound, see DESIGN)" % (nwin_f, MADS_PER_ADDITION)),
        "roofline_whole_call": dict(roofline(ALGO_BYTES_RECOVER_ROW * nb, median(ts) * 1e3, "whole call, pageable host pointers",
                                             ALGO_BYTES_RECOVER_ROW * nb, bound="pcie", peak=PCIE_PEAK_GBS),
                                    traffic_source="by construction: inputs and outputs cross the link exactly once"),
        "note": "64 of 128 cells per row (every other cell), same columns in every row; cells and proofs out"}
    # one rank's share of configs[4] at N = 2, 4, 8 (see above): a shard of 256/N rows
    for nr in (2, 4, 8):
        k = nb // nr
        a2 = (rc_buf, rp_buf, None, kidx, data, C.c_uint64(len(keep)), C.c_uint64(k), C.c_void_p(sp))
        L.recover(*a2)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            rc = L.recover(*a2)
            ts.append(time.perf_counter() - t)
        if rc == 0:
            shares.setdefault(str(nr), {}).update({"recover_rows": k, "recover_ms": round(median(ts) * 1e3, 3)})
    out["one_rank_share_emulated"] = dict(shares, note="a rank of an N-GPU job runs exactly these calls (independent shards, no collective): "
                                                       "MEASURED here on one GPU with host_threads = cpus // N; not seen: N ranks sharing host memory bandwidth")
    return out


def footprint_curve(mod, L, torch, dev, blobs, local_rank):
    """What table memory buys: commitments/s (1024 resident blobs), one-blob compute_cells_and_kzg_proofs latency
    (host pointers) and the FK20 batch rate (1024 resident blobs) at increasing window widths, each point its own
    load_trusted_setup after the previous point's tables were freed.  The 16/16/13-bit point is the main
    KZGSettings of this run (appended by the caller)."""
    pts = []
    nb = 1024
    sub = blobs[:nb]
    status = torch.empty((nb,), dtype=torch.uint8, device=dev)
    out48 = torch.empty((nb, 48), dtype=torch.uint8, device=dev)
    cells = torch.empty((nb, 128, 2048), dtype=torch.uint8, device=dev)
    proofs = torch.empty((nb, 128, 48), dtype=torch.uint8, device=dev)
    hb1 = blobs[0].cpu().numpy().tobytes()
    hc1 = C.create_string_buffer(128 * 2048)
    hp1 = C.create_string_buffer(128 * 48)
    for cw, pw, fw in ((10, 8, 8), (12, 12, 10), (13, 13, 11), (14, 14, 12)):
        t0 = time.perf_counter()
        k = mod.Kzg(mod.HIP_SO, options={"device": local_rank, "commit_wbits": cw, "proof_wbits": pw, "fk20_wbits": fw})
        load_s = time.perf_counter() - t0
        try:
            sp = C.addressof(k.s)
            fn1 = L._fn("compute_cells_and_kzg_proofs", [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p])
            L.commit_dev(out48.data_ptr(), status.data_ptr(), sub.data_ptr(), nb, sp)
            ts = []
            for _ in range(3):
                t = time.perf_counter()
                rc = L.commit_dev(out48.data_ptr(), status.data_ptr(), sub.data_ptr(), nb, sp)
                ts.append(time.perf_counter() - t)
            fn1(hc1, hp1, hb1, sp)
            t1 = []
            for _ in range(10):
                t = time.perf_counter()
                rc |= fn1(hc1, hp1, hb1, sp)
                t1.append(time.perf_counter() - t)
            L.cells_dev(cells.data_ptr(), proofs.data_ptr(), status.data_ptr(), sub.data_ptr(), nb, sp)
            t = time.perf_counter()
            rc |= L.cells_dev(cells.data_ptr(), proofs.data_ptr(), status.data_ptr(), sub.data_ptr(), nb, sp)
            tb = time.perf_counter() - t
            if rc != 0:
                raise RuntimeError("footprint point %d/%d/%d failed rc=%d" % (cw, pw, fw, rc))
            pts.append({"tables": tables_of(L, k), "load_s": round(load_s, 2),
                        "commit_blobs_per_s": round(nb / median(ts), 1),
                        "cells_and_proofs_one_blob_ms": round(median(t1) * 1e3, 3),
                        "cells_and_proofs_batch_blobs_per_s": round(nb / tb, 1)})
        finally:
            k.close()
    return pts


def predicted_scaling(value_n1, host_ptr_n1, sec, lib_budget=None, effective_cores=None):
    """What the 1/2/4/8-GPU curve of each BASELINE config should be on THIS host, from numbers measured in this run at
    N = 1 and the resource that bounds each.  Ranks are independent (no data-path collective);
    what they share is the host: its cores (hash threads, staging copies) and its memory bandwidth."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    visible = cores
    if lib_budget:
        cores = min(cores, lib_budget)      # what the library itself sizes its pools by (affinity mask)
    if effective_cores:
        cores = min(cores, effective_cores)  # what the host really delivered to the CPU baseline of this run
    MEMCPY_GBPS_PER_CORE, SHA_US_PER_BLOB_THREAD, COPY_US_PER_BLOB, GPU_SHA_US, EVAL_US, TAIL_US = 8.0, 66.0, 2.4, 3900.0, 600.0, 2000.0   # (round 6: hash on compute units of its own, evaluation from the bytes)
    out = {"host_cpus_visible": visible, "host_cores_assumed": cores,
           "host_cores_source": "min(affinity mask, the library's ckzg_hip_host_thread_budget, parallel speed-up the CPU baseline of this run reached)",
           "model": {"memcpy_GBps_per_core": MEMCPY_GBPS_PER_CORE, "sha256_us_per_blob_per_thread": SHA_US_PER_BLOB_THREAD,
                     "pcie_us_per_blob": COPY_US_PER_BLOB, "gpu_sha_us": GPU_SHA_US, "threads_per_rank": "cpus // N (ckzg_hip_host_thread_budget)"},
           "by_config": {}}
    ns = (1, 2, 4, 8)
    thr = {n: max(1, cores // n) for n in ns}
    out["by_config"]["configs[1] commitments, inputs resident (value)"] = {
        "bound": "GPU integer VALU (k_msm_accumulate); nothing is shared between ranks", "unit": "blobs/s",
        "predicted": {str(n): round(n * value_n1, 0) for n in ns}}
    if host_ptr_n1:
        out["by_config"]["configs[1] commitments, pageable host pointers"] = {
            "bound": "per rank min(GPU + PCIe pipeline, staging memcpy on min(8, threads) cores)", "unit": "blobs/s",
            "predicted": {str(n): round(n * min(host_ptr_n1, min(8, thr[n]) * MEMCPY_GBPS_PER_CORE * 1e9 / 131072.0), 0) for n in ns}}
    try:
        one = sec["cells_and_proofs"]["one_blob"]["calls_per_s"]
        out["by_config"]["configs[2] compute_cells_and_kzg_proofs, one blob per call"] = {
            "bound": "latency of one call (replicas only: one caller per GPU)", "unit": "calls/s",
            "predicted": {str(n): round(n * one, 1) for n in ns}}
    except (KeyError, TypeError):
        pass
    try:
        m1 = sec["verify_blob_kzg_proof_batch_n4096"]["ms"]
        pred = {}
        emu = sec.get("one_rank_share_emulated") or {}
        for n in ns:
            if n == 1:
                pred["1"] = round(4096 / (m1 * 1e-3), 0)
                continue
            if emu.get(str(n), {}).get("verify_ms"):
                pred[str(n)] = round(4096 / (emu[str(n)]["verify_ms"] * 1e-3), 0)   # every rank takes this long for its shard
                continue
            per = 4096 // n
            t = min(32, thr[n])
            host_us, copy_us = per * SHA_US_PER_BLOB_THREAD / t, per * COPY_US_PER_BLOB
            us = (max(copy_us, host_us) if host_us <= copy_us + GPU_SHA_US else copy_us + GPU_SHA_US + EVAL_US) + TAIL_US
            pred[str(n)] = round(4096 / (us * 1e-6), 0)
        out["by_config"]["configs[3] verify_blob_kzg_proof_batch, 4096 blobs sharded"] = {
            "bound": "per shard: host SHA-256 on cpus//N threads or (automatic below ~1 thread per 30 blobs) the GPU hash's fixed 4.9 ms, "
                     "+ ~2 ms of sums and pairing per shard; N = 1 is this run's measurement, N > 1 the measured time of one rank's shard "
                     "(one_rank_share_emulated) when present", "unit": "blobs/s", "predicted": pred}
    except (KeyError, TypeError):
        pass
    try:
        m1 = sec["recover_cells_and_kzg_proofs_batch256"]["ms"]
        floor = 7.0
        emu = sec.get("one_rank_share_emulated") or {}
        out["by_config"]["configs[4] recover_cells_and_kzg_proofs, 256 rows sharded"] = {
            "bound": "latency floor of a small shard (the four to six dependent ladder launches of FK20 + the recovery transforms); N > 1: "
                     "the measured time of one rank's shard (one_rank_share_emulated) when present", "unit": "rows/s",
            "predicted": {str(n): round(256 / ((emu[str(n)]["recover_ms"] if n > 1 and emu.get(str(n), {}).get("recover_ms")
                                                else floor + max(0.0, m1 - floor) / n) * 1e-3), 0) for n in ns}}
    except (KeyError, TypeError):
        pass
    return out


def committed_single_gpu_value():
    """The newest committed N = 1 line (profiles/rNN_final_bench_lines.json): what an N-rank run is predicted from."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_bench_line*.json")), reverse=True):
        try:
            d = json.load(open(f))
            d = d[0] if isinstance(d, list) else d
            d = d.get("default_command_python_bench_py", d)   # (tools/collect_profiles.py keeps two lines per file)
            if d.get("n_gpus") == 1 and d.get("value"):
                return float(d["value"]), os.path.relpath(f, ROOT)
        except Exception:  # noqa: BLE001
            continue
    return None, None


def pin_to_gpu_numa_node(torch, local_rank):
    """A rank's host threads (staging copies, transcript hashing) belong on the NUMA node its GPU hangs off:
    sysfs gives the node of the PCI function and the node's CPU list.  Returns what was done, for the JSON line."""
    try:
        props = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return {"pci": bdf, "numa_node": node, "pinned": False}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "cpus": len(cpus), "pinned": True}
    except Exception as e:  # noqa: BLE001 -- containers often hide sysfs; never fatal
        return {"pinned": False, "error": str(e)[:120]}


def sharded_rows(L, hip, mod, torch, dist, rank, world, red_dev, base):
    """BASELINE configs[3] / configs[4] in their multi-GPU form (tools/run_sharded.py, c-kzg-4844_amd/multi_gpu.py):
    4096 blobs / 256 rows in contiguous shards, one per rank, no data-path collective; the verdict is an AND over
    the shards, the recovered proofs are all-gathered.  Every rank runs this; rank 0 reports.  Timed between
    barriers, MAX over ranks."""
    import importlib
    mg = importlib.import_module("ckzg_4844_amd.multi_gpu")
    sp = C.addressof(hip.s)
    ub = [base[i].tobytes() for i in range(8)]
    cm = [hip.blob_to_kzg_commitment(b) for b in ub]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(ub, cm)]
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in ub]
    n_verify, n_rows = 4096, 256
    lo, hi = mg.shard_bounds(n_verify, world)[rank]
    bb = b"".join(ub[i % 8] for i in range(lo, hi))
    cc = b"".join(cm[i % 8] for i in range(lo, hi))
    pp = b"".join(pr[i % 8] for i in range(lo, hi))
    ok = C.c_bool(False)

    def verify(a, b):
        rc = L.verify_blobs(C.byref(ok), bb, cc, pp, b - a, sp)
        return rc, ok.value

    def timed(fn):
        fn()   # warm-up: arenas, pinned staging
        dist.barrier()
        t = time.perf_counter()
        res = fn()
        dt = time.perf_counter() - t
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return res, float(tt.item())

    (rc, verdict), dt_v = timed(lambda: mg.sharded_verify(verify, n_verify, red_dev))
    keep = list(range(0, 128, 2))
    rlo, rhi = mg.shard_bounds(n_rows, world)[rank]
    data = b"".join(b"".join(cp[b % 8][0][i] for i in keep) for b in range(rlo, rhi))
    kidx = (C.c_uint64 * len(keep))(*keep)
    nloc = rhi - rlo
    rp_buf = C.create_string_buffer(max(nloc, 1) * 128 * 48)

    def recover(a, b):
        r = L.recover(None, rp_buf, None, kidx, data, C.c_uint64(len(keep)), C.c_uint64(b - a), C.c_void_p(sp))
        if r != 0:
            raise RuntimeError("sharded recover rc=%d" % r)
        return torch.frombuffer(bytearray(rp_buf.raw[:(b - a) * 128 * 48]), dtype=torch.uint8).reshape(b - a, 128 * 48)

    proofs, dt_r = timed(lambda: mg.sharded_map(recover, n_rows, 128 * 48, red_dev))
    good = all(bytes(proofs[b].cpu().numpy().tobytes()) == b"".join(cp[b % 8][1]) for b in (0, n_rows // 2, n_rows - 1))
    return {"verify_blob_kzg_proof_batch_n4096_sharded": {"ranks": world, "rc": rc, "ok": verdict, "ms": round(dt_v * 1e3, 3),
                                                          "blobs_per_s": round(n_verify / dt_v, 1),
                                                          "collective": "MAX/AND all-reduce of (return code, verdict): 8 bytes"},
            "recover_cells_and_kzg_proofs_batch256_sharded": {"ranks": world, "proofs_correct": good, "ms": round(dt_r * 1e3, 3),
                                                              "rows_per_s": round(n_rows / dt_r, 1),
                                                              "collective": "all-gather of 6,144 B of proofs per row"}}


def concurrency_rows(mod, hip, blobs_u8, seconds=0.4, threads=(1, 8, 32, 128, 256)):
    """The reference API is one blob per call; its parallel shape is N threads each calling it on a shared
    KZGSettings (bindings/go/main_test.go:953-971).  N native threads (fanout.py -> libckzg_callers.so: plain C
    against ckzg.h) call the UNCHANGED blob_to_kzg_commitment / compute_cells_and_kzg_proofs / verify_blob_kzg_proof for a fixed time;
    the library coalesces them into batch launches (csrc/combiner.hpp).  Per thread count: calls/s, mean and worst
    call latency, launches and mean units per batch launch (ckzg_hip_coalesce_stats)."""
    fo = mod.fanout
    ub = [blobs_u8[i].tobytes() for i in range(32)]
    out = {"driver": "libckzg_callers.so: pthreads, one blob and one output buffer per thread, %.1f s per row" % seconds,
           "coalescing": "on (csrc/combiner.hpp; 2 launches in flight per operation)"}
    cm = [hip.blob_to_kzg_commitment(b) for b in ub]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(ub, cm)]
    for name, op, idx in (("blob_to_kzg_commitment", fo.OP_COMMIT, 0), ("compute_cells_and_kzg_proofs", fo.OP_CELLS_PROOFS, 3),
                          ("verify_blob_kzg_proof", fo.OP_VERIFY_BLOB, 6)):
        rows = {}
        for nt in threads:
            ins = [ub[t % 32] for t in range(nt)]
            aux = [cm[t % 32] + pr[t % 32] for t in range(nt)] if op == fo.OP_VERIFY_BLOB else None
            # warm-up: arenas, page-locked batch buffers -- a crowd touches all eight stream slots and up to six batch
            # buffers per operation before the steady state the row is about
            fo.run(hip, mod.HIP_SO, op, ins, seconds=0.5 if nt >= 128 else 0.2, aux=aux)
            before = fo.coalesce_stats(hip, idx)
            st, rets, _ = fo.run(hip, mod.HIP_SO, op, ins, seconds=seconds, aux=aux)
            after = fo.coalesce_stats(hip, idx)
            row = {"calls_per_s": round(st["calls_per_s"], 1), "mean_call_ms": round(st["mean_call_ms"], 3),
                   "p50_call_ms": round(st["p50_call_ms"], 3), "p99_call_ms": round(st["p99_call_ms"], 3),
                   "worst_call_ms": round(st["worst_call_ms"], 3), "failed_calls": st["not_ok"] + sum(1 for r in rets if r != 0)}
            if before and after:
                d = {k: after[k] - before[k] for k in ("solo", "batches", "batched", "run_us")}
                row["launches"] = d["solo"] + d["batches"]
                row["mean_units_per_batch_launch"] = round(d["batched"] / d["batches"], 1) if d["batches"] else None
                row["mean_batch_launch_ms"] = round(d["run_us"] / d["batches"] / 1e3, 3) if d["batches"] else None
            rows[str(nt)] = row
        out[name] = rows
    return out


EFFECTIVE_CORES = None   # set by main() from the cpu_baseline of the run, before the secondary rows

LAST_LINE_BUDGET = 4000   # characters; the driver keeps an ~8 KB stdout tail and parses the LAST line (round 4's 25 KB line was cut: parsed = null)


def compact_line(full, side_file=None):
    """The LAST stdout line: the contract's keys, `roofline`, `roofline_valu`, `cpu_baseline` and the BASELINE configs by
    name -- nothing else.  Sweeps, curves, per-row counters and the scaling MODEL go to `side_file` and to an earlier
    stdout line.  Pure function of the full record (tests/test_bench_line.py feeds it the committed 25 KB line)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: full.get(k) for k in keep}
    cfg = full.get("config") or {}
    tabs = cfg.get("tables") or {}
    out["config"] = {"workload": cfg.get("workload"), "blobs_per_step_per_gpu": cfg.get("blobs_per_step_per_gpu"),
                     "table_wbits": cfg.get("table_wbits"),
                     "tables": "wide" if (cfg.get("table_wbits") or 0) >= 16 else "narrower than the 16-bit headline set",
                     "tables_gb": round(tabs["bytes"] / 1e9, 1) if isinstance(tabs, dict) and tabs.get("bytes") else None,
                     "parallelism": cfg.get("parallelism")}
    r = full.get("roofline") or {}
    pmc = r.get("pmc_cross_check") or {}
    out["roofline"] = {"bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
                       "frac": r.get("frac"), "traffic": pmc.get("traffic_bytes_per_launch"),
                       "traffic_source": pmc.get("traffic_file"), "traffic_analytic": r.get("traffic"),
                       "kernel": r.get("kernel"), "kernel_ms": r.get("kernel_ms"),
                       "algorithmic_bytes_per_launch": r.get("algorithmic_bytes")}
    v = full.get("roofline_valu") or {}
    out["roofline_valu"] = {k: v.get(k) for k in ("bound", "unit", "peak", "achieved", "frac")}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        ac = cb.get("all_cores") or {}
        out["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": (cb.get("sample") or "")[:160],
                               "all_cores": {"value": ac.get("value"), "cores": ac.get("cores"), "driver": ac.get("driver")},
                               "compute_cells_and_kzg_proofs_ms_per_call": cb.get("compute_cells_and_kzg_proofs_ms_per_call")}
    elif cb is not None:
        out["cpu_baseline"] = cb if len(json.dumps(cb)) < 300 else {"error": str(cb)[:200]}
    out["value_host_pointer"] = full.get("value_host_pointer")
    out["value_semantics"] = "value: inputs resident in HBM; value_host_pointer: pageable host memory, H2D+D2H timed (SURVEY 8d)"
    out["parity_spot_check_vs_oracle"] = full.get("parity_spot_check_vs_oracle")
    bc = full.get("baseline_configs")
    if isinstance(bc, dict):
        c2 = dict(bc.get("configs[2]") or {})
        c2.pop("note", None)
        out["baseline_configs"] = {"configs[1]": bc.get("configs[1]"), "configs[2]": c2,
                                   "configs[3]_verify_4096_ms": bc.get("configs[3]"),
                                   "configs[4]_recover_256_ms": bc.get("configs[4]"),
                                   "tables": "configs[2] default = library default tables (5 GB); every other figure: wide tables (238 GB)"}
    for k in ("predicted", "per_rank"):   # multi-rank runs: small, and what the driver's scaling check may want
        if full.get(k) is not None and len(json.dumps(full[k])) < 1200:
            out[k] = full[k]
    if isinstance(full.get("secondary"), dict) and "error" in full["secondary"]:
        out["secondary_error"] = str(full["secondary"]["error"])[:200]
    out["secondary_file"] = side_file
    text = json.dumps(out)
    if len(text) > LAST_LINE_BUDGET:   # never let an unexpected field push the head of the line out of the driver's tail
        for k in ("per_rank", "predicted", "baseline_configs", "parity_spot_check_vs_oracle"):
            out.pop(k, None)
            if len(json.dumps(out)) <= LAST_LINE_BUDGET:
                break
    return out


def emit(full):
    """Full record -> gpurun_out/bench_secondary.json and one stderr line (written first); the compact record is the ONE
    line on stdout."""
    side = os.path.join("gpurun_out", "bench_secondary.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, side), "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        sys.stderr.write("bench: could not write %s: %s\n" % (side, e))
        side = None
    sys.stderr.write(json.dumps({"bench_full_record": full}) + "\n")
    sys.stderr.flush()
    sys.stdout.write(json.dumps(compact_line(full, side)) + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--wbits", type=int, default=int(os.environ.get("CKZG_BENCH_WBITS", str(WIDE["commit_wbits"]))))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn(args.gpus)  # does not return

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d\n" % (args.gpus, world, world))
    one_gpu = bool(os.environ.get("CKZG_BENCH_ONE_GPU"))
    if not one_gpu and torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench: rank %d needs device %d but only %d visible" % (rank, local_rank, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  CKZG_BENCH_BACKEND=gloo + CKZG_BENCH_ONE_GPU=1 exist only to exercise
        # this file's multi-rank control flow on a one-GPU box (all ranks share device 0).
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    numa = pin_to_gpu_numa_node(torch, local_rank) if world > 1 else None
    if world > 1:
        # RCCL comes up BEFORE the tables are allocated: its buffers and hardware queues are claimed while the HBM
        # is still empty, and the line records what was free on every rank before and after the load.
        # The ranks only meet at barriers and at one MAX over their timings (the path has no data-path collective,
        # SURVEY 8e).  RCCL carries them; if it cannot come up on this node the run FAILS (value null, exit code 3).
        backend = os.environ.get("CKZG_BENCH_BACKEND", "nccl")
        try:
            dist.init_process_group(backend, rank=rank, world_size=world)
            t = torch.zeros(1, device=dev if backend == "nccl" else torch.device("cpu"))
            dist.all_reduce(t)
            if backend == "nccl":
                torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            # An N-GPU number timed over anything but RCCL would not be the number asked for: no silent fallback.
            # Rank 0 prints a line whose value is null, every rank exits non-zero.
            sys.stderr.write("bench: rank %d: %s did not come up: %s\n" % (rank, backend, str(e)[:300]))
            if rank == 0:
                print(json.dumps({"metric": "blob_to_kzg_commitment throughput", "value": None, "unit": "blobs/s", "n_gpus": world,
                                  "error": "%s (RCCL) initialisation failed: %s" % (backend, str(e)[:300])}))
            sys.exit(3)
    red_dev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")

    import __graft_entry__ as ge
    mod = ge.load_package()
    # the ONE wide-table load of the run: commitment, low-latency proof and FK20 tables together (GLV
    # half-scalar tables: 103 + 103 + 32 GB at 16 / 16 / 13 bits); the library narrows what does not fit
    # Progressive widening ("async_tables"): the load returns with the library's default-width tables (~0.4 s), a
    # first commitment is served at once, and the wide tables are built in the background; the timed region below
    # starts only after ckzg_hip_wait_tables.
    opts = dict(WIDE, device=local_rank, commit_wbits=args.wbits, async_tables=1)
    if one_gpu and world > 1:
        opts.update(commit_wbits=min(args.wbits, 12), proof_wbits=8, fk20_wbits=8)  # ranks share one GPU's HBM
    import hashlib
    first_blob = b"".join(b"\x00" + hashlib.sha256(b"first%d" % j).digest()[:31] for j in range(4096))
    free_before = torch.cuda.mem_get_info(local_rank)[0]
    t_load = time.perf_counter()
    hip = mod.Kzg(mod.HIP_SO, options=opts)
    returned_s = time.perf_counter() - t_load
    hip.lib.ckzg_hip_set_option(b"async_tables", 0)   # later loads of this process are ordinary ones
    # (round 4 switched "commit_graph" off here for multi-rank runs: RCCL's watchdog thread makes HIP calls of its own and
    # a stream capture could not be kept away from it.  The graph is built node by node now -- nothing to switch off.)
    first_commitment = hip.blob_to_kzg_commitment(first_blob)
    first_commit_s = time.perf_counter() - t_load
    L = Lib(hip.lib)
    sp = C.addressof(hip.s)
    if L.wait_tables(sp) != 0:
        raise SystemExit("bench: ckzg_hip_wait_tables failed")
    load_s = time.perf_counter() - t_load
    if hip.blob_to_kzg_commitment(first_blob) != first_commitment:
        raise SystemExit("bench: commitment from the wide tables differs from the one served while they were built")
    free_after = torch.cuda.mem_get_info(local_rank)[0]
    wbits = int(L.table_wbits(sp, 0))  # what was actually built

    # synthetic blobs: 31 random bytes per field element, top byte 0 => canonical
    # (same distribution as bindings/go/main_test.go:31-51), fixed seed per rank
    g = torch.Generator(device=dev)
    g.manual_seed(0xC4B64844 + rank)
    blobs = torch.randint(0, 256, (BLOBS_PER_STEP, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
    blobs[:, :, 0] = 0
    out = torch.empty((BLOBS_PER_STEP, 48), dtype=torch.uint8, device=dev)
    status = torch.empty((BLOBS_PER_STEP,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    mg = None
    if world > 1:
        import importlib
        mg = importlib.import_module("ckzg_4844_amd.multi_gpu")
    gathered = [None]

    def step():
        rc = L.commit_dev(out.data_ptr(), status.data_ptr(), blobs.data_ptr(), BLOBS_PER_STEP, sp)
        if rc != 0:
            raise RuntimeError("commit batch failed rc=%d" % rc)
        if mg is not None:
            # north_star's "trivial RCCL gather over xGMI": every step's 48-byte commitments go to rank 0 INSIDE the timed
            # region (49 KB per rank and step; enqueued behind the finished batch, the next step starts underneath it)
            gathered[0] = mg.gather_to_rank0(out.clone())   # (a copy: the next step overwrites `out` while the gather may be in flight)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        step()
        kern_ms.append(L.kms(sp, 1))   # hipEvents around k_msm_accumulate on the library's stream
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank == 0:   # what rank 0 holds after the last step: its own shard first, every rank's rows behind it
            g0 = gathered[0]
            if g0 is None or tuple(g0.shape) != (world * BLOBS_PER_STEP, 48) or \
                    not torch.equal(g0[:BLOBS_PER_STEP].cpu(), out.cpu()):
                raise SystemExit("bench: the gathered commitments on rank 0 are not the ranks' outputs")

    # the reference-shaped call: pageable host pointers, H2D and D2H inside the timed region, same step count
    host_ptr = None
    if not args.no_pcie:
        hb = blobs.cpu().numpy().tobytes()
        ho = C.create_string_buffer(48 * BLOBS_PER_STEP)
        hs = C.create_string_buffer(BLOBS_PER_STEP)
        for _ in range(max(1, min(args.warmup, 2))):
            rc = L.commit_host(ho, hs, hb, BLOBS_PER_STEP, sp)  # warm-up: pinned staging, buffers
        if world > 1:
            dist.barrier()
        ho_t = torch.frombuffer(ho, dtype=torch.uint8).reshape(BLOBS_PER_STEP, 48) if mg is not None else None
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rc = L.commit_host(ho, hs, hb, BLOBS_PER_STEP, sp)
            if mg is not None:   # the same gather as in the resident leg, from the host results
                mg.gather_to_rank0(ho_t.to(red_dev))
        if mg is not None and red_dev.type == "cuda":
            torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        if rc != 0:
            raise SystemExit("bench: host-pointer commitment batch failed rc=%d" % rc)
        if ho.raw != out.cpu().numpy().tobytes():
            raise SystemExit("bench: host-pointer and device-pointer paths disagree")
        if world > 1:
            dist.barrier()
            t = torch.tensor([dth], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dth = float(t.item())
        # the same call on page-locked caller memory (hipHostMalloc / hipHostRegister): DMA'd from directly
        pinned_rate = None
        try:
            hpin = blobs.cpu().pin_memory()
            L.commit_host(ho, hs, C.cast(hpin.data_ptr(), C.c_char_p), BLOBS_PER_STEP, sp)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                rc = L.commit_host(ho, hs, C.cast(hpin.data_ptr(), C.c_char_p), BLOBS_PER_STEP, sp)
            dtp = time.perf_counter() - t1
            if rc == 0 and ho.raw == out.cpu().numpy().tobytes():
                pinned_rate = round(BLOBS_PER_STEP * args.steps / dtp, 2)
        except Exception as e:  # reported as null
            sys.stderr.write("bench: pinned host-pointer leg failed: %s\n" % e)
        host_ptr = {"value": round(BLOBS_PER_STEP * args.steps * world / dth, 2), "unit": "blobs/s",
                    "pinned_caller_memory_blobs_per_s_this_rank": pinned_rate,
                    "ms_per_step": round(dth / args.steps * 1e3, 3), "steps": args.steps,
                    "note": "ckzg_hip_blob_to_kzg_commitment_batch on pageable host memory: staging copy, H2D, "
                            "kernels and D2H inside the timed region (SURVEY 8d form); whole job over all ranks"}

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

    # the other rows of the path, all from the settings loaded above plus ONE co-resident default-table load
    secondary = None

    def secondary_rows():
        sec = {"load_trusted_setup_s": {"wide_tables": round(load_s, 2), "load_call_returned_after": round(returned_s, 3),
                                        "time_to_first_commitment": round(first_commit_s, 3),
                                        "mode": "async_tables: default-width tables first, wide tables built in the background "
                                                "(wide_tables = until ckzg_hip_wait_tables returned)",
                                        "wide_tables_phases": load_phases(L, hip)}}
        sec["cells_and_proofs"] = cells_rows(L, hip, torch, dev, blobs, "wide tables (same KZGSettings as the headline)")
        try:
            sec.update(verify_and_recover_rows(L, hip, blobs[:8].cpu().numpy()))
        except Exception as e:  # reported, never fatal for the headline
            sec["verify_recover_error"] = str(e)
        try:
            sec["concurrent_callers"] = concurrency_rows(mod, hip, blobs[:32].cpu().numpy())
        except Exception as e:
            sec["concurrent_callers"] = {"error": str(e)}
        # default footprint: what a caller gets from load_trusted_setup(precompute=0) without any option
        t1 = time.perf_counter()
        small = mod.Kzg(mod.HIP_SO, options={"device": local_rank, "commit_wbits": 10, "proof_wbits": 8, "fk20_wbits": 0})
        sec["load_trusted_setup_s"]["default_tables"] = round(time.perf_counter() - t1, 2)
        sec["load_trusted_setup_s"]["default_tables_phases"] = load_phases(L, small)
        try:
            sps = C.addressof(small.s)
            L.commit_dev(out.data_ptr(), status.data_ptr(), blobs.data_ptr(), BLOBS_PER_STEP, sps)
            ts, ks = [], []
            for _ in range(5):
                t1 = time.perf_counter()
                rc = L.commit_dev(out.data_ptr(), status.data_ptr(), blobs.data_ptr(), BLOBS_PER_STEP, sps)
                ts.append(time.perf_counter() - t1)
                ks.append(L.kms(sps, 1))
            if rc != 0:
                raise RuntimeError("default-table commit rc=%d" % rc)
            d = {"tables": tables_of(L, small),
                 "commit_blobs_per_s": round(BLOBS_PER_STEP / median(ts), 1),
                 "commit_roofline": dict(roofline(ALGO_BYTES_PER_BLOB * BLOBS_PER_STEP, median(ks), "k_msm_accumulate",
                                                  analytic_traffic(int(L.table_wbits(sps, 0)), BLOBS_PER_STEP)),
                                         traffic_source="analytic (bench.py: analytic_traffic)"),
                 "commit_roofline_valu": valu_roofline(BLOBS_PER_STEP * 2 * (127 // int(L.table_wbits(sps, 0)) + 1) * 4096 * MADS_PER_ADDITION,
                                                       median(ks), "nwin x 4096 mixed additions per blob")}
            d["cells_and_proofs"] = cells_rows(L, small, torch, dev, blobs, "default tables")
            try:
                d["concurrent_callers"] = concurrency_rows(mod, small, blobs[:32].cpu().numpy())
            except Exception as e:
                d["concurrent_callers"] = {"error": str(e)}
            sec["default_footprint"] = d
        finally:
            small.close()
        return sec

    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu_base = cpu_baseline()
            global EFFECTIVE_CORES
            EFFECTIVE_CORES = max(1, int(round(cpu_base["all_cores"]["value"] / cpu_base["value"])))
        except Exception as e:  # the oracle is only a reported baseline
            cpu_base = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            secondary = secondary_rows()
        except BaseException as e:  # noqa: BLE001 -- report, keep the headline
            if isinstance(e, SystemExit) and ("disagree" in str(e) or "differs" in str(e)):
                raise
            secondary = {"error": "%s: %s" % (type(e).__name__, e)}

    # per-rank bookkeeping and the sharded forms of configs[3] / configs[4]: every rank takes part, rank 0 reports
    per_rank = sharded = None
    if world > 1:
        t = torch.tensor([load_s, free_before / 1e9, free_after / 1e9, sum(kern_ms) / len(kern_ms)], dtype=torch.float64, device=red_dev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        per_rank = [{"rank": r, "load_trusted_setup_s": round(float(q[0]), 2), "free_hbm_gb_before_load": round(float(q[1]), 1),
                     "free_hbm_gb_after_load": round(float(q[2]), 1), "k_msm_accumulate_ms": round(float(q[3]), 3)}
                    for r, q in enumerate(parts)]
        if not args.no_secondary:
            try:
                sharded = sharded_rows(L, hip, mod, torch, dist, rank, world, red_dev, blobs[:8].cpu().numpy())
            except BaseException as e:  # noqa: BLE001 -- every rank fails alike or the barrier inside times out
                sharded = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    # C-ABI multi-device form (one process, ckzg_hip_set_option("devices", mask), host threads fan the batch
    # out): measured by rank 0 after the timed region while the other ranks wait, on small tables that fit
    # next to every rank's wide ones
    fan_out = None
    if world > 1 and not args.no_secondary:
        if rank == 0:
            try:
                ndev = torch.cuda.device_count()
                # (the one-GPU control-flow test stands in for the devices with table replicas on device 0)
                fan = {"replicas": min(world, 8), "device": 0} if one_gpu else \
                      {"devices": (1 << world) - 1 if world <= ndev else -1}
                multi = mod.Kzg(mod.HIP_SO, options=dict(fan, commit_wbits=10, proof_wbits=0, fk20_wbits=0))
                hip.lib.ckzg_hip_set_option(b"replicas", 1)
                hip.lib.ckzg_hip_set_option(b"devices", 0)
                spm = C.addressof(multi.s)
                nd = int(L.num_devices(spm))
                n = BLOBS_PER_STEP * nd
                hb = blobs.cpu().numpy().tobytes() * nd
                ho = C.create_string_buffer(48 * n)
                hs = C.create_string_buffer(n)
                L.commit_host(ho, hs, hb, n, spm)
                t1 = time.perf_counter()
                rc = L.commit_host(ho, hs, hb, n, spm)
                dtm = time.perf_counter() - t1
                multi.close()
                fan_out = {"devices": nd, "blobs": n, "blobs_per_s": round(n / dtm, 1), "rc": rc, "table_wbits": 10,
                           "note": "one process, ckzg_hip_blob_to_kzg_commitment_batch fanning contiguous ranges over "
                                   "all devices from host threads (host pointers, PCIe inclusive)"}
            except BaseException as e:  # noqa: BLE001
                fan_out = {"error": "%s: %s" % (type(e).__name__, e)}
        dist.barrier()

    if rank == 0:
        total_blobs = BLOBS_PER_STEP * args.steps * world
        value = total_blobs / dt
        avg_k = sum(kern_ms) / len(kern_ms)
        adds_per_blob = (2 * (127 // wbits + 1)) * 4096
        line = {
            "metric": "blob_to_kzg_commitment throughput",
            "value": round(value, 2), "unit": "blobs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "blob_to_kzg_commitment batch of 1024 blobs per GPU (4096-point G1 MSM per blob), "
                                   "inputs resident in HBM" + ("" if world == 1 else
                                                               ", every step's commitments gathered to rank 0 inside the timed region"),
                       "blobs_per_step_per_gpu": BLOBS_PER_STEP,
                       "table_wbits": wbits, "tables": tables_of(L, hip),
                       "parallelism": "independent blob shards per GPU" + (
                           ", no collective" if world == 1 else
                           "; one gather of 48 B per blob to rank 0 per step (torch.distributed.gather), no other data-path collective"),
                       "barrier_backend": dist.get_backend() if world > 1 else None},
            "roofline": dict(roofline(ALGO_BYTES_PER_BLOB * BLOBS_PER_STEP, avg_k, "k_msm_accumulate",
                                      analytic_traffic(wbits, BLOBS_PER_STEP)),
                             traffic_source="analytic: nwin*4096 table gathers of 96 B + int16 digits per blob, from the "
                                            "table geometry of THIS run (bench.py: analytic_traffic)",
                             pmc_cross_check=pmc_cross_check(),
                             note="integer-VALU-bound kernel (v_mad_u64_u32 chains); HBM fraction is small by nature"),
            # the physical bound of this kernel: integer multiply-add issue rate.  peak = measured
            # v_mad_u64_u32 rate of the chip (tools/ubench/instr_rates.hip: 32.9e12 lane-ops/s);
            # achieved counts only the multiply-adds of the field products of each table addition
            # (MADS_PER_ADDITION; nwin*4096 additions per blob), not the ~25 % of other instructions.
            "roofline_valu": {"bound": "v_mad_u64_u32 issue", "unit": "T lane-mad/s", "peak": 32.9,
                              "achieved": round(BLOBS_PER_STEP * adds_per_blob * MADS_PER_ADDITION / (avg_k * 1e-3) / 1e12, 3),
                              "frac": round(BLOBS_PER_STEP * adds_per_blob * MADS_PER_ADDITION / (avg_k * 1e-3) / 32.9e12, 4),
                              "pmc": "roofline.pmc_cross_check (SQ_INSTS_VALU of the newest committed profile)"},
            # `value` is the figure the bench contract defines: whole-job throughput with the inputs already resident in
            # HBM when the timed region starts; the figure of SURVEY 8(d) -- pageable host pointers, staging copy, H2D
            # and D2H inside the timed region, what a caller of the reference-shaped API gets -- stands beside it
            "value_semantics": "value = inputs resident in HBM (bench contract); value_host_pointer = the same work from "
                               "pageable host memory, H2D + D2H inside the timed region (SURVEY 8d)",
            "value_resident": round(value, 2),
            "value_host_pointer": None if host_ptr is None else host_ptr["value"],
            "host_pointer": host_ptr,
            "pcie_inclusive_blobs_per_s": None if host_ptr is None else host_ptr["value"],
            "parity_spot_check_vs_oracle": parity,
            "secondary": secondary,
        }
        if fan_out is not None:
            line["c_abi_fan_out"] = fan_out
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["numa"] = numa
        if sharded is not None:
            line["sharded_rows"] = sharded
        if cpu_base is not None:
            line["cpu_baseline"] = cpu_base
        # the 1/2/4/8-GPU curve this host should give, per BASELINE config, and what bounds it
        if world == 1:
            eff = None
            try:
                cb = line["cpu_baseline"]
                eff = max(1, int(round(cb["all_cores"]["value"] / cb["value"])))
            except (KeyError, TypeError, ZeroDivisionError):
                pass
            budget = L._fn("ckzg_hip_host_thread_budget", [])()
            line["predicted_scaling"] = predicted_scaling(value, None if host_ptr is None else host_ptr["value"], secondary or {},
                                                          lib_budget=int(budget), effective_cores=eff)
        else:
            v1, src = committed_single_gpu_value()
            line["predicted"] = {"value": None if v1 is None else round(world * v1, 0), "unit": "blobs/s",
                                 "from": "N x the committed one-GPU value (%s): ranks share nothing on this path" % src,
                                 "bound": "GPU integer VALU per rank"}
        if isinstance(secondary, dict) and "cells_and_proofs" in secondary:
            # BASELINE configs by name, so that a reader does not have to know which row is which
            dflt = secondary.get("default_footprint", {}).get("cells_and_proofs", {}).get("one_blob", {})
            line["baseline_configs"] = {
                "configs[1]": {"blobs_per_s_resident": line["value"], "blobs_per_s_host_pointers": line["value_host_pointer"]},
                "configs[2]": {"precompute": 0,
                               "library_default_tables_ms_per_call": dflt.get("ms_per_call"),
                               "wide_tables_ms_per_call": secondary["cells_and_proofs"]["one_blob"]["ms_per_call"],
                               "note": "load_trusted_setup(..., precompute = 0) with no option set gives the default-table "
                                       "figure (5 GB of tables); the wide figure needs commit/proof/FK20 widths 16/16/13 (238 GB)"},
                "configs[3]": secondary.get("verify_blob_kzg_proof_batch_n4096", {}).get("ms"),
                "configs[4]": secondary.get("recover_cells_and_kzg_proofs_batch256", {}).get("ms")}
    tables_main = tables_of(L, hip) if rank == 0 else None
    hip.close()
    if rank == 0 and world == 1 and not args.no_secondary and isinstance(secondary, dict) and "error" not in secondary:
        # after the 238 GB are released: what narrower tables give (each point is its own load)
        try:
            curve = footprint_curve(mod, L, torch, dev, blobs, local_rank)
            curve.append({"tables": tables_main, "commit_blobs_per_s": line["value"],
                          "cells_and_proofs_one_blob_ms": secondary["cells_and_proofs"]["one_blob"]["ms_per_call"],
                          "cells_and_proofs_batch_blobs_per_s": secondary["cells_and_proofs"]["batch_2048"]["blobs_per_s"],
                          "note": "the main KZGSettings of this run (batch of 2048)"})
            secondary["footprint_curve"] = curve
        except BaseException as e:  # noqa: BLE001
            secondary["footprint_curve"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if rank == 0:
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
