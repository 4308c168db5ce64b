"""Runs under LD_PRELOAD=failalloc.so (tests/test_gpu_alloc_failures.py starts it): every device / page-locked
allocation of every entry point is made to fail in turn.  What must hold each time (the reference's error
convention, src/common/ret.h:24-29): the call returns C_KZG_MALLOC -- or succeeds with the right bytes, where the
library has a smaller fallback --, never another code, never a crash; nothing is leaked; the same call right after
is fine.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from kzg_ctypes import Kzg, KzgError, HIP_SO  # noqa: E402

LIB = os.environ.get("CKZG_HIP_SO") or HIP_SO
fa = C.CDLL(os.environ["FAILALLOC_SO"])
fa.failalloc_arm.argtypes = [C.c_long, C.c_int]
fa.failalloc_fired.restype = C.c_long
fa.failalloc_seen.restype = C.c_long
fa.failalloc_free_bytes.restype = C.c_longlong
C_KZG_MALLOC = 3


def blobs(n, seed):
    a = np.random.default_rng(seed).integers(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    a[:, :, 0] = 0
    return a.reshape(n, -1).tobytes()


def ret_code(e):
    s = str(e)
    return int(s.rsplit(" ", 1)[1]) if "C_KZG_RET" in s or "->" in s else None


def load(options=None, precompute=0):
    return Kzg(LIB, "", precompute=precompute, options=options or {})


report = {"load": {}}
k0 = load()
blob = blobs(1, 1)
_many = blobs(40, 2)
many = [_many[i * 131072:(i + 1) * 131072] for i in range(40)]
commitment = k0.blob_to_kzg_commitment(blob)
proof = k0.compute_blob_kzg_proof(blob, commitment)
cells, cproofs = k0.compute_cells_and_kzg_proofs(blob)
many_c = [k0.blob_to_kzg_commitment(b) for b in many]
many_p = [k0.compute_blob_kzg_proof(b, c) for b, c in zip(many, many_c)]
half = list(range(0, 128, 2))
half_cells = [cells[i] for i in half]
z = (7).to_bytes(32, "big")
kproof, ky = k0.compute_kzg_proof(blob, z)
k0.close()

OPS = {
    "blob_to_kzg_commitment": lambda k: k.blob_to_kzg_commitment(blob),
    "compute_kzg_proof": lambda k: k.compute_kzg_proof(blob, z),
    "compute_blob_kzg_proof": lambda k: k.compute_blob_kzg_proof(blob, commitment),
    "verify_kzg_proof": lambda k: k.verify_kzg_proof(commitment, z, ky, kproof),
    "verify_blob_kzg_proof": lambda k: k.verify_blob_kzg_proof(blob, commitment, proof),
    "verify_blob_kzg_proof_batch": lambda k: k.verify_blob_kzg_proof_batch(many, many_c, many_p),
    "compute_cells_and_kzg_proofs": lambda k: k.compute_cells_and_kzg_proofs(blob),
    "compute_cells": lambda k: k.compute_cells(blob),
    "recover_cells_and_kzg_proofs": lambda k: k.recover_cells_and_kzg_proofs(half, half_cells),
    "verify_cell_kzg_proof_batch": lambda k: k.verify_cell_kzg_proof_batch([commitment] * 128, list(range(128)),
                                                                           cells, cproofs),
}


# the additive batch entry points (include/ckzg_hip.h), host pointers: staging buffers, pipelines, result drains
def commit_batch(k, n=40):
    out, st = C.create_string_buffer(48 * n), C.create_string_buffer(n)
    k._call("ckzg_hip_blob_to_kzg_commitment_batch", out, st, b"".join(many[:n]), C.c_uint64(n), k.sp)
    return out.raw, st.raw


def cells_batch(k, n=9):   # more blobs than the low-latency proof path takes: the FK20 pipeline
    cl, pr, st = C.create_string_buffer(128 * 2048 * n), C.create_string_buffer(128 * 48 * n), C.create_string_buffer(n)
    k._call("ckzg_hip_compute_cells_and_kzg_proofs_batch", cl, pr, st, b"".join(many[:n]), C.c_uint64(n), k.sp)
    return cl.raw, pr.raw, st.raw


def proof_batch(k, n=5):
    pr, st = C.create_string_buffer(48 * n), C.create_string_buffer(n)
    k._call("ckzg_hip_compute_blob_kzg_proof_batch", pr, st, b"".join(many[:n]), b"".join(many_c[:n]), C.c_uint64(n), k.sp)
    return pr.raw, st.raw


def recover_batch(k, rows=3):
    idx = (C.c_uint64 * 64)(*half)
    cl, pr, st = (C.create_string_buffer(128 * 2048 * rows), C.create_string_buffer(128 * 48 * rows),
                  C.create_string_buffer(rows))
    k._call("ckzg_hip_recover_cells_and_kzg_proofs_batch", cl, pr, st, idx, b"".join(half_cells) * rows,
            C.c_uint64(64), C.c_uint64(rows), k.sp)
    return cl.raw, pr.raw, st.raw


# the resident form of the blob-batch verification (inputs in HBM: GPU challenges, evaluation from the bytes, transcript
# rows assembled on the device, page-locked buffers for what comes back).  The inputs are put on the device ONCE, before
# any failure is armed: the test's own hipMalloc calls go through the same interposer.
_rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
_rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_resident = {}


def _resident_inputs(n):
    if n not in _resident:
        ptrs = []
        for src in (b"".join(many[:n]), b"".join(many_c[:n]), b"".join(many_p[:n])):
            q = C.c_void_p()
            assert _rt.hipMalloc(C.byref(q), len(src)) == 0
            assert _rt.hipMemcpy(q, C.cast(C.c_char_p(src), C.c_void_p), len(src), 1) == 0
            ptrs.append(q)
        _resident[n] = ptrs
    return _resident[n]


def verify_resident(k, n=40):
    d = _resident_inputs(n)
    ok = C.c_bool(False)
    k._call("ckzg_hip_verify_blob_kzg_proof_batch_device", C.byref(ok), d[0], d[1], d[2], C.c_uint64(n), k.sp)
    return bool(ok.value)


_resident_inputs(40)
OPS.update({
    "ckzg_hip_verify_blob_kzg_proof_batch_device": verify_resident,
    "ckzg_hip_blob_to_kzg_commitment_batch": commit_batch,
    "ckzg_hip_compute_cells_and_kzg_proofs_batch": cells_batch,
    "ckzg_hip_compute_blob_kzg_proof_batch": proof_batch,
    "ckzg_hip_recover_cells_and_kzg_proofs_batch": recover_batch,
})
only = os.environ.get("FAILALLOC_ONLY")
if only:
    OPS = {n: f for n, f in OPS.items() if n in only.split(",")}
# FAILALLOC_SECTIONS: which walks this process runs (tests/test_gpu_alloc_failures.py runs each in a process of its own,
# under its own deadline); default: all of them
ALL_SECTIONS = ("ops", "load", "ops_streams_events", "load_streams_events", "fan_out", "coalesced_callers", "widening")
SECTIONS = tuple(x for x in os.environ.get("FAILALLOC_SECTIONS", ",".join(ALL_SECTIONS)).split(",") if x)
assert all(x in ALL_SECTIONS for x in SECTIONS), SECTIONS


def progress(msg):
    """where a stalled run was: the last of these lines on stderr names the case (tests/watchdog.py prints the tail)"""
    sys.stderr.write("[failalloc] %s\n" % msg)
    sys.stderr.flush()

problems = []
# Leak checks compare hipMemGetInfo's free figure with the one taken after a SUCCESSFUL run of the same thing: the
# runtime keeps pools of its own (hundreds of MB after the first kernels, constant afterwards) that are no leak.
# A failure can also send a call down a path whose kernels have not run yet in this process (a verification whose
# call-time table does not fit falls back to ladder sums), and the runtime then grows its pools once more -- 788 MB of
# scratch, kept.  So one step per watch is taken as the runtime's and becomes the new baseline; what a leak does, and
# a pool does not, is come back: any second step is reported (the sticky pass repeats every case of the single one).
LEAK = 4 << 20
max_delta = 0


class LeakWatch:
    """steps: how many one-off growths of the runtime's own pools this walk may see (a leak comes back with every
    failure; a pool grows once per code path that runs for the first time in the process)"""

    def __init__(self, steps=1):
        self.base = fa.failalloc_free_bytes()
        self.firsts = []
        self.steps = steps

    def check(self, what):
        global max_delta
        d = self.base - fa.failalloc_free_bytes()
        if d <= LEAK:
            max_delta = max(max_delta, d)
            return 0
        if len(self.firsts) < self.steps:
            self.firsts.append((what, d))
            self.base -= d
            return 0
        max_delta = max(max_delta, d)
        problems.append("%s: %d bytes of device memory not returned (after earlier steps %s)" % (what, d, self.firsts))
        return d


C_KZG_ERROR = 2
fa.failalloc_class.argtypes = [C.c_int]


def walk_ops(cls, allowed, stickies, key):
    """every allocation of the FIRST call of each entry point on a fresh KZGSettings (that is where the arenas, the
    page-locked staging buffers, the events and the lazily built tables are created)"""
    fa.failalloc_class(cls)
    for name, op in OPS.items():
        k = load()
        want = op(k)
        k.close()
        watch = LeakWatch()
        for sticky in stickies:
            fired_total = 0
            n_alloc = None
            for nth in range(0, 64):
                progress("%s: %s, failure %d, sticky=%d" % (key, name, nth, sticky))
                k = load()
                fa.failalloc_arm(nth, sticky)
                got, code = None, 0
                try:
                    got = op(k)
                except KzgError as e:
                    code = ret_code(e)
                fired = fa.failalloc_fired()
                seen = fa.failalloc_seen()
                fa.failalloc_disarm()
                if not fired:
                    n_alloc = seen
                    if got != want:
                        problems.append("%s: unarmed result differs" % name)
                    k.close()
                    break
                fired_total += 1
                what = "%s, %s %d failed (sticky=%d)" % (name, key, nth, sticky)
                if code not in allowed:
                    problems.append("%s -> C_KZG_RET %s" % (what, code))
                if code == 0 and got != want:
                    problems.append("%s -> OK with WRONG result" % what)
                try:   # the same settings, the same call, right after the failure
                    if op(k) != want:
                        problems.append("%s: wrong result on the call after" % what)
                except KzgError as e:
                    problems.append("%s: call after -> %s" % (what, e))
                k.close()
                watch.check(what)
            report[key].setdefault(name, {})["sticky" if sticky else "single"] = {"seen": n_alloc,
                                                                                  "failures_injected": fired_total}
    fa.failalloc_class(0)


def walk_load(cls, allowed, stickies, key):
    """load_trusted_setup itself: the error code, the struct left freeable, device memory back where it was"""
    fa.failalloc_class(cls)
    k = load()
    assert k.blob_to_kzg_commitment(blob) == commitment
    k.close()
    watch = LeakWatch()
    for sticky in stickies:
        injected, n_alloc, leaked = 0, None, 0
        nth = 0
        while nth < 4000:
            progress("load (%s): failure %d, sticky=%d" % (key, nth, sticky))
            fa.failalloc_arm(nth, sticky)
            code, k = 0, None
            try:
                k = load()
            except KzgError as e:
                code = ret_code(e)
            fired, seen = fa.failalloc_fired(), fa.failalloc_seen()
            fa.failalloc_disarm()
            if k is not None:
                try:
                    if k.blob_to_kzg_commitment(blob) != commitment:
                        problems.append("load: settings loaded around failed %s %d commit wrongly" % (key, nth))
                except KzgError as e:
                    problems.append("load: commitment after a load around failed %s %d -> %s" % (key, nth, e))
                k.close()
            if not fired:
                n_alloc = seen
                break
            injected += 1
            if code not in allowed:
                problems.append("load: %s %d failed (sticky=%d) -> %s" % (key, nth, sticky, code))
            leaked = max(leaked, watch.check("load, failed %s %d (sticky=%d)" % (key, nth, sticky)))
            nth += 1 if nth < 48 else 7
        report["load"].setdefault(key, {})["sticky" if sticky else "single"] = {"seen": n_alloc, "failures_injected": injected,
                                                                               "leaked_bytes": leaked}
    fa.failalloc_class(0)


report["ops"], report["ops_streams_events"] = {}, {}
if "ops" in SECTIONS:
    walk_ops(0, (0, C_KZG_MALLOC), (0, 1), "ops")
if "load" in SECTIONS:
    walk_load(0, (0, C_KZG_MALLOC), (0, 1), "ops")
# streams and events that cannot be created: an internal error (C_KZG_ERROR) or, as the runtime reports it here,
# out of memory; same rules otherwise
if "ops_streams_events" in SECTIONS:
    walk_ops(1, (0, C_KZG_ERROR, C_KZG_MALLOC), (0,), "ops_streams_events")
if "load_streams_events" in SECTIONS:
    walk_load(1, (0, C_KZG_ERROR, C_KZG_MALLOC), (0,), "ops_streams_events")


def section_fan_out():
    """two table sets ("replicas": the one-GPU stand-in for "devices"): the load builds both, a batch fans out over
    both with one worker thread each, and a failure on either side must come back as C_KZG_MALLOC just the same"""
    k = load(options={"replicas": 2})
    want_fan = (commit_batch(k, 40), k.verify_blob_kzg_proof_batch(many, many_c, many_p))
    k.close()
    # (in a process of its own this walk is the first to send a verification down BOTH of its fallback paths -- the
    # call-time table that does not fit, then the ladder sums' scratch: two one-off growths of the runtime's pools,
    # 310 MB and 788 MB; the monolithic round-5 driver had been through them in the walks before)
    watch = LeakWatch(steps=2)
    injected = 0
    for nth in range(0, 96):
        progress("fan-out: failure %d" % nth)
        fa.failalloc_arm(nth, 0)
        code, k = 0, None
        try:
            k = load(options={"replicas": 2})
        except KzgError as e:
            code = ret_code(e)
        load_fired = fa.failalloc_fired()
        if k is not None and not load_fired:   # the load is through: the failure lands in the fanned-out calls
            got = None
            try:
                got = (commit_batch(k, 40), k.verify_blob_kzg_proof_batch(many, many_c, many_p))
            except KzgError as e:
                code = ret_code(e)
            if got is not None and got != want_fan:
                problems.append("fan-out: wrong result with failed allocation %d" % nth)
        fired = fa.failalloc_fired()
        fa.failalloc_disarm()
        if k is not None:
            try:
                if (commit_batch(k, 40), k.verify_blob_kzg_proof_batch(many, many_c, many_p)) != want_fan:
                    problems.append("fan-out: wrong result on the calls after failed allocation %d" % nth)
            except KzgError as e:
                problems.append("fan-out: calls after failed allocation %d -> %s" % (nth, e))
            k.close()
        if not fired:
            break
        injected += 1
        if code not in (0, C_KZG_MALLOC):
            problems.append("fan-out: allocation %d failed -> C_KZG_RET %s" % (nth, code))
        watch.check("fan-out, failed allocation %d" % nth)
    k0.lib.ckzg_hip_set_option(b"replicas", 1)
    report["fan_out"] = {"failures_injected": injected}


def section_coalesced_callers():
    """concurrent one-blob callers (csrc/combiner.hpp): 24 native threads share batch launches while the n-th
    allocation from now on fails (sticky: the page-locked batch buffers, the arenas and staging of the slots the
    launches lease, whatever comes n-th).  Every call must come back with the right commitment or with C_KZG_MALLOC
    -- no hang, no crash, no wrong bytes -- and once allocations work again so must every caller."""
    import importlib.util
    _pkg = os.path.join(os.path.dirname(os.path.dirname(HERE)), "c-kzg-4844_amd")
    _spec = importlib.util.spec_from_file_location("ckzg_fanout", os.path.join(_pkg, "fanout.py"))
    fo = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(fo)
    watch = LeakWatch()
    threads = 24
    ins = [many[t % 40] for t in range(threads)]
    want_cm = [many_c[t % 40] for t in range(threads)]
    coalesce_injected, coalesce_batches = 0, 0
    for nth in list(range(0, 12)) + [16, 24, 40]:
        progress("coalesced callers: allocations fail from the %d-th" % nth)
        k = load()
        fa.failalloc_arm(nth, 1)
        st, rets, outs = fo.run(k, LIB, fo.OP_COMMIT, ins, max_calls=6)
        fired = fa.failalloc_fired()
        fa.failalloc_disarm()
        for t in range(threads):
            if rets[t] == 0 and outs[t] != want_cm[t]:
                problems.append("coalesced callers, allocations failing from the %d-th: thread %d got OK with WRONG bytes" % (nth, t))
            elif rets[t] not in (0, C_KZG_MALLOC):
                problems.append("coalesced callers, allocations failing from the %d-th: thread %d -> C_KZG_RET %d" % (nth, t, rets[t]))
        progress("coalesced callers: allocations work again (failed from the %d-th)" % nth)
        st2, rets2, outs2 = fo.run(k, LIB, fo.OP_COMMIT, ins, max_calls=4)
        if rets2 != [0] * threads or outs2 != want_cm:
            problems.append("coalesced callers: wrong results after allocations work again (failed from the %d-th)" % nth)
        cs = fo.coalesce_stats(k, 0)
        coalesce_batches += cs["batches"] if cs else 0
        if cs and (cs["rescued"] or cs["gave_up"]):
            problems.append("coalesced callers, allocations failing from the %d-th: the queueing protocol needed its net "
                            "(open batches released by a member's periodic look: %d, calls that gave up at the deadline: %d)" %
                            (nth, cs["rescued"], cs["gave_up"]))
        k.close()
        coalesce_injected += 1 if fired else 0
        watch.check("coalesced callers, allocations failing from the %d-th" % nth)
    report["coalesced_callers"] = {"levels_with_failures": coalesce_injected, "batch_launches": coalesce_batches}


def section_widening():
    """background widening under an exhausted device: the tables stay at whatever width was reached, calls go on"""
    watch = LeakWatch()
    progress("widening: load")
    k = load(options={"async_tables": 1, "commit_wbits": 13, "proof_wbits": 11, "fk20_wbits": 10})
    fa.failalloc_arm(0, 1)
    k.lib.ckzg_hip_wait_tables.argtypes = [C.c_void_p]
    progress("widening: wait_tables")
    k.lib.ckzg_hip_wait_tables(k.sp)
    fa.failalloc_disarm()
    progress("widening: calls after")
    try:
        if k.blob_to_kzg_commitment(blob) != commitment:
            problems.append("widening: wrong commitment after the widener ran out of memory")
        if k.compute_cells_and_kzg_proofs(blob) != (cells, cproofs):
            problems.append("widening: wrong cells/proofs after the widener ran out of memory")
    except KzgError as e:
        problems.append("widening: %s" % e)
    for o, v in (("async_tables", 0), ("commit_wbits", 10), ("proof_wbits", 8), ("fk20_wbits", 0)):
        k.lib.ckzg_hip_set_option(o.encode(), v)
    k.close()
    watch.check("after the widening run")
    report["widening"] = {"ran": True}


if "fan_out" in SECTIONS:
    section_fan_out()
if "coalesced_callers" in SECTIONS:
    section_coalesced_callers()
if "widening" in SECTIONS:
    section_widening()
report["sections"] = list(SECTIONS)
report["max_free_delta_bytes"] = max_delta
report["problems"] = problems
progress("done")
print(json.dumps(report))
sys.exit(1 if problems else 0)
