"""GPU tests for the round-3 work: the pipelined (chunked, DMA-in-place) form of verify_blob_kzg_proof_batch for
batches of >= 1024 blobs (src/eip4844/eip4844.c:775-844), the device-resident entry point
ckzg_hip_verify_blob_kzg_proof_batch_device, and the library's promise to leave the caller's HIP device alone."""
import ctypes as C

import pytest

from test_gpu_commitment import R, rand_blob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def material(oracle):
    """Expected values come from the CPU oracle, never from the library under test."""
    blobs = [rand_blob(171, i) for i in range(8)]
    cm = [oracle.blob_to_kzg_commitment(b) for b in blobs]
    pr = [oracle.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    return blobs, cm, pr


@pytest.fixture(scope="module")
def rt():
    lib = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    lib.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    lib.hipHostFree.argtypes = [C.c_void_p]
    lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.hipFree.argtypes = [C.c_void_p]
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.hipGetDevice.argtypes = [C.POINTER(C.c_int)]
    return lib


def _host(hip, bb, cc, pp, n):
    f = hip.lib.verify_blob_kzg_proof_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    ok = C.c_bool(False)
    as_p = lambda x: x if isinstance(x, C.c_void_p) else C.cast(C.c_char_p(x), C.c_void_p)  # noqa: E731
    rc = f(C.byref(ok), as_p(bb), as_p(cc), as_p(pp), n, C.addressof(hip.s))
    return rc, ok.value


def _device(hip, rt, bb, cc, pp, n):
    f = hip.lib.ckzg_hip_verify_blob_kzg_proof_batch_device
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    d = [C.c_void_p() for _ in range(3)]
    try:
        for q, src in zip(d, (bb, cc, pp)):
            assert rt.hipMalloc(C.byref(q), max(len(src), 1)) == 0
            if src:
                assert rt.hipMemcpy(q, C.cast(C.c_char_p(src), C.c_void_p), len(src), 1) == 0
        ok = C.c_bool(False)
        rc = f(C.byref(ok), d[0], d[1], d[2], n, C.addressof(hip.s))
        return rc, ok.value
    finally:
        for q in d:
            rt.hipFree(q)


def _inputs(material, n):
    blobs, cm, pr = material
    order = [(3 * i + i // 11) % 8 for i in range(n)]
    return (b"".join(blobs[k] for k in order), b"".join(cm[k] for k in order), b"".join(pr[k] for k in order), order)


@pytest.mark.parametrize("n", [1024, 1100])
def test_pipelined_verification_pageable_pinned_and_resident(hip, rt, material, n):
    # 1024: exactly four chunks of 256; 1100: a ragged fifth chunk of 76 blobs
    blobs, cm, pr = material
    bb, cc, pp, order = _inputs(material, n)
    assert _host(hip, bb, cc, pp, n) == (0, True)
    pin = C.c_void_p()
    assert rt.hipHostMalloc(C.byref(pin), len(bb), 0) == 0
    try:
        C.memmove(pin, bb, len(bb))
        assert _host(hip, pin, cc, pp, n) == (0, True)
        assert _device(hip, rt, bb, cc, pp, n) == (0, True)
        # one wrong (but valid) proof in the first chunk, in a middle chunk, in the last blob
        for at in (5, 600, n - 1):
            bad = pp[:48 * at] + pr[(order[at] + 1) % 8] + pp[48 * (at + 1):]
            assert _host(hip, bb, cc, bad, n) == (0, False), at
            assert _host(hip, pin, cc, bad, n) == (0, False), at
            assert _device(hip, rt, bb, cc, bad, n) == (0, False), at
        # a commitment that belongs to another blob
        wrong = cc[:48 * 777] + cm[(order[777] + 3) % 8] + cc[48 * 778:]
        assert _host(hip, pin, wrong, pp, n) == (0, False)
        assert _device(hip, rt, bb, wrong, pp, n) == (0, False)
        # a non-canonical field element in the LAST chunk: BADARGS from every form (blob.c:31-38)
        nb = bytearray(bb)
        at = (n - 2) * 131072 + 32 * 4000
        nb[at:at + 32] = R.to_bytes(32, "big")
        assert _host(hip, bytes(nb), cc, pp, n)[0] == 1
        C.memmove(pin, bytes(nb), len(nb))
        assert _host(hip, pin, cc, pp, n)[0] == 1
        assert _device(hip, rt, bytes(nb), cc, pp, n)[0] == 1
        # an invalid point encoding (x not on the curve / flag bits) anywhere: BADARGS (bytes.c:81-95)
        C.memmove(pin, bb, len(bb))
        badc = cc[:48 * 300] + b"\x8f" + cc[48 * 300 + 1:]
        rc_h = _host(hip, pin, badc, pp, n)[0]
        assert rc_h == 1 and _device(hip, rt, bb, badc, pp, n)[0] == 1
    finally:
        rt.hipHostFree(pin)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 64])
def test_resident_verification_small_batches(hip, rt, material, n):
    blobs, cm, pr = material
    bb, cc, pp, order = _inputs(material, n)
    assert _device(hip, rt, bb, cc, pp, n) == (0, True)
    if n:
        bad = pp[:48 * (n - 1)] + pr[(order[n - 1] + 1) % 8]
        assert _device(hip, rt, bb, cc, bad, n) == (0, False)
        assert _host(hip, bb, cc, bad, n) == (0, False)


def test_pipe_threshold_can_be_lowered_for_every_chunk_shape(material):
    """ckzg_hip_set_option("verify_pipe_min", 2): a child process (the option is process-wide) runs 300 / 513 blobs
    through the pipelined form (two chunks, the second of 44 / a third chunk of one blob)."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import ctypes as C, json, sys
sys.path.insert(0, "tests")
from kzg_ctypes import HIP_SO, Kzg
from test_gpu_commitment import rand_blob
hip = Kzg(HIP_SO, "", precompute=0, options={"commit_wbits": 8, "proof_wbits": 0, "verify_pipe_min": 2})
blobs = [rand_blob(171, i) for i in range(4)]
cm = [hip.blob_to_kzg_commitment(b) for b in blobs]
pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
out = {}
for n in (300, 513):
    o = [(5 * i) % 4 for i in range(n)]
    good = hip.verify_blob_kzg_proof_batch([blobs[k] for k in o], [cm[k] for k in o], [pr[k] for k in o])
    p2 = [pr[k] for k in o]; p2[n - 1] = pr[(o[n - 1] + 1) % 4]
    bad = hip.verify_blob_kzg_proof_batch([blobs[k] for k in o], [cm[k] for k in o], p2)
    out[str(n)] = [good, bad]
print(json.dumps(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    from watchdog import run_watched
    r = run_watched([sys.executable, "-c", code], cwd=root, env=env, timeout=280, name="round3_pipelined_verify")
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res == {"300": [True, False], "513": [True, False]}


def test_calls_leave_the_callers_hip_device_selected(hip, rt):
    dev = C.c_int(-1)
    assert rt.hipGetDevice(C.byref(dev)) == 0
    before = dev.value
    hip.blob_to_kzg_commitment(rand_blob(172, 0))
    assert rt.hipGetDevice(C.byref(dev)) == 0 and dev.value == before


# ---------------------------------------------------------------------------------------------
# progressive widening ("async_tables"): the load returns with default-width tables, wider ones are built in the
# background and published one at a time; every call sees one consistent set
# ---------------------------------------------------------------------------------------------

def test_async_tables_serve_calls_while_they_widen(oracle):
    import time
    from kzg_ctypes import HIP_SO, Kzg
    from test_gpu_round2 import _restore
    blob = rand_blob(173, 0)
    exp_c = oracle.blob_to_kzg_commitment(blob)
    exp_cp = oracle.compute_cells_and_kzg_proofs(blob)
    t0 = time.perf_counter()
    api = Kzg(HIP_SO, "", precompute=0, options={"async_tables": 1, "commit_wbits": 14, "proof_wbits": 13, "fk20_wbits": 12})
    t_load = time.perf_counter() - t0
    _restore(api)
    api.lib.ckzg_hip_set_option(b"async_tables", 0)
    try:
        wb = api.lib.ckzg_hip_table_wbits
        wb.restype = C.c_int
        ready = api.lib.ckzg_hip_tables_ready
        ready.restype = C.c_int
        seen = set()
        rounds = 0
        # four more callers hammer the same KZGSettings while the tables change underneath them
        import threading
        errs, stop = [], threading.Event()

        def hammer():
            while not stop.is_set():
                if api.blob_to_kzg_commitment(blob) != exp_c:
                    errs.append("commitment")
                    return

        th = [threading.Thread(target=hammer) for _ in range(4)]
        for t in th:
            t.start()
        try:
            while True:
                done = bool(ready(api.sp))
                seen.add(tuple(int(wb(api.sp, k)) for k in range(3)))
                assert api.blob_to_kzg_commitment(blob) == exp_c
                got = api.compute_cells_and_kzg_proofs(blob)
                assert got[0] == exp_cp[0] and got[1] == exp_cp[1]
                rounds += 1
                if done:
                    break
        finally:
            stop.set()
            for t in th:
                t.join()
        assert not errs
        assert api.lib.ckzg_hip_wait_tables(api.sp) == 0
        final = tuple(int(wb(api.sp, k)) for k in range(3))
        assert final == (14, 12, 13), (final, sorted(seen))
        assert (10, 8, 8) in seen or rounds == 1, sorted(seen)   # the first calls really ran on the narrow tables
        # a batch on the widened FK20 table, and the widened commitment table at batch size
        got = api.compute_cells_and_kzg_proofs(blob)
        assert got[1] == exp_cp[1]
        assert t_load < 60.0   # (no tight bound: a load that follows a process or test which has just released a
        # few hundred GB waits for the driver's VRAM scrub)
    finally:
        api.close()


def test_async_tables_cancelled_by_free(rt):
    """free_trusted_setup while the wide tables are still being built: the build stops, everything is released."""
    from kzg_ctypes import HIP_SO, Kzg
    from test_gpu_round2 import _restore
    rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    free0, total = C.c_size_t(), C.c_size_t()
    assert rt.hipMemGetInfo(C.byref(free0), C.byref(total)) == 0
    api = Kzg(HIP_SO, "", precompute=0, options={"async_tables": 1, "commit_wbits": 16, "proof_wbits": 0, "fk20_wbits": 8})
    _restore(api)
    api.lib.ckzg_hip_set_option(b"async_tables", 0)
    assert api.blob_to_kzg_commitment(bytes(131072)) == b"\xc0" + bytes(47)
    api.close()   # the 103 GB build is somewhere in the middle
    free1 = C.c_size_t()
    assert rt.hipMemGetInfo(C.byref(free1), C.byref(total)) == 0
    assert free1.value >= free0.value - (1 << 30), (free0.value, free1.value)


def test_setup_points_at_infinity_give_infinity_table_rows(tmp_path):
    """A setup file may hold the point at infinity (the reference checks curve membership only, setup.c:447-477): the
    table rows of such a base are all infinity and every sum simply skips them.  Oracle and product load the same
    doctored file (Lagrange point 5 and monomial point 100 replaced) and must agree on commitments, cells and proofs."""
    from conftest import ORACLE_SO
    from kzg_ctypes import HIP_SO, TRUSTED_SETUP, Kzg
    from oracle_binding import OracleKzg
    from test_gpu_round2 import _restore
    lines = open(TRUSTED_SETUP).read().split("\n")
    inf = "c0" + "00" * 47
    assert lines[0].strip() == "4096" and lines[1].strip() == "65" and len(lines[2]) == 96
    lines[2 + 5] = inf
    lines[2 + 4096 + 65 + 100] = inf
    path = tmp_path / "setup_with_infinity.txt"
    path.write_text("\n".join(lines))
    blob = rand_blob(174, 0)
    orc = OracleKzg(ORACLE_SO, precompute=0, setup_path=str(path))
    try:
        exp_c = orc.blob_to_kzg_commitment(blob)
        exp_cp = orc.compute_cells_and_kzg_proofs(blob)
    finally:
        orc.close()
    for opts in ({"commit_wbits": 9, "proof_wbits": 6, "fk20_wbits": 8}, {"commit_wbits": 12, "proof_wbits": 0, "direct_max": 0, "fk20_wbits": 10}):
        api = Kzg(HIP_SO, "", precompute=0, setup_path=str(path), options=opts)
        _restore(api)
        try:
            assert api.blob_to_kzg_commitment(blob) == exp_c
            got = api.compute_cells_and_kzg_proofs(blob)
            assert got[0] == exp_cp[0] and got[1] == exp_cp[1]
        finally:
            api.close()


# ---------------------------------------------------------------------------------------------
# the round-3 paths behind the C-ABI multi-device fan-out (two table replicas on the one GPU stand in for two devices)
# ---------------------------------------------------------------------------------------------

def test_fan_out_of_pipelined_verification_and_async_tables_over_two_replicas(material):
    from kzg_ctypes import HIP_SO, Kzg
    from test_gpu_round2 import _restore
    blobs, cm, pr = material
    api = Kzg(HIP_SO, "", precompute=0, options={"replicas": 2, "async_tables": 1, "commit_wbits": 12, "proof_wbits": 0, "fk20_wbits": 9})
    _restore(api)
    try:
        nd = api.lib.ckzg_hip_num_devices
        nd.restype = C.c_int
        assert nd(api.sp) == 2
        n = 2304   # two shards of 1152 blobs: each takes the pipelined form
        bb, cc, pp, order = _inputs(material, n)
        assert _host(api, bb, cc, pp, n) == (0, True)
        for at in (7, 1151, 1152, n - 1):   # either shard, and both sides of the boundary
            bad = pp[:48 * at] + pr[(order[at] + 1) % 8] + pp[48 * (at + 1):]
            assert _host(api, bb, cc, bad, n) == (0, False), at
        assert api.lib.ckzg_hip_wait_tables(api.sp) == 0
        wb = api.lib.ckzg_hip_table_wbits
        wb.restype = C.c_int
        assert (int(wb(api.sp, 0)), int(wb(api.sp, 1))) == (12, 9)
        # both replicas were widened: commitments from either pool (single calls go round robin) match the oracle's
        for i in range(4):
            assert api.blob_to_kzg_commitment(blobs[i]) == cm[i]
        assert _host(api, bb, cc, pp, n) == (0, True)
    finally:
        api.close()


def test_call_time_table_sums_with_points_at_infinity(hip, rt, material):
    """Host-pointer batches of >= 2560 blobs take their three random-linear-combination sums from a fixed-base table
    built over the batch's own commitments and proofs while the blobs are copied (msm.hip: call-time tables).  The
    all-zero blob has the point at infinity as commitment AND proof: its table rows are all infinity."""
    blobs, cm, pr = material
    n = 2600
    inf48 = b"\xc0" + bytes(47)
    zero = bytes(131072)
    assert hip.blob_to_kzg_commitment(zero) == inf48 and hip.compute_blob_kzg_proof(zero, inf48) == inf48
    order = [(3 * i + i // 11) % 8 for i in range(n)]
    zeros = set(range(5, n, 9)) | {0, n - 1}
    bb = b"".join(zero if i in zeros else blobs[order[i]] for i in range(n))
    cc = b"".join(inf48 if i in zeros else cm[order[i]] for i in range(n))
    pp = b"".join(inf48 if i in zeros else pr[order[i]] for i in range(n))
    assert _host(hip, bb, cc, pp, n) == (0, True)
    for at in (1, 1300, n - 2):          # a wrong (valid) proof for a non-zero blob
        assert at not in zeros
        bad = pp[:48 * at] + pr[(order[at] + 1) % 8] + pp[48 * (at + 1):]
        assert _host(hip, bb, cc, bad, n) == (0, False), at
    # a non-trivial proof claimed for the zero blob, and infinity claimed for a real one
    bad = pp[:48 * 5] + pr[0] + pp[48 * 6:]
    assert _host(hip, bb, cc, bad, n) == (0, False)
    bad = pp[:48 * 6] + inf48 + pp[48 * 7:]
    assert 6 not in zeros and _host(hip, bb, cc, bad, n) == (0, False)
    # the same verdicts from the resident form (per-term ladders)
    assert _device(hip, rt, bb, cc, pp, n) == (0, True)
    assert _device(hip, rt, bb, cc, bad, n) == (0, False)


def test_call_time_table_sums_in_cell_batch_verification(hip, oracle, material):
    """verify_cell_kzg_proof_batch over >= 6144 cells takes its four sums from a table built over the batch's proofs,
    its distinct commitments and the 64 monomial setup points while the host hashes the transcript (eip7594.c:825-974).
    Cells of the all-zero blob carry the point at infinity as commitment and as proof; unsorted, duplicated cells."""
    blobs, cm, pr = material
    cp = [oracle.compute_cells_and_kzg_proofs(b) for b in blobs[:3]]
    inf48 = b"\xc0" + bytes(47)
    zero_cells, zero_proofs = hip.compute_cells_and_kzg_proofs(bytes(131072))
    assert all(p == inf48 for p in zero_proofs)
    n = 6200
    ent = [((7 * i + i // 13) % 4, (31 * i + 5) % 128) for i in range(n)]   # blob 3 = the zero blob; duplicates occur
    com = [inf48 if b == 3 else cm[b] for b, _ in ent]
    cells = [zero_cells[c] if b == 3 else cp[b][0][c] for b, c in ent]
    proofs = [inf48 if b == 3 else cp[b][1][c] for b, c in ent]
    idx = [c for _, c in ent]
    assert hip.verify_cell_kzg_proof_batch(com, idx, cells, proofs) is True
    assert hip.verify_cell_kzg_proof_batch(com[:6143], idx[:6143], cells[:6143], proofs[:6143]) is True   # ladders: same verdict
    for at in (0, 2222, n - 1):
        b, c = ent[at]
        p2 = list(proofs)
        p2[at] = cp[(b + 1) % 3][1][c]       # a valid point, the wrong proof
        assert hip.verify_cell_kzg_proof_batch(com, idx, cells, p2) is False, at
    c2 = list(com)
    assert ent[17][0] != 3
    c2[17] = cm[(ent[17][0] + 1) % 3]
    assert hip.verify_cell_kzg_proof_batch(c2, idx, cells, proofs) is False
    i2 = list(idx)
    at = next(i for i in range(100, n) if ent[i][0] != 3)   # (for the zero blob every index is a true statement)
    i2[at] = (i2[at] + 1) % 128
    assert hip.verify_cell_kzg_proof_batch(com, i2, cells, proofs) is False
    zi = next(i for i in range(100, n) if ent[i][0] == 3)
    i3 = list(idx)
    i3[zi] = (i3[zi] + 1) % 128
    assert hip.verify_cell_kzg_proof_batch(com, i3, cells, proofs) is True
    # a small cross-check of the whole construction against the oracle on a prefix that it can afford
    m = 300
    assert oracle.verify_cell_kzg_proof_batch(com[:m], idx[:m], cells[:m], proofs[:m]) is True


# ---------------------------------------------------------------------------------------------
# the ladder sums stay a supported path (option "verify_call_table" = 0, and the fallback of a device too full for
# the call-time table): the same batches, table off, in a child process (the option is process-wide)
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_ladder_sums_without_the_call_time_table(oracle):
    import json
    import os
    import subprocess
    import sys
    code = r'''
import ctypes as C, json, sys
sys.path.insert(0, "tests")
from kzg_ctypes import HIP_SO, Kzg
from test_gpu_commitment import rand_blob
hip = Kzg(HIP_SO, "", precompute=0, options={"commit_wbits": 8, "verify_call_table": 0})
blobs = [rand_blob(173, i) for i in range(3)] + [bytes(131072)]     # the zero blob: commitment and proofs at infinity
cm = [hip.blob_to_kzg_commitment(b) for b in blobs]
pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
cp = [hip.compute_cells_and_kzg_proofs(b) for b in blobs]
out = {"commitments": [c.hex() for c in cm]}
for n in (9, 300, 1100):
    o = [(5 * i) % 4 for i in range(n)]
    good = hip.verify_blob_kzg_proof_batch([blobs[k] for k in o], [cm[k] for k in o], [pr[k] for k in o])
    p2 = [pr[k] for k in o]; p2[n - 1] = pr[(o[n - 1] + 1) % 4]
    bad = hip.verify_blob_kzg_proof_batch([blobs[k] for k in o], [cm[k] for k in o], p2)
    out["blobs_%d" % n] = [good, bad]
for n in (130, 700):
    ent = [((7 * i) % 4, (11 * i) % 128) for i in range(n)]
    args = ([cm[b] for b, _ in ent], [c for _, c in ent], [cp[b][0][c] for b, c in ent])
    good = hip.verify_cell_kzg_proof_batch(*args, [cp[b][1][c] for b, c in ent])
    prf = [cp[b][1][c] for b, c in ent]; prf[3] = cp[(ent[3][0] + 1) % 3][1][ent[3][1]]
    bad = hip.verify_cell_kzg_proof_batch(*args, prf)
    out["cells_%d" % n] = [good, bad]
print(json.dumps(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    from watchdog import run_watched
    r = run_watched([sys.executable, "-c", code], cwd=root, env=env, timeout=280, name="round3_forms")
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    from test_gpu_commitment import rand_blob
    want_cm = [oracle.blob_to_kzg_commitment(rand_blob(173, i)).hex() for i in range(3)] + ["c0" + "00" * 47]
    assert res.pop("commitments") == want_cm
    assert res == {"blobs_9": [True, False], "blobs_300": [True, False], "blobs_1100": [True, False],
                   "cells_130": [True, False], "cells_700": [True, False]}


@pytest.mark.parametrize("n", [255, 256, 257, 259, 1023, 1025, 1031])
def test_evaluation_kernel_forms_at_their_switch_points(hip, rt, material, n):
    """k_eval_tree takes three launch forms by batch size (verify.hip: four waves per blob below 256 blobs, a wave per
    blob with four blobs per workgroup up to 1024, eight per workgroup above): both sides of each hand-over, with batch
    sizes that leave a workgroup's last turn ragged (257 = 64 x 4 + 1, 1025 = 128 x 8 + 1, 1031 = 128 x 8 + 7).  The
    expected verdicts come from the oracle's proofs; a wrong proof and a non-canonical field element are placed in the
    LAST blob (the ragged wave) and the element walks over a thread's tile positions (evaluate_polynomial_in_evaluation_form,
    src/eip4844/eip4844.c:192-240; bytes_to_bls_field, src/common/bytes.c:52-70)."""
    blobs, cm, pr = material
    bb, cc, pp, order = _inputs(material, n)
    assert _device(hip, rt, bb, cc, pp, n) == (0, True)
    assert _host(hip, bb, cc, pp, n) == (0, True)
    bad = pp[:48 * (n - 1)] + pr[(order[n - 1] + 1) % 8]
    assert _device(hip, rt, bb, cc, bad, n) == (0, False)
    assert _host(hip, bb, cc, bad, n) == (0, False)
    # element index inside the last blob: first / last leaf of a thread, of a tile of four, of the blob
    for k, elem in enumerate((0, 3, 4, 63, 64, 2049, 4095)):
        nb = bytearray(bb)
        at = (n - 1) * 131072 + 32 * elem
        nb[at:at + 32] = (R + k).to_bytes(32, "big")
        assert _device(hip, rt, bytes(nb), cc, pp, n)[0] == 1, elem
        if k % 3 == 0:
            assert _host(hip, bytes(nb), cc, pp, n)[0] == 1, elem
    # the largest canonical element is fine (and changes the blob: the proof no longer fits)
    nb = bytearray(bb)
    at = (n - 1) * 131072 + 32 * 777
    nb[at:at + 32] = (R - 1).to_bytes(32, "big")
    assert _device(hip, rt, bytes(nb), cc, pp, n) == (0, False)
    # host pointers with the challenges hashed on the GPU (what a rank with few host threads takes): from 640 blobs the
    # hash runs on its own compute units as in the resident form, with and without the partition
    hip.lib.ckzg_hip_set_option(b"gpu_sha_min", 1)
    try:
        for part in (1, 0):
            assert hip.lib.ckzg_hip_set_option(b"verify_cu_partition", part) == 0
            assert _host(hip, bb, cc, pp, n) == (0, True), part
            assert _host(hip, bb, cc, bad, n) == (0, False), part
            assert _device(hip, rt, bb, cc, pp, n) == (0, True), part
    finally:
        hip.lib.ckzg_hip_set_option(b"gpu_sha_min", 0)
        hip.lib.ckzg_hip_set_option(b"verify_cu_partition", 1)


def test_resident_verification_when_an_earlier_batch_sized_the_arena(hip, rt, material):
    """A slot's arena keeps its size between calls (one and a half times what the call that grew it needed): a batch
    one and a half times as large as the one before fits the old block only if every buffer of the call is in the
    request.  Round 6 forgot the transcript rows (160 bytes per blob) of the resident form there: 512 blobs and then 768
    came back C_KZG_MALLOC (found by tools/ubench/sha_contention_probe.py).  Ratios around 1.5 in both forms, growing
    and shrinking, with and without the call-time table."""
    for table in (1, 0):
        assert hip.lib.ckzg_hip_set_option(b"verify_call_table", table) == 0
        try:
            for n in (512, 768, 769, 1152, 1153, 767, 512, 8, 12, 13):
                bb, cc, pp, _ = _inputs(material, n)
                assert _device(hip, rt, bb, cc, pp, n) == (0, True), (table, n)
                if n <= 768:
                    assert _host(hip, bb, cc, pp, n) == (0, True), (table, n)
        finally:
            hip.lib.ckzg_hip_set_option(b"verify_call_table", 1)


def test_concurrent_resident_verifications_share_the_partition_rule(hip, rt, material):
    """Four threads verify resident batches of >= 640 blobs at once: the first to arrive hashes on the partitioned
    compute units, the others (a GPU hash is already in flight on the device) on the plain streams -- every verdict
    right, valid and invalid batches mixed (ckzg_api2.hip: GpuHashCall)."""
    import threading
    blobs, cm, pr = material
    n = 700
    bb, cc, pp, order = _inputs(material, n)
    bad = pp[:48 * 333] + pr[(order[333] + 1) % 8] + pp[48 * 334:]
    results = {}

    def worker(k):
        out = []
        for rep in range(3):
            want_bad = (k + rep) % 2 == 1
            out.append((_device(hip, rt, bb, cc, bad if want_bad else pp, n), want_bad))
        results[k] = out

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    assert all(not t.is_alive() for t in ts)
    for k in range(4):
        for (rc, ok), want_bad in results[k]:
            assert rc == 0 and ok == (not want_bad), (k, rc, ok, want_bad)
