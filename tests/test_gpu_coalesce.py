"""GPU tests of the combiner (csrc/combiner.hpp): concurrent one-unit callers of the UNCHANGED ckzg.h entry points
-- the reference's only parallel shape, bindings/go/main_test.go:953-971 -- are served by shared batch launches.
What must hold: every caller gets exactly the bytes and the return code the one-blob call gives (expected values
from the CPU oracle), a unit the batch path rejects fails its own caller only, and launches really are shared
(ckzg_hip_coalesce_stats)."""
import ctypes as C
import struct
import threading

import pytest

from conftest import HIP_SO
from kzg_ctypes import Kzg, KzgError
from test_gpu_commitment import rand_blob

pytestmark = pytest.mark.gpu


def _fanout():
    import sys
    return sys.modules["ckzg_4844_amd"].fanout


def _spoil(blob, element=2111):
    b = bytearray(blob)
    b[32 * element:32 * element + 32] = b"\xff" * 32   # >= r: non-canonical (tests/*/invalid_blob_1 of the reference)
    return bytes(b)


@pytest.fixture(scope="module")
def material(oracle):
    blobs = [rand_blob(4004, i) for i in range(16)]
    cm = [oracle.blob_to_kzg_commitment(b) for b in blobs]
    return blobs, cm


@pytest.fixture(scope="module")
def cells_material(oracle, material):
    blobs, _ = material
    return [oracle.compute_cells_and_kzg_proofs(b) for b in blobs[:6]]


def test_64_native_threads_of_mixed_commitments(hip, material):
    """64 threads, every fourth one holding a blob with a non-canonical element: return codes and bytes per caller."""
    fo = _fanout()
    blobs, cm = material
    ins, exp = [], []
    for t in range(64):
        if t % 4 == 3:
            ins.append(_spoil(blobs[t % 16], 17 * t))
            exp.append(None)
        else:
            ins.append(blobs[t % 16])
            exp.append(cm[t % 16])
    before = fo.coalesce_stats(hip, 0)
    st, rets, outs = fo.run(hip, HIP_SO, fo.OP_COMMIT, ins, max_calls=40)
    after = fo.coalesce_stats(hip, 0)
    assert st["calls"] == 64 * 40
    for t in range(64):
        if exp[t] is None:
            assert rets[t] == 1, "thread %d: a bad blob must give C_KZG_BADARGS" % t
        else:
            assert rets[t] == 0 and outs[t] == exp[t], "thread %d" % t
    assert st["not_ok"] == 16 * 40      # the bad blobs' callers, nobody else
    assert after["calls"] - before["calls"] == 64 * 40
    assert after["batches"] > before["batches"] and after["largest"] >= 8, after   # launches were shared


def test_python_threads_of_mixed_calls(hip, oracle, material, cells_material):
    """Commitments, cells+proofs, cells only, blob proofs from Python threads at once (ctypes drops the GIL)."""
    blobs, cm = material
    proofs = [oracle.compute_blob_kzg_proof(blobs[i], cm[i]) for i in range(4)]
    errors = []

    def commit(t):
        for k in range(6):
            i = (t + k) % 16
            if hip.blob_to_kzg_commitment(blobs[i]) != cm[i]:
                errors.append(("commit", t, i))
            try:
                hip.blob_to_kzg_commitment(_spoil(blobs[i], t))
                errors.append(("commit accepted a bad blob", t, i))
            except KzgError:
                pass

    def cells(t):
        for k in range(3):
            i = (t + k) % 6
            c, p = hip.compute_cells_and_kzg_proofs(blobs[i])
            if (c, p) != cells_material[i]:
                errors.append(("cells+proofs", t, i))
            if t % 2 and hip.compute_cells(blobs[i]) != cells_material[i][0]:
                errors.append(("cells", t, i))
            try:
                hip.compute_cells_and_kzg_proofs(_spoil(blobs[i], 4095))
                errors.append(("cells accepted a bad blob", t, i))
            except KzgError:
                pass

    def blob_proof(t):
        for k in range(4):
            i = (t + k) % 4
            if hip.compute_blob_kzg_proof(blobs[i], cm[i]) != proofs[i]:
                errors.append(("blob proof", t, i))
            try:
                hip.compute_blob_kzg_proof(blobs[i], b"\x8f" + cm[i][1:])   # not a point of the curve / subgroup
                errors.append(("blob proof accepted a bad commitment", t, i))
            except KzgError:
                pass

    def guard(f):   # an exception in a thread must fail the test, not vanish
        def run(t):
            try:
                f(t)
            except Exception as e:  # noqa: BLE001
                errors.append((f.__name__, t, repr(e)))
        return run

    th = [threading.Thread(target=guard(f), args=(t,)) for t in range(12) for f in (commit, cells, blob_proof)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]


def test_cells_and_proofs_shared_launches_match_oracle(hip, material, cells_material):
    fo = _fanout()
    blobs, _ = material
    ins = [blobs[t % 6] for t in range(24)]
    ins[5] = _spoil(ins[5])
    before = fo.coalesce_stats(hip, 3)
    st, rets, outs = fo.run(hip, HIP_SO, fo.OP_CELLS_PROOFS, ins, max_calls=4)
    after = fo.coalesce_stats(hip, 3)
    for t in range(24):
        if t == 5:
            assert rets[t] == 1
            continue
        c, p = cells_material[t % 6]
        assert rets[t] == 0 and outs[t] == b"".join(c) + b"".join(p), "thread %d" % t
    assert after["batches"] > before["batches"] and after["largest"] >= 4, after
    # proofs only / cells only are operations of their own
    st, rets, outs = fo.run(hip, HIP_SO, fo.OP_PROOFS, ins[:5] * 3, max_calls=2)
    for t in range(15):
        assert rets[t] == 0 and outs[t][fo.CELLS_BYTES:] == b"".join(cells_material[t % 5][1])
    st, rets, outs = fo.run(hip, HIP_SO, fo.OP_CELLS, ins[:5] * 3, max_calls=2)
    for t in range(15):
        assert rets[t] == 0 and outs[t][:fo.CELLS_BYTES] == b"".join(cells_material[t % 5][0])


def test_recover_callers_share_a_launch_per_index_set(hip, cells_material):
    """Two groups of callers with different column sets run at once: a launch never mixes index sets."""
    fo = _fanout()
    sets = [list(range(0, 128, 2)), list(range(64)) + [100, 127]]
    results = {}

    def group(g):
        idx = sets[g]
        ins = [b"".join(cells_material[t % 4][0][i] for i in idx) for t in range(10)]
        if g == 1:
            ins[3] = ins[3][:40] + b"\xff" * 32 + ins[3][72:]     # a non-canonical element in one caller's cells
        results[g] = fo.run(hip, HIP_SO, fo.OP_RECOVER, ins, max_calls=3, aux=struct.pack("<%dQ" % len(idx), *idx),
                            aux_n=len(idx))

    # (two libckzg_callers runs side by side: each has its own threads and buffers)
    th = [threading.Thread(target=group, args=(g,)) for g in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for g in (0, 1):
        st, rets, outs = results[g]
        for t in range(10):
            if g == 1 and t == 3:
                assert rets[t] == 1
                continue
            c, p = cells_material[t % 4]
            assert rets[t] == 0 and outs[t] == b"".join(c) + b"".join(p), (g, t)
    assert fo.coalesce_stats(hip, 5)["batches"] > 0


def test_blob_proof_callers_with_one_bad_commitment(hip, oracle, material):
    fo = _fanout()
    blobs, cm = material
    exp = [oracle.compute_blob_kzg_proof(blobs[i], cm[i]) for i in range(4)]
    ins = [blobs[t % 4] for t in range(20)]
    aux = [cm[t % 4] for t in range(20)]
    aux[7] = b"\x8f" + aux[7][1:]
    st, rets, outs = fo.run(hip, HIP_SO, fo.OP_BLOB_PROOF, ins, max_calls=5, aux=aux)
    for t in range(20):
        if t == 7:
            assert rets[t] == 1
        else:
            assert rets[t] == 0 and outs[t] == exp[t % 4], t
    assert fo.coalesce_stats(hip, 4)["batches"] > 0


def test_coalescing_can_be_switched_off(oracle, material):
    blobs, cm = material
    k = Kzg(HIP_SO, "", precompute=0, options={"coalesce": 0, "commit_wbits": 8})
    k.lib.ckzg_hip_set_option(b"coalesce", 1)
    k.lib.ckzg_hip_set_option(b"commit_wbits", 10)
    try:
        fo = _fanout()
        assert fo.coalesce_stats(k, 0) is None
        st, rets, outs = fo.run(k, HIP_SO, fo.OP_COMMIT, [blobs[t % 16] for t in range(16)], max_calls=5)
        assert rets == [0] * 16 and outs == [cm[t % 16] for t in range(16)]
    finally:
        k.close()


def test_a_lone_caller_never_queues(hip, material):
    fo = _fanout()
    blobs, cm = material
    before = fo.coalesce_stats(hip, 0)
    for i in range(5):
        assert hip.blob_to_kzg_commitment(blobs[i]) == cm[i]
    after = fo.coalesce_stats(hip, 0)
    assert after["solo"] - before["solo"] == 5 and after["batches"] == before["batches"]


def test_single_blob_verifications_share_a_batch_and_bad_ones_answer_for_themselves(hip, oracle, material):
    """verify_blob_kzg_proof from 48 native threads: valid triples, a wrong proof (valid point, other blob's), a commitment
    that is no curve point, a non-canonical blob.  Every caller gets exactly what the single call gives: true / false /
    C_KZG_BADARGS -- whatever batch it happened to share."""
    fo = _fanout()
    blobs, cm = material
    pr = [oracle.compute_blob_kzg_proof(blobs[i], cm[i]) for i in range(8)]
    ins, aux, exp = [], [], []
    for t in range(48):
        i = t % 8
        if t in (5, 29):
            ins.append(blobs[i]); aux.append(cm[i] + pr[(i + 1) % 8]); exp.append((0, 0))          # wrong proof -> false
        elif t == 11:
            ins.append(blobs[i]); aux.append(b"\x8f" + cm[i][1:] + pr[i]); exp.append((1, 0))       # malformed commitment -> BADARGS
        elif t == 40:
            ins.append(_spoil(blobs[i], 99)); aux.append(cm[i] + pr[i]); exp.append((1, 0))         # non-canonical element -> BADARGS
        else:
            ins.append(blobs[i]); aux.append(cm[i] + pr[i]); exp.append((0, 1))
    before = fo.coalesce_stats(hip, 6)
    st, rets, outs = fo.run(hip, HIP_SO, fo.OP_VERIFY_BLOB, ins, max_calls=6, aux=aux)
    after = fo.coalesce_stats(hip, 6)
    for t in range(48):
        assert (rets[t], outs[t][0]) == exp[t], (t, rets[t], outs[t][0], exp[t])
    assert after["batches"] > before["batches"] and after["retried"] > before["retried"], after
    # all-valid callers: batches come out true, nobody is sent back
    good = [t for t in range(48) if exp[t] == (0, 1)]
    before = fo.coalesce_stats(hip, 6)
    st, rets, outs = fo.run(hip, HIP_SO, fo.OP_VERIFY_BLOB, [ins[t] for t in good], max_calls=6, aux=[aux[t] for t in good])
    after = fo.coalesce_stats(hip, 6)
    assert rets == [0] * len(good) and all(o[0] == 1 for o in outs)
    assert after["retried"] == before["retried"] and after["largest"] >= 4, after


def test_first_calls_of_concurrent_callers_on_fresh_settings(material):
    """The one-blob commitment is a captured hipGraph per stream slot; three callers that arrive together on a fresh
    KZGSettings each capture theirs.  Every later call on every slot must still work (two concurrent captures once left
    a stream in an invalidated capture: found by the ThreadSanitizer pass, whose timing made it likely)."""
    fo = _fanout()
    blobs, cm = material
    for rnd in range(6):
        k = Kzg(HIP_SO, "", precompute=0, options={"commit_wbits": 8, "proof_wbits": 4, "fk20_wbits": 4})
        try:
            # first calls of everything at once: three callers capture their graphs while five others make the first
            # allocations of their slots (arenas, page-locked staging) and copy synchronously
            first_errors = []

            def first(t):
                try:
                    if t < 3:
                        for _ in range(3):
                            if k.blob_to_kzg_commitment(blobs[t]) != cm[t]:
                                first_errors.append(("commit", t))
                    elif t < 6:
                        k.compute_blob_kzg_proof(blobs[t], cm[t])
                    else:
                        k.compute_cells(blobs[t])
                except Exception as e:  # noqa: BLE001
                    first_errors.append((t, repr(e)))

            th = [threading.Thread(target=first, args=(t,)) for t in range(8)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            assert not first_errors, (rnd, first_errors[:3])
            for nt in (3, 3, 8):
                st, rets, outs = fo.run(k, HIP_SO, fo.OP_COMMIT, [blobs[t % 16] for t in range(nt)], max_calls=3)
                assert rets == [0] * nt and outs == [cm[t % 16] for t in range(nt)], (rnd, nt, rets)
            # every slot serves other work afterwards
            pr = k.compute_blob_kzg_proof(blobs[0], cm[0])
            bad = []

            def verify():
                try:
                    for _ in range(3):
                        if not k.verify_blob_kzg_proof(blobs[0], cm[0], pr):
                            bad.append("false")
                except KzgError as e:
                    bad.append(str(e))

            th = [threading.Thread(target=verify) for _ in range(8)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            assert not bad, (rnd, bad[:3])
            assert k.compute_cells(blobs[1]) is not None
        finally:
            k.close()
    k.lib.ckzg_hip_set_option(b"commit_wbits", 10)
    k.lib.ckzg_hip_set_option(b"proof_wbits", 8)
    k.lib.ckzg_hip_set_option(b"fk20_wbits", 0)
