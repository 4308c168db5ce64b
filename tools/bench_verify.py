#!/usr/bin/env python3
"""Secondary measurements: verify_blob_kzg_proof_batch (BASELINE configs[3] shape, one GPU's shard)
and recover_cells_and_kzg_proofs (configs[4] shape), through the C-ABI with host pointers."""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def blob(i):
    return b"".join(b"\x00" + hashlib.sha256(b"v%d|%d" % (i, j)).digest()[:31] for j in range(4096))


def main():
    mod = ge.load_package()
    hip = mod.Kzg(mod.HIP_SO, options={"commit_wbits": 13, "fk20_wbits": 8, "proof_wbits": 13})
    nmax = int(os.environ.get("NVERIFY", "512"))
    uniq = [blob(i) for i in range(8)]
    commits = [hip.blob_to_kzg_commitment(b) for b in uniq]
    t = time.perf_counter()
    proofs = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(uniq, commits)]
    print("compute_blob_kzg_proof: %.2f ms/call" % ((time.perf_counter() - t) / 8 * 1e3))
    t = time.perf_counter()
    for b, c, p in zip(uniq, commits, proofs):
        assert hip.verify_blob_kzg_proof(b, c, p)
    print("verify_blob_kzg_proof: %.2f ms/call" % ((time.perf_counter() - t) / 8 * 1e3))
    fv = hip.lib.verify_blob_kzg_proof_batch
    fv.restype = C.c_int
    fv.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
    for n in (8, 64, 512, nmax):
        bl = [uniq[i % 8] for i in range(n)]
        cm = [commits[i % 8] for i in range(n)]
        pr = [proofs[i % 8] for i in range(n)]
        # time the C call itself (the ctypes wrapper's b"".join of n blobs is harness overhead)
        bb, cc, pp = b"".join(bl), b"".join(cm), b"".join(pr)
        okv = C.c_bool(False)
        fv(C.byref(okv), bb, cc, pp, min(n, 2), C.addressof(hip.s))
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            rc = fv(C.byref(okv), bb, cc, pp, n, C.addressof(hip.s))
            best = min(best, time.perf_counter() - t)
        print("verify_blob_kzg_proof_batch n=%d: %.1f ms -> %.0f blobs/s (rc=%d ok=%s)" % (n, best * 1e3, n / best, rc, okv.value))
    pr[3] = proofs[0]
    assert not hip.verify_blob_kzg_proof_batch(bl, cm, pr)
    f = hip.lib.ckzg_hip_compute_blob_kzg_proof_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
    nb = 512
    out = C.create_string_buffer(48 * nb)
    stt = C.create_string_buffer(nb)
    bb, cc = b"".join(uniq[i % 8] for i in range(nb)), b"".join(commits[i % 8] for i in range(nb))
    f(out, stt, bb, cc, nb, C.addressof(hip.s))
    t = time.perf_counter()
    rc = f(out, stt, bb, cc, nb, C.addressof(hip.s))
    dt = time.perf_counter() - t
    print("compute_blob_kzg_proof batch n=%d: %.1f ms -> %.0f proofs/s (rc=%d)" % (nb, dt * 1e3, nb / dt, rc))
    assert out.raw[:48] == proofs[0]
    cells, cproofs = hip.compute_cells_and_kzg_proofs(uniq[0])
    for name, idx in (("first half missing", list(range(64, 128))), ("every other cell", list(range(0, 128, 2)))):
        hip.recover_cells_and_kzg_proofs(idx, [cells[i] for i in idx])
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            rc, rp = hip.recover_cells_and_kzg_proofs(idx, [cells[i] for i in idx])
            ts.append(time.perf_counter() - t)
        assert rc == cells and rp == cproofs
        print("recover_cells_and_kzg_proofs (%s): %.2f ms/call" % (name, sorted(ts)[2] * 1e3))
    n = 128
    t = time.perf_counter()
    ok = hip.verify_cell_kzg_proof_batch([commits[0]] * n, list(range(n)), cells, cproofs)
    print("verify_cell_kzg_proof_batch n=128: %.1f ms (ok=%s)" % ((time.perf_counter() - t) * 1e3, ok))
    hip.close()


if __name__ == "__main__":
    main()
