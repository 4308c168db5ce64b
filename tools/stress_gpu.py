#!/usr/bin/env python3
"""Soak test of the re-entrant C-ABI: T threads share ONE KZGSettings (loaded with async_tables, so the tables change
underneath them for the first seconds) and issue a random mix of calls -- single and batch commitments, cells + proofs,
blob-batch verification (small, mid-size and pipelined), cell-batch verification, recovery -- for a given time; every
result is compared with values computed once by the CPU oracle.  One more thread ("churn", on by default) meanwhile
loads and frees OTHER KZGSettings in the same process -- plain loads, progressive loads freed while their tables are
still being widened, progressive loads waited for -- and checks a commitment and a cells + proofs call on each: the
registry of GPU contexts, the widener threads and the device allocator are shared with the callers above.
Prints one JSON line; exit code 1 on any mismatch.
usage: python tools/stress_gpu.py [seconds=60] [threads=8] [churn=1]"""
import ctypes as C
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import __graft_entry__ as ge  # noqa: E402
from test_gpu_commitment import rand_blob  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    churn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    mod = ge.load_package()
    orc = mod.Kzg(os.path.join(ROOT, "oracle", "liboracle.so"), "okzg_")
    blobs = [rand_blob(555, i) for i in range(4)]
    cm = [orc.blob_to_kzg_commitment(b) for b in blobs]
    pr = [orc.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    cp = [orc.compute_cells_and_kzg_proofs(b) for b in blobs]
    orc.close()
    hip = mod.Kzg(mod.HIP_SO, options={"async_tables": 1, "commit_wbits": 15, "proof_wbits": 14, "fk20_wbits": 12, "streams": 6})
    hip.lib.ckzg_hip_set_option(b"async_tables", 0)
    sp = C.addressof(hip.s)
    fb = hip.lib.ckzg_hip_blob_to_kzg_commitment_batch
    fb.restype = C.c_int
    fb.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    counts, errors = {}, []
    lock = threading.Lock()
    t_end = time.perf_counter() + seconds

    def note(op):
        with lock:
            counts[op] = counts.get(op, 0) + 1

    def fail(msg):
        with lock:
            errors.append(msg)

    def worker(tid):
        rnd = random.Random(1000 + tid)
        while time.perf_counter() < t_end and not errors:
            op = rnd.choice(["commit", "commit", "commit_batch", "cells", "cells", "verify_small", "verify_mid", "verify_piped",
                             "verify_cells", "recover"])
            k = rnd.randrange(4)
            try:
                if op == "commit":
                    if hip.blob_to_kzg_commitment(blobs[k]) != cm[k]:
                        fail("commit")
                elif op == "commit_batch":
                    n = rnd.choice([2, 63, 65, 200, 300])
                    order = [rnd.randrange(4) for _ in range(n)]
                    out, st = C.create_string_buffer(48 * n), C.create_string_buffer(n)
                    rc = fb(out, st, b"".join(blobs[j] for j in order), n, sp)
                    if rc != 0 or out.raw != b"".join(cm[j] for j in order):
                        fail("commit_batch n=%d rc=%d" % (n, rc))
                elif op == "cells":
                    got = hip.compute_cells_and_kzg_proofs(blobs[k])
                    if got[0] != cp[k][0] or got[1] != cp[k][1]:
                        fail("cells")
                elif op in ("verify_small", "verify_mid", "verify_piped"):
                    n = {"verify_small": rnd.choice([1, 2, 3]), "verify_mid": rnd.choice([5, 40, 130]), "verify_piped": 1030}[op]
                    order = [rnd.randrange(4) for _ in range(n)]
                    p = [pr[j] for j in order]
                    want = True
                    if rnd.random() < 0.4:
                        at = rnd.randrange(n)
                        p[at] = pr[(order[at] + 1) % 4]
                        want = False
                    if hip.verify_blob_kzg_proof_batch([blobs[j] for j in order], [cm[j] for j in order], p) != want:
                        fail("%s n=%d want=%s" % (op, n, want))
                elif op == "verify_cells":
                    n = rnd.choice([1, 17, 128, 300])
                    ent = [(rnd.randrange(4), rnd.randrange(128)) for _ in range(n)]
                    prf = [cp[b][1][c] for b, c in ent]
                    want = True
                    if rnd.random() < 0.4:
                        at = rnd.randrange(n)
                        prf[at] = cp[(ent[at][0] + 1) % 4][1][ent[at][1]]
                        want = False
                    if hip.verify_cell_kzg_proof_batch([cm[b] for b, _ in ent], [c for _, c in ent], [cp[b][0][c] for b, c in ent], prf) != want:
                        fail("verify_cells n=%d want=%s" % (n, want))
                else:
                    keep = sorted(rnd.sample(range(128), rnd.choice([64, 70, 100])))
                    rows = rnd.choice([1, 1, 9])
                    if rows == 1:
                        got = hip.recover_cells_and_kzg_proofs(keep, [cp[k][0][i] for i in keep])
                        if got[0] != cp[k][0] or got[1] != cp[k][1]:
                            fail("recover")
                    else:
                        ks = [rnd.randrange(4) for _ in range(rows)]
                        rc_, rp_ = hip.recover_cells_and_kzg_proofs_batch(keep, [[cp[j][0][i] for i in keep] for j in ks])
                        if any(rc_[r] != cp[j][0] or rp_[r] != cp[j][1] for r, j in enumerate(ks)):
                            fail("recover_batch")
            except Exception as e:  # noqa: BLE001
                fail("%s raised %s" % (op, e))
            note(op)

    def churner():
        rnd = random.Random(77)
        while time.perf_counter() < t_end and not errors:
            mode = rnd.choice(["plain", "freed_while_widening", "widened"])
            try:
                opts = {"async_tables": 0 if mode == "plain" else 1, "commit_wbits": rnd.choice([10, 12]),
                        "proof_wbits": rnd.choice([8, 10]), "fk20_wbits": rnd.choice([8, 9]), "streams": 2}
                other = mod.Kzg(mod.HIP_SO, options=opts)
                if mode == "widened":
                    other.lib.ckzg_hip_wait_tables(C.c_void_p(C.addressof(other.s)))
                k = rnd.randrange(4)
                if other.blob_to_kzg_commitment(blobs[k]) != cm[k]:
                    fail("churn %s: commitment" % mode)
                got = other.compute_cells_and_kzg_proofs(blobs[k])
                if got[0] != cp[k][0] or got[1] != cp[k][1]:
                    fail("churn %s: cells" % mode)
                other.close()
            except Exception as e:  # noqa: BLE001
                fail("churn %s raised %s" % (mode, e))
            note("load_free_" + mode)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    if churn:
        th.append(threading.Thread(target=churner))
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    hip.lib.ckzg_hip_wait_tables(C.c_void_p(sp))
    wb = hip.lib.ckzg_hip_table_wbits
    wb.restype = C.c_int
    out = {"seconds": round(time.perf_counter() - t0, 1), "threads": nthreads, "calls": counts, "total_calls": sum(counts.values()),
           "final_table_wbits": [int(wb(C.c_void_p(sp), i)) for i in range(3)], "errors": errors[:10]}
    print(json.dumps(out))
    hip.close()
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
