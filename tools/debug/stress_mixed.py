"""stress_mixed.py SECONDS [THREADS] [commit_graph 0|1|2] [foreign 0|1] -- the body of
tests/test_gpu_round2.py::test_mixed_concurrent_calls_are_correct in a loop: rounds of THREADS fresh Python threads, three
calls of five kinds each on one shared KZGSettings, every answer checked against the first (single-threaded) one.
Prints rounds done; exit code 1 on a wrong answer.  Run under LD_PRELOAD=tools/debug/libsegv_bt.so to get the native
backtrace of a crash.  foreign = 1 (default) adds a thread that is NOT this library and calls hipMalloc / hipMemcpy /
hipFree in a loop for the whole run -- what RCCL's watchdog or PyTorch's allocator are to an embedding process; with
commit_graph = 2 every lone one-blob commitment builds its graph anew meanwhile (round 4's CAPTURED graph faulted
inside libamdhip64 under exactly this; round 5's node-by-node graph must not).  Diagnostic tooling (GPU box)."""
import hashlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kzg_ctypes import HIP_SO, Kzg  # noqa: E402


def rand_blob(seed, i):
    out = bytearray()
    for j in range(4096):
        out += b"\x00" + hashlib.sha256(b"%d|%d|%d" % (seed, i, j)).digest()[:31]
    return bytes(out)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    graph = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    foreign_on = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    hip = Kzg(HIP_SO, "", precompute=0)
    assert hip.lib.ckzg_hip_set_option(b"commit_graph", graph) == 0
    import ctypes as C
    stop = threading.Event()
    foreign = {"loops": 0, "bad": 0}

    def foreign_hip_user():
        rt = C.CDLL("libamdhip64.so")
        rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rt.hipFree.argtypes = [C.c_void_p]
        host = C.create_string_buffer(1 << 20)
        while not stop.is_set():
            d = C.c_void_p()
            if rt.hipMalloc(C.byref(d), 1 << 20) != 0:
                foreign["bad"] += 1
                continue
            foreign["bad"] += rt.hipMemcpy(d, host, 1 << 20, 1) != 0
            foreign["bad"] += rt.hipMemcpy(host, d, 1 << 20, 2) != 0
            foreign["bad"] += rt.hipFree(d) != 0
            foreign["loops"] += 1

    ft = threading.Thread(target=foreign_hip_user)
    if foreign_on:
        ft.start()
    blobs = [rand_blob(82, i) for i in range(4)]
    cm = [hip.blob_to_kzg_commitment(b) for b in blobs]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in blobs]
    errs = []

    def work(t, rnd):
        try:
            for k in range(3):
                i = (t + k + rnd) % 4
                kind = (t + k + rnd) % 5
                if kind == 0:
                    ok = hip.blob_to_kzg_commitment(blobs[i]) == cm[i]
                elif kind == 1:
                    ok = hip.compute_cells_and_kzg_proofs(blobs[i]) == cp[i]
                elif kind == 2:
                    ok = hip.verify_blob_kzg_proof_batch(blobs, cm, pr)
                elif kind == 3:
                    keep = list(range(0, 128, 2))
                    ok = hip.recover_cells_and_kzg_proofs(keep, [cp[i][0][c] for c in keep]) == cp[i]
                else:
                    cols = list(range(16 * t % 128, 16 * t % 128 + 16))
                    ok = hip.verify_cell_kzg_proof_batch([cm[i]] * 16, cols, [cp[i][0][c] for c in cols],
                                                         [cp[i][1][c] for c in cols])
                if not ok:
                    errs.append((t, k, kind))
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    t_end = time.time() + seconds
    rounds = 0
    while time.time() < t_end and not errs:
        th = [threading.Thread(target=work, args=(t, rounds)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if rounds % 3 == 0 and hip.blob_to_kzg_commitment(blobs[rounds % 4]) != cm[rounds % 4]:   # a lone call: the graph path
            errs.append(("lone", rounds))
        rounds += 1
        if rounds % 20 == 0:
            print("rounds", rounds, flush=True)
    stop.set()
    if foreign_on:
        ft.join()
    st = (C.c_uint64 * 3)()
    hip.lib.ckzg_hip_commit_graph_stats(st)
    print("stress_mixed: commit_graph=%d threads=%d rounds=%d errs=%s graphs built=%d failed builds=%d launches=%d foreign=%s"
          % (graph, nthreads, rounds, errs[:4], st[0], st[1], st[2], foreign), flush=True)
    if foreign["bad"] or st[1]:
        errs.append("foreign HIP calls failed or graph builds failed")
    hip.close()
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
