"""GPU tests for round-4 fixes: the verification's ladder fallback when HBM is REALLY too full for its call-time
table (the allocation-failure interposer of tests/failalloc never enters the runtime, so it cannot see an error the
runtime keeps), and the pointer checks of the *_device entry points."""
import ctypes as C

import pytest

from conftest import HIP_SO
from kzg_ctypes import Kzg
from test_gpu_commitment import rand_blob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def material(oracle):
    blobs = [rand_blob(404, i) for i in range(4)]
    cm = [oracle.blob_to_kzg_commitment(b) for b in blobs]
    pr = [oracle.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    return blobs, cm, pr


def test_batch_verification_when_hbm_is_too_full_for_the_call_time_table():
    """verify_blob_kzg_proof_batch builds a fixed-base table over the batch's points when its arena can hold it
    (434 MB of arena at n = 512) and falls back to ladder sums when it cannot (ckzg_api2.hip: verify_blobs_core).
    tools/oom_fallback_probe.py fills the HBM for real and calls it on a fresh KZGSettings with 900 MB left (the
    table fits) and with 330 MB left (it cannot): both calls must succeed with the right verdict for a valid batch
    and for one with a wrong proof.  The failed arena allocation of the second level leaves an out-of-memory error
    in the runtime's per-thread state; round 3 did not clear it and the next hipGetLastError() after a kernel launch
    would have reported it.  The probe runs in a process of its own: a device with no memory left at
    all makes the HSA runtime abort its process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from watchdog import run_watched
    r = run_watched([sys.executable, os.path.join(root, "tools", "oom_fallback_probe.py"), "900", "330"], timeout=280,
                    cwd=root, name="oom_fallback_probe")
    rows = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{") and '"rc"' in line]
    assert r.returncode == 0 and len(rows) == 2, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    for row in rows:
        assert row["rc"] == 0 and row["ok"] is True, rows
        assert row["rc_bad"] == 0 and row["ok_bad"] is False, rows
    assert rows[0]["free_mb"] >= 800 and rows[1]["free_mb"] <= 340, rows   # the second level cannot have held the table


_POINTER_CHECKS = r'''
import ctypes as C, json, sys
import torch                       # FIRST: its copy of the HIP runtime becomes the one the library binds to as well
torch.cuda.init()
sys.path.insert(0, "tests")
from kzg_ctypes import HIP_SO, Kzg
blob, cm0, pr0 = (bytes.fromhex(x) for x in json.loads(sys.stdin.read()))
hip = Kzg(HIP_SO, "", precompute=0)
lib = hip.lib
p, u64 = C.c_void_p, C.c_uint64
d_blob = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
d_out = torch.zeros(48, dtype=torch.uint8, device="cuda")
d_st = torch.zeros(1, dtype=torch.uint8, device="cuda")
h_out = C.create_string_buffer(48)
f = lib.ckzg_hip_blob_to_kzg_commitment_batch_device
f.restype = C.c_int
f.argtypes = [p, p, p, u64, p]
sp = C.addressof(hip.s)
assert f(d_out.data_ptr(), d_st.data_ptr(), d_blob.data_ptr(), 1, sp) == 0
assert bytes(d_out.cpu().numpy().tobytes()) == cm0
assert f(d_out.data_ptr(), None, d_blob.data_ptr(), 1, sp) == 0                       # the status array is optional
assert f(C.cast(h_out, p), d_st.data_ptr(), d_blob.data_ptr(), 1, sp) == 1            # host output buffer
assert f(d_out.data_ptr(), d_st.data_ptr(), C.cast(C.c_char_p(blob), p), 1, sp) == 1  # host blobs
assert f(None, d_st.data_ptr(), d_blob.data_ptr(), 1, sp) == 1
g = lib.ckzg_hip_compute_cells_and_kzg_proofs_batch_device
g.restype = C.c_int
g.argtypes = [p, p, p, p, u64, p]
d_proofs = torch.zeros(128 * 48, dtype=torch.uint8, device="cuda")
h_cells = C.create_string_buffer(128 * 2048)
assert g(None, d_proofs.data_ptr(), d_st.data_ptr(), d_blob.data_ptr(), 1, sp) == 0
assert g(C.cast(h_cells, p), d_proofs.data_ptr(), d_st.data_ptr(), d_blob.data_ptr(), 1, sp) == 1
v = lib.ckzg_hip_verify_blob_kzg_proof_batch_device
v.restype = C.c_int
v.argtypes = [p, p, p, p, u64, p]
ok = C.c_bool(False)
d_c = torch.frombuffer(bytearray(cm0), dtype=torch.uint8).cuda()
d_p = torch.frombuffer(bytearray(pr0), dtype=torch.uint8).cuda()
assert v(C.byref(ok), d_blob.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), 1, sp) == 0 and ok.value
assert v(C.byref(ok), d_blob.data_ptr(), C.cast(C.c_char_p(cm0), p), d_p.data_ptr(), 1, sp) == 1
hip.close()
print("POINTER_CHECKS_OK")
'''


def test_device_entry_points_reject_pointers_that_are_not_on_their_gpu(material, tmp_path):
    """The caller's device buffers come from torch here, so torch and the library must share ONE HIP runtime: torch
    brings its own copy of libamdhip64 (torch/lib, same soname), and whichever copy a process loads first is the one both
    bind to.  A process that has already loaded /opt/rocm's copy through the library -- this pytest session, once any
    fixture has run -- ends up with two runtimes, torch's pointers unknown to the library's and, on some boxes, torch
    finding "No HIP GPUs" (round-6 run 1: the test ran after the parity files for the first time).  The checks therefore
    run in a process of their own that imports torch first, as an application embedding both would."""
    import json
    import os
    import sys
    from watchdog import run_watched
    blobs, cm, pr = material
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inp = tmp_path / "in.json"
    inp.write_text(json.dumps([blobs[0].hex(), cm[0].hex(), pr[0].hex()]))
    r = run_watched([sys.executable, "-c", "import sys; sys.stdin = open(%r); exec(%r)" % (str(inp), _POINTER_CHECKS)],
                    cwd=root, timeout=280, name="round4_pointer_checks")
    assert r.returncode == 0 and "POINTER_CHECKS_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])


# ---------------------------------------------------------------------------------------------
# the graph of the lone one-blob commitment.  Round 4 CAPTURED it, and a capture that overlaps other threads' HIP calls
# faults inside this runtime (profiles/r04_capture_crash.txt); round 5 builds it node by node (msm.hip:
# commit_one_graph_build), which no HIP call of any other thread -- of this library or foreign to it -- can disturb.
# ---------------------------------------------------------------------------------------------

def _graph_stats(api):
    st = (C.c_uint64 * 3)()
    api.lib.ckzg_hip_commit_graph_stats(st)
    return list(st)


def test_lone_commitment_goes_out_as_a_graph(hip, oracle):
    blob = rand_blob(401, 0)
    want = oracle.blob_to_kzg_commitment(blob)
    before = _graph_stats(hip)
    for _ in range(20):
        assert hip.blob_to_kzg_commitment(blob) == want
    after = _graph_stats(hip)
    assert after[2] - before[2] >= 20, (before, after)       # every call launched as a graph
    assert after[1] == before[1], (before, after)            # no build failed
    hip.lib.ckzg_hip_set_option(b"commit_graph", 0)
    try:
        mid = _graph_stats(hip)
        assert hip.blob_to_kzg_commitment(blob) == want
        assert _graph_stats(hip)[2] == mid[2]
    finally:
        hip.lib.ckzg_hip_set_option(b"commit_graph", 1)


def test_graph_builds_survive_busy_library_and_foreign_hip_threads(hip, oracle):
    import threading
    blobs = [rand_blob(402, i) for i in range(4)]
    cm = [hip.blob_to_kzg_commitment(b) for b in blobs]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in blobs]
    assert cm[2] == oracle.blob_to_kzg_commitment(blobs[2])
    errs = []

    def work(t, rnd):
        try:
            for k in range(3):
                i = (t + k + rnd) % 4
                kind = (t + k + rnd) % 4
                if kind == 0 or kind == 2:
                    ok = hip.blob_to_kzg_commitment(blobs[i]) == cm[i]
                elif kind == 1:
                    ok = hip.compute_cells_and_kzg_proofs(blobs[i]) == cp[i]
                else:
                    ok = hip.verify_blob_kzg_proof_batch(blobs, cm, pr)
                if not ok:
                    errs.append((t, k, kind))
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    # a HIP user that is NOT this library: raw hipMalloc / hipMemcpy / hipFree from its own thread, for the whole test
    # (what RCCL's watchdog or PyTorch's allocator are to an embedding process)
    # ... through the SAME runtime the library runs on (a process has one runtime that owns the GPU): the copy of
    # libamdhip64 that is mapped already, by its path -- a bare dlopen("libamdhip64.so") can resolve to another copy that a
    # test imported earlier (torch/lib carries one under exactly that name) and that could never initialise
    loaded = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln and "/torch/" not in ln})
    assert loaded, "the library's HIP runtime is not mapped?"
    rt = C.CDLL(loaded[0])
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    stop = threading.Event()
    foreign = {"loops": 0, "bad": 0}

    def foreign_hip_user():
        host = C.create_string_buffer(1 << 20)
        while not stop.is_set():
            d = C.c_void_p()
            if rt.hipMalloc(C.byref(d), 1 << 20) != 0:
                foreign["bad"] += 1
                continue
            foreign["bad"] += rt.hipMemcpy(d, host, 1 << 20, 1) != 0      # hipMemcpyHostToDevice
            foreign["bad"] += rt.hipMemcpy(host, d, 1 << 20, 2) != 0      # hipMemcpyDeviceToHost
            foreign["bad"] += rt.hipFree(d) != 0
            foreign["loops"] += 1

    ft = threading.Thread(target=foreign_hip_user)
    ft.start()
    hip.lib.ckzg_hip_set_option(b"commit_graph", 2)   # every lone one-blob commitment builds its graph anew
    try:
        before = _graph_stats(hip)
        for rnd in range(25):
            th = [threading.Thread(target=work, args=(t, rnd)) for t in range(12)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            assert not errs, errs[:4]
            assert hip.blob_to_kzg_commitment(blobs[rnd % 4]) == cm[rnd % 4]
        after = _graph_stats(hip)
        assert after[0] - before[0] >= 25, (before, after)     # graphs were built, also while the others were inside
        assert after[1] == before[1], (before, after)          # and none of the builds failed
        assert after[2] - before[2] >= 25, (before, after)
    finally:
        hip.lib.ckzg_hip_set_option(b"commit_graph", 1)
        stop.set()
        ft.join()
    assert foreign["loops"] > 0 and foreign["bad"] == 0, foreign
    assert hip.blob_to_kzg_commitment(blobs[0]) == cm[0]
