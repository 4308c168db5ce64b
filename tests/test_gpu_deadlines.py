"""The wait deadline on real hardware (csrc/device.hpp "bounded waits"): the reference cannot block (straight-line
code, src/eip4844/eip4844.c:264-280; an internal failure is C_KZG_ERROR, returned: src/common/ret.h:24-29); this library
waits for the GPU and for other callers, and every such wait gives up at `wait_deadline_ms`.

A device wait that expires marks the device as not answering for the rest of the process, so the expiry itself is
provoked in a child process: with a deadline of one millisecond a 1024-blob commitment (10+ ms of kernels) cannot be
waited for.  What must hold: that call returns C_KZG_ERROR -- it does not hang and does not return garbage as OK --,
every later call on the device fails at once, the counters say what happened, ckzg_hip_debug_dump describes the
library without taking a lock, and free_trusted_setup and process exit do not block."""
import ctypes as C
import json
import os
import sys

import pytest

from watchdog import run_watched

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, json, os, sys, time
sys.path.insert(0, "tests")
from kzg_ctypes import HIP_SO, Kzg, KzgError
from test_gpu_commitment import rand_blob

out = {}
k = Kzg(HIP_SO, "", precompute=0)
lib = k.lib
lib.ckzg_hip_wait_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
def stats():
    v = (C.c_uint64 * 3)()
    assert lib.ckzg_hip_wait_stats(v, 3) == 3
    return [int(x) for x in v]
def dump():
    r, w = os.pipe()
    lib.ckzg_hip_debug_dump(w)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        data += chunk
    os.close(r)
    return data.decode()

out["stats_fresh"] = stats()
blob = rand_blob(601, 0)
want = k.blob_to_kzg_commitment(blob)
out["dump_idle"] = dump()
# option range
out["set_bad"] = [int(lib.ckzg_hip_set_option(b"wait_deadline_ms", C.c_int64(v))) for v in (0, -5, 10 ** 9)]
n = 1024
blobs = blob * n
res, st = C.create_string_buffer(48 * n), C.create_string_buffer(n)
f = lib.ckzg_hip_blob_to_kzg_commitment_batch
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
assert f(res, st, blobs, n, k.sp) == 0 and res.raw[:48] == want and res.raw[-48:] == want   # warm: arenas, staging
assert lib.ckzg_hip_set_option(b"wait_deadline_ms", C.c_int64(1)) == 0
t0 = time.perf_counter()
rc = f(res, st, blobs, n, k.sp)
out["expired_call"] = {"rc": int(rc), "seconds": round(time.perf_counter() - t0, 3)}
out["stats_after"] = stats()
assert lib.ckzg_hip_set_option(b"wait_deadline_ms", C.c_int64(30000)) == 0
t0 = time.perf_counter()
try:
    k.blob_to_kzg_commitment(blob)
    out["call_after"] = "OK"
except KzgError as e:
    out["call_after"] = str(e)
out["call_after_seconds"] = round(time.perf_counter() - t0, 3)
out["dump_after"] = dump()
t0 = time.perf_counter()
k.close()
out["free_seconds"] = round(time.perf_counter() - t0, 3)
print("RESULT " + json.dumps(out))
'''


def test_device_wait_deadline_on_the_gpu():
    r = run_watched([sys.executable, "-c", CHILD], cwd=ROOT, timeout=200, name="device_wait_deadline")
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(lines[-1][7:])
    assert out["stats_fresh"] == [30000, 0, 0]                       # default deadline, nothing expired, no device marked
    assert out["set_bad"] == [1, 1, 1]                               # C_KZG_BADARGS outside 1 ms .. 1 day
    assert "no thread is inside a wait" in out["dump_idle"] and "8 of 8 stream slots free" in out["dump_idle"]
    # the call that could not be waited for: C_KZG_ERROR (2), promptly; the wait was counted and the device marked
    assert out["expired_call"]["rc"] == 2 and out["expired_call"]["seconds"] < 5.0, out["expired_call"]
    assert out["stats_after"][1] >= 1 and out["stats_after"][2] == 1, out["stats_after"]
    assert "wait deadline exceeded" in r.stderr
    # afterwards: calls fail at once instead of queueing behind kernels that may still run; nothing blocks
    assert "C_KZG_RET" in out["call_after"] or "->" in out["call_after"], out["call_after"]
    assert out["call_after_seconds"] < 2.0 and out["free_seconds"] < 5.0, out
    assert "devices that stopped answering: mask 0x1" in out["dump_after"]


def test_debug_dump_next_to_busy_callers(hip):
    """ckzg_hip_debug_dump from one thread while 16 native threads share batch launches: it takes no lock it could wait
    for, so it returns whatever the others do, and what it prints is the combiner's queue."""
    import importlib.util
    import threading
    spec = importlib.util.spec_from_file_location("ckzg_fanout", os.path.join(ROOT, "c-kzg-4844_amd", "fanout.py"))
    fo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fo)
    from conftest import HIP_SO
    from test_gpu_commitment import rand_blob
    ins = [rand_blob(602, t % 4) for t in range(16)]
    want = [hip.blob_to_kzg_commitment(b) for b in ins]
    dumps = []
    stop = threading.Event()

    def dumper():
        while not stop.is_set():
            r, w = os.pipe()
            hip.lib.ckzg_hip_debug_dump(w)
            os.close(w)
            data = b""
            while True:
                chunk = os.read(r, 65536)
                if not chunk:
                    break
                data += chunk
            os.close(r)
            dumps.append(data.decode())

    th = threading.Thread(target=dumper)
    th.start()
    try:
        st, rets, outs = fo.run(hip, HIP_SO, fo.OP_COMMIT, ins, seconds=0.5)
    finally:
        stop.set()
        th.join()
    assert rets == [0] * 16 and outs == want
    assert len(dumps) >= 3 and all("end of ckzg_hip_debug_dump" in d or "registry of loaded KZGSettings" in d for d in dumps)
    assert any("combiner commit" in d for d in dumps)
    cs = fo.coalesce_stats(hip, 0)
    assert cs["rescued"] == 0 and cs["gave_up"] == 0, cs
