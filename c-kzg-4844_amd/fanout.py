"""N native threads, each making ONE-UNIT calls of the unchanged ckzg.h API on a shared KZGSettings -- the
reference's only parallel shape (bindings/go/main_test.go:953-971: goroutines calling BlobToKZGCommitment /
ComputeCellsAndKZGProofs on their own blobs).  Python threads cannot drive 10^5 calls/s through the GIL, so the
callers are pthreads in ``libckzg_callers.so`` (csrc/fanout_callers.c, plain C against include/ckzg.h); this module
only hands them buffers.  Measurement / test driver: it calls nothing but the public C-ABI."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
CALLERS_SO = os.path.join(PKG, "libckzg_callers.so")

OP_COMMIT, OP_CELLS_PROOFS, OP_BLOB_PROOF, OP_RECOVER, OP_CELLS, OP_PROOFS, OP_VERIFY_BLOB = range(7)
CELLS_BYTES = 128 * 2048
PROOFS_BYTES = 128 * 48
OUT_STRIDE = {OP_COMMIT: 48, OP_BLOB_PROOF: 48, OP_CELLS_PROOFS: CELLS_BYTES + PROOFS_BYTES,
              OP_RECOVER: CELLS_BYTES + PROOFS_BYTES, OP_CELLS: CELLS_BYTES + PROOFS_BYTES,
              OP_PROOFS: CELLS_BYTES + PROOFS_BYTES, OP_VERIFY_BLOB: 1}

_lib = None


def _callers(libpath):
    """libckzg_callers.so leaves the ckzg.h symbols undefined: the library under test goes in first, globally."""
    global _lib
    if _lib is None:
        C.CDLL(libpath, mode=C.RTLD_GLOBAL)
        _lib = C.CDLL(CALLERS_SO)
        _lib.callers_run.restype = C.c_int
        _lib.callers_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.POINTER(C.c_double), C.POINTER(C.c_int)]
    return _lib


def run(kzg, libpath, op, inputs, threads=None, seconds=1.0, max_calls=0, aux=None, aux_n=0):
    """inputs: one bytes object per thread (blob, or the concatenated cells of a recover call).
    aux: None | one bytes object shared by every thread (recover: little-endian u64 cell indices, aux_n of them)
    | a list with one 48-byte commitment per thread (blob proofs) | a list with commitment + proof (96 bytes) per thread
    (OP_VERIFY_BLOB: the output byte is the verdict).
    Returns (stats dict, last return code per thread, last output bytes per thread)."""
    lib = _callers(libpath)
    threads = threads or len(inputs)
    assert len(inputs) == threads
    in_stride = len(inputs[0])
    assert all(len(b) == in_stride for b in inputs)
    ins = C.create_string_buffer(b"".join(inputs), in_stride * threads)
    if aux is None:
        auxs, aux_stride = None, 0
    elif isinstance(aux, (bytes, bytearray)):
        auxs, aux_stride = C.create_string_buffer(bytes(aux), len(aux)), 0
    else:
        assert len(aux) == threads
        aux_stride = len(aux[0])
        auxs = C.create_string_buffer(b"".join(aux), aux_stride * threads)
    out_stride = OUT_STRIDE[op]
    outs = C.create_string_buffer(out_stride * threads)
    stats = (C.c_double * 8)()
    rets = (C.c_int * threads)()
    rc = lib.callers_run(C.addressof(kzg.s), op, threads, float(seconds), int(max_calls), ins, in_stride,
                         auxs, aux_stride, aux_n, outs, out_stride, stats, rets)
    if rc != 0:
        raise RuntimeError("callers_run could not start %d threads" % threads)
    raw = outs.raw
    st = {"threads": threads, "calls": int(stats[0]), "not_ok": int(stats[1]), "seconds": stats[2],
          "calls_per_s": stats[0] / stats[2] if stats[2] > 0 else 0.0, "worst_call_ms": stats[3],
          "mean_call_ms": stats[4], "p50_call_ms": stats[5], "p99_call_ms": stats[6], "p999_call_ms": stats[7]}
    return st, list(rets), [raw[i * out_stride:(i + 1) * out_stride] for i in range(threads)]


def coalesce_stats(kzg, op_index):
    """ckzg_hip_coalesce_stats: op_index 0 commitment, 1/2/3 cells / proofs / both, 4 blob proof, 5 recover, 6 blob verification."""
    f = kzg.lib.ckzg_hip_coalesce_stats
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int]
    v = (C.c_uint64 * 9)()
    k = f(C.addressof(kzg.s), op_index, v, 9)
    names = ("calls", "solo", "batches", "batched", "largest", "run_us", "retried", "rescued", "gave_up")
    return {n: int(v[i]) for i, n in enumerate(names)} if k == 9 else None
