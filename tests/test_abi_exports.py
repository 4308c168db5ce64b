"""CPU-side checks of the drop-in boundary: libckzg_hip.so loads without a GPU, exports every
symbol declared in include/ckzg.h and include/ckzg_hip.h, KZGSettings has the reference layout,
and a call without a usable GPU context fails loudly instead of falling back to the CPU."""
import ctypes as C
import os
import re

import pytest

from kzg_ctypes import HIP_SO, KZGSettings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for hdr in ("ckzg.h", "ckzg_hip.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b([a-z_0-9]+)\s*\(", src):
            n = m.group(1)
            if n.startswith(("ckzg_hip_", "load_", "free_", "blob_", "compute_", "verify_", "recover_", "bytes_")):
                names.add(n)
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(HIP_SO):
        pytest.fail("libckzg_hip.so not built; run __graft_entry__.build()")
    return C.CDLL(HIP_SO)


def test_header_declares_the_reference_api():
    names = declared_symbols()
    # the 12 public functions + 2 test-exposed internals of the reference
    for n in ["load_trusted_setup", "load_trusted_setup_file", "free_trusted_setup", "blob_to_kzg_commitment",
              "compute_kzg_proof", "compute_blob_kzg_proof", "verify_kzg_proof", "verify_blob_kzg_proof",
              "verify_blob_kzg_proof_batch", "compute_challenge", "compute_cells_and_kzg_proofs",
              "recover_cells_and_kzg_proofs", "verify_cell_kzg_proof_batch",
              "compute_verify_cell_kzg_proof_batch_challenge", "bytes_to_kzg_commitment", "bytes_from_bls_field"]:
        assert n in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_kzgsettings_layout_is_the_reference_abi():
    # src/setup/settings.h:27-79: 8 pointers + 2 size_t
    assert C.sizeof(KZGSettings) == 80
    offs = [getattr(KZGSettings, f).offset for f, _ in KZGSettings._fields_]
    assert offs == [0, 8, 16, 24, 32, 40, 48, 56, 64, 72]


def test_options_are_validated(lib):
    f = lib.ckzg_hip_set_option
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_int64]
    assert f(b"commit_wbits", 17) == 1
    assert f(b"fk20_wbits", 17) == 1
    assert f(b"streams", 0) == 1 and f(b"streams", 65) == 1 and f(b"streams", 8) == 0
    assert f(b"replicas", 0) == 1 and f(b"replicas", 1) == 0
    assert f(b"devices", -2) == 1 and f(b"devices", 0) == 0
    assert f(b"commit_wbits", 3) == 1
    assert f(b"nonsense", 1) == 1
    assert f(b"commit_wbits", 10) == 0
    assert f(b"fk20_wbits", 0) == 0


def test_no_silent_cpu_fallback(lib):
    # a zeroed / foreign KZGSettings has no GPU context: the hot-path calls must return
    # C_KZG_ERROR (2), never compute on the CPU
    s = KZGSettings()
    out = C.create_string_buffer(48)
    blob = bytes(131072)
    f = lib.blob_to_kzg_commitment
    f.restype = C.c_int
    assert f(out, blob, C.byref(s)) == 2
    g = lib.compute_cells_and_kzg_proofs
    g.restype = C.c_int
    cells = C.create_string_buffer(128 * 2048)
    assert g(cells, None, blob, C.byref(s)) == 2
    # freeing a zeroed struct is safe and idempotent (setup.c:162-190)
    fr = lib.free_trusted_setup
    fr.restype = None
    fr(C.byref(s))
    fr(C.byref(s))


def test_load_rejects_bad_arguments_before_touching_the_gpu(lib):
    s = KZGSettings()
    f = lib.load_trusted_setup
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64]
    g1 = bytes(4096 * 48)
    g2 = bytes(65 * 96)
    assert f(C.byref(s), g1, len(g1), g1, len(g1), g2, len(g2), 16) == 1   # precompute > 15
    assert f(C.byref(s), g1, len(g1) - 48, g1, len(g1), g2, len(g2), 0) == 1  # wrong sizes
    assert f(C.byref(s), g1, len(g1), g1, len(g1), g2, len(g2), 0) == 1   # not valid encodings


def test_null_arguments_are_rejected_not_dereferenced(lib):
    # (the reference dereferences these; found by clang --analyze over the host side)
    s = KZGSettings()
    f = lib.load_trusted_setup_file
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    assert f(None, None, 0) == 1
    assert f(C.byref(s), None, 0) == 1
    r = lib.ckzg_hip_recover_cells_and_kzg_proofs_batch
    r.restype = C.c_int
    r.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    cells = C.create_string_buffer(128 * 2048)
    assert r(cells, None, None, None, cells, 64, 1, C.byref(s)) == 1
    idx = (C.c_uint64 * 64)(*range(64))
    assert r(cells, None, None, idx, None, 64, 1, C.byref(s)) == 1
    g = lib.ckzg_hip_commit_graph_stats
    g.restype = None
    st = (C.c_uint64 * 3)(7, 7, 7)
    g(None)
    g(st)
    assert list(st) == [0, 0, 0]
    o = lib.ckzg_hip_set_option
    o.restype = C.c_int
    o.argtypes = [C.c_char_p, C.c_int64]
    assert o(b"commit_graph", 3) == 1 and o(b"commit_graph", -1) == 1 and o(b"commit_graph", 0) == 0 and o(b"commit_graph", 1) == 0
