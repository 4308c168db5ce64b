"""GPU parity for compute_cells_and_kzg_proofs (src/eip7594/eip7594.c:61-157) through the C-ABI:
consensus-spec vectors, random blobs against the CPU oracle, batch == single."""
import ctypes as C
import hashlib

import pytest

import golden_util as G
from test_gpu_commitment import rand_blob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", G.case_names("compute_cells"))
def test_golden_cells(hip, name):
    got, exp = G.run_case(hip, "compute_cells", name)
    assert got == exp


@pytest.mark.parametrize("name", G.case_names("compute_cells_and_kzg_proofs"))
def test_golden_cells_and_proofs(hip, name):
    got, exp = G.run_case(hip, "compute_cells_and_kzg_proofs", name)
    assert got == exp


def test_random_vs_oracle(hip, oracle):
    b = rand_blob(21, 0)
    cells, proofs = hip.compute_cells_and_kzg_proofs(b)
    ecells, eproofs = oracle.compute_cells_and_kzg_proofs(b)
    assert cells == ecells
    assert proofs == eproofs


def test_proofs_only_and_cells_only(hip):
    b = rand_blob(22, 1)
    cells, proofs = hip.compute_cells_and_kzg_proofs(b)
    c2, p2 = hip.compute_cells_and_kzg_proofs(b, True, False)
    c3, p3 = hip.compute_cells_and_kzg_proofs(b, False, True)
    assert c2 == cells and p2 is None and c3 is None and p3 == proofs


def test_batch_matches_single(hip):
    n = 5
    blobs = [rand_blob(23, i) for i in range(n)]
    bad = bytearray(blobs[3])
    bad[0:32] = b"\xff" * 32
    blobs[3] = bytes(bad)
    cells = C.create_string_buffer(n * 128 * 2048)
    proofs = C.create_string_buffer(n * 128 * 48)
    status = C.create_string_buffer(n)
    f = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    ret = f(cells, proofs, status, b"".join(blobs), n, C.addressof(hip.s))
    assert ret == 1 and list(status.raw) == [0, 0, 0, 1, 0]
    for i in (0, 1, 2, 4):
        c, p = hip.compute_cells_and_kzg_proofs(blobs[i])
        assert b"".join(c) == cells.raw[i * 128 * 2048:(i + 1) * 128 * 2048]
        assert b"".join(p) == proofs.raw[i * 128 * 48:(i + 1) * 128 * 48]


# ---- the FK20 (throughput) path, forced by direct_max = 0 ----

@pytest.mark.parametrize("name", [n for n in G.case_names("compute_cells_and_kzg_proofs") if "valid" in n])
def test_golden_cells_and_proofs_fk20_path(hip_fk20, name):
    got, exp = G.run_case(hip_fk20, "compute_cells_and_kzg_proofs", name)
    assert got == exp


@pytest.mark.parametrize("name", [n for n in G.case_names("recover_cells_and_kzg_proofs") if "valid" in n])
def test_golden_recover_fk20_path(hip_fk20, name):
    got, exp = G.run_case(hip_fk20, "recover_cells_and_kzg_proofs", name)
    assert got == exp


def test_direct_and_fk20_paths_agree(hip, hip_fk20):
    b = rand_blob(31, 0)
    assert hip.compute_cells_and_kzg_proofs(b) == hip_fk20.compute_cells_and_kzg_proofs(b)


def test_large_batch_takes_fk20_and_16_lane_msm_path(hip):
    # 40 blobs > direct_max: FK20 path (one-lane radix-4 G1 FFT steps at this size) with 5120 small MSMs ->
    # the 16-lanes-per-vector kernel
    n = 40
    base = [rand_blob(32, i) for i in range(4)]
    blobs = [base[i % 4] for i in range(n)]
    cells = C.create_string_buffer(n * 128 * 2048)
    proofs = C.create_string_buffer(n * 128 * 48)
    status = C.create_string_buffer(n)
    f = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    assert f(cells, proofs, status, b"".join(blobs), n, C.addressof(hip.s)) == 0
    single = [hip.compute_cells_and_kzg_proofs(b) for b in base]  # low-latency path
    for i in range(n):
        c, p = single[i % 4]
        assert b"".join(c) == cells.raw[i * 128 * 2048:(i + 1) * 128 * 2048]
        assert b"".join(p) == proofs.raw[i * 128 * 48:(i + 1) * 128 * 48]


@pytest.mark.parametrize("n", [2, 3, 5, 9, 16, 17, 24, 32, 33])
def test_small_batches_take_every_g1_fft_form(hip, oracle, n):
    """The two G1 transforms of FK20 take the form that fits the batch (fk20.hip): radix-8 steps on raw records with the
    three-wave ladder and two twiddles per workgroup (<= 8 blobs), with the two-wave ladder (9..16), with the one-wave
    quad ladder (17..48); radix-4 steps with the one-wave quad ladder (<= 128), radix-2 stages beyond.  Every form against the one-blob path (itself checked against the
    oracle on every vector), one blob of each batch against the oracle directly; a zero blob (all proofs at infinity:
    every ladder input is the point at infinity) and a constant polynomial ride along."""
    base = [rand_blob(77, i) for i in range(3)] + [bytes(131072), (b"\x00" * 31 + b"\x07") * 4096]
    blobs = [base[i % 5] for i in range(n)]
    cells = C.create_string_buffer(n * 128 * 2048)
    proofs = C.create_string_buffer(n * 128 * 48)
    status = C.create_string_buffer(n)
    f = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    assert f(cells, proofs, status, b"".join(blobs), n, C.addressof(hip.s)) == 0
    assert status.raw == bytes(n)
    single = [hip.compute_cells_and_kzg_proofs(b) for b in base]
    for i in range(n):
        c, p = single[i % 5]
        assert b"".join(c) == cells.raw[i * 128 * 2048:(i + 1) * 128 * 2048], i
        assert b"".join(p) == proofs.raw[i * 128 * 48:(i + 1) * 128 * 48], i
    ec, ep = oracle.compute_cells_and_kzg_proofs(base[1])
    assert b"".join(ep) == proofs.raw[128 * 48:2 * 128 * 48] and b"".join(ec) == cells.raw[128 * 2048:2 * 128 * 2048]
