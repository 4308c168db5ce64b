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
    # 40 blobs > direct_max: FK20 path (radix-8 G1 FFT steps with the one-wave quad ladder at this size: 17..48 blobs)
    # with 5120 small MSMs -> the one-wave-per-vector kernel
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


# The sizes on BOTH sides of every point where the G1 transforms or the small MSMs change form (fk20.hip:
# r8_pipe_max_transforms = 16, r8_max_transforms = 48, r4_max_transforms = 128; two twiddles per workgroup up to 8;
# msm.hip msm_small_vectors_device: one wave per vector below 8192 vectors = 64 blobs, 16 lanes per vector from there,
# 8 lanes from 65,536 vectors = 512 blobs), plus interior sizes.
FORM_SIZES = [2, 3, 5, 8, 9, 16, 17, 24, 32, 33, 48, 49, 63, 64, 128, 129, 511, 512]


@pytest.fixture(scope="module")
def form_blobs(oracle):
    """five distinct blobs -- three random, the zero blob (all proofs at infinity: every ladder input is the point at
    infinity), a constant polynomial -- and what the ORACLE says their cells and proofs are"""
    base = [rand_blob(77, i) for i in range(3)] + [bytes(131072), (b"\x00" * 31 + b"\x07") * 4096]
    exp = []
    for b in base:
        c, p = oracle.compute_cells_and_kzg_proofs(b)
        exp.append((b"".join(c), b"".join(p), c))
    return base, exp


@pytest.mark.parametrize("n", FORM_SIZES)
def test_small_batches_take_every_g1_fft_form(hip, form_blobs, n):
    """The two G1 transforms of FK20 take the form that fits the batch (fk20.hip; the reference's single form:
    src/eip7594/fft.c:164-240 inside src/eip7594/fk20.c:139-286): radix-8 steps on raw records with the three-wave
    ladder and two twiddles per workgroup (<= 8 blobs), with the two-wave ladder (9..16), with the one-wave quad ladder
    (17..48); radix-4 steps with the one-wave quad ladder (49..128), radix-2 stages beyond.  EVERY blob of every batch
    against the oracle, at the last size of each form and the first of the next."""
    base, exp = form_blobs
    blobs = [base[(i + i // 5) % 5] for i in range(n)]
    cells = C.create_string_buffer(n * 128 * 2048)
    proofs = C.create_string_buffer(n * 128 * 48)
    status = C.create_string_buffer(n)
    f = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    assert f(cells, proofs, status, b"".join(blobs), n, C.addressof(hip.s)) == 0
    assert status.raw == bytes(n)
    craw, praw = memoryview(cells).cast("B"), memoryview(proofs).cast("B")
    for i in range(n):
        ec, ep, _ = exp[(i + i // 5) % 5]
        assert praw[i * 128 * 48:(i + 1) * 128 * 48] == ep, i
        assert craw[i * 128 * 2048:(i + 1) * 128 * 2048] == ec, i


@pytest.mark.parametrize("rows", [8, 9, 16, 17, 32, 48, 49, 128, 129])
def test_recover_batches_take_every_g1_fft_form(hip, form_blobs, rows):
    """recover_cells_and_kzg_proofs over `rows` rows that hold the same 64 columns (src/eip7594/eip7594.c:177-304 per
    row): the recovered polynomial goes through the same FK20 forms; 32 rows is one GPU's shard of BASELINE configs[4]
    on an 8-GPU node.  Every row against the oracle's cells and proofs of the blob it was cut from."""
    base, exp = form_blobs
    idx = [(3 * j + 1) % 128 for j in range(128) if j % 2 == 0]   # 64 columns, not sorted by construction ...
    idx = sorted(set(idx))                                         # ... the API wants them ascending
    assert len(idx) == 64
    which = [(r + r // 3) % 5 for r in range(rows)]
    row_cells = [[exp[w][2][j] for j in idx] for w in which]
    rc, rp = hip.recover_cells_and_kzg_proofs_batch(idx, row_cells)
    for r, w in enumerate(which):
        assert b"".join(rc[r]) == exp[w][0], r
        assert b"".join(rp[r]) == exp[w][1], r
