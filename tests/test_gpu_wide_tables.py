"""GPU parity on the table widths the bench times and the reference's `precompute` can reach.

* `hip_wide` is EXACTLY bench.py's WIDE KZGSettings (commitment 16 bit + monomial/proof 16 bit + FK20 13 bit,
  238 GB): every commitment, cells(+proofs) and recover vector of the consensus-spec corpus goes through it, on
  the low-latency proof path (single calls) AND on the 13-bit FK20 path (batch calls above `direct_max`).
* Besides random blobs, blobs whose FK20 scalars sit on the edges of the GLV split and of the signed-window
  recoding (fk20_edge.py: every digit of the chosen scalars is +-2^(c-1)), for every k_msm_small lane form.
The reference's `precompute` = 10..15 and the other FK20 widths: tests/test_gpu_fk20_widths.py (its 116 / 206 GB
tables do not fit next to this module's 238 GB).
Reference: src/eip7594/fk20.c:222-247 (fixed-base MSM per column when precompute > 0), src/setup/setup.c:291-323."""
import ctypes as C

import pytest

import fk20_edge as E
import golden_util as G
from kzg_ctypes import HIP_SO, Kzg
from test_gpu_commitment import R, _batch, rand_blob
from test_gpu_round2 import _cells_batch, _edge_scalars, _restore

pytestmark = pytest.mark.gpu

WIDE = {"commit_wbits": 16, "proof_wbits": 16, "fk20_wbits": 13}   # == bench.py: WIDE
# (the cases with an expected output: "valid" is a substring of "invalid")
VALID_CP = [n for n in G.case_names("compute_cells_and_kzg_proofs") if G.get_case("compute_cells_and_kzg_proofs", n)[1] is not None]
VALID_REC = [n for n in G.case_names("recover_cells_and_kzg_proofs") if G.get_case("recover_cells_and_kzg_proofs", n)[1] is not None]


def _wbits(api):
    f = api.lib.ckzg_hip_table_wbits
    f.restype = C.c_int
    return tuple(int(f(api.sp, k)) for k in range(3))   # (commitment, FK20, proof)


@pytest.fixture(scope="module")
def hip_wide():
    api = Kzg(HIP_SO, "", precompute=0, options=dict(WIDE))
    _restore(api)
    # really built at these widths, not silently narrowed to what fits (device_ctx.hip: fit_wbits)
    assert _wbits(api) == (16, 13, 16)
    yield api
    api.close()


def edge_blob(wbits, salt=0):
    """FK20 scalars (row i, frequency f < 63) cycling through the GLV / window edge values for `wbits`."""
    vals = _edge_scalars(wbits)
    blob, _ = E.blob_with_fk20_scalars(lambda i, f: vals[(i * 63 + f + salt) % len(vals)])
    return blob


# ---------------------------------------------------------------------------------------------
# bench.py's WIDE settings
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", G.case_names("blob_to_kzg_commitment"))
def test_wide_commitment_vectors(hip_wide, name):
    got, exp = G.run_case(hip_wide, "blob_to_kzg_commitment", name)
    assert got == exp


@pytest.mark.parametrize("name", G.case_names("compute_cells_and_kzg_proofs"))
def test_wide_cells_and_proofs_vectors_direct_path(hip_wide, name):
    got, exp = G.run_case(hip_wide, "compute_cells_and_kzg_proofs", name)   # one blob <= direct_max: 16-bit monomial table
    assert got == exp


@pytest.mark.parametrize("name", G.case_names("compute_cells"))
def test_wide_cells_vectors(hip_wide, name):
    got, exp = G.run_case(hip_wide, "compute_cells", name)
    assert got == exp


def test_wide_cells_and_proofs_vectors_fk20_13bit(hip_wide):
    # every valid vector in ONE batch call of 7 (> direct_max = 4): FK20 on the 13-bit table, 896 small MSMs
    cases = [G.get_case("compute_cells_and_kzg_proofs", n) for n in VALID_CP]
    n = len(cases)
    assert n > 4
    rc, cells, proofs, st = _cells_batch(hip_wide, b"".join(inp["blob"] for inp, _ in cases), n)
    assert rc == 0 and not any(st.raw)
    for i, (_, exp) in enumerate(cases):
        assert cells.raw[i * 262144:(i + 1) * 262144] == b"".join(exp[0]), VALID_CP[i]
        assert proofs.raw[i * 6144:(i + 1) * 6144] == b"".join(exp[1]), VALID_CP[i]


@pytest.mark.parametrize("name", G.case_names("recover_cells_and_kzg_proofs"))
def test_wide_recover_vectors(hip_wide, name):
    got, exp = G.run_case(hip_wide, "recover_cells_and_kzg_proofs", name)
    assert got == exp


@pytest.mark.parametrize("name", VALID_REC)
def test_wide_recover_vectors_fk20_13bit(hip_wide, name):
    inp, exp = G.get_case("recover_cells_and_kzg_proofs", name)
    rows = 6   # > direct_max: the batch form takes FK20
    rc, rp = hip_wide.recover_cells_and_kzg_proofs_batch(inp["cell_indices"], [inp["cells"]] * rows)
    for b in range(rows):
        assert rc[b] == exp[0] and rp[b] == exp[1]


def test_wide_random_and_edge_blobs_vs_oracle(hip_wide, oracle):
    blobs = [rand_blob(131, 0), edge_blob(13), edge_blob(16, 5)]
    exp = [oracle.compute_cells_and_kzg_proofs(b) for b in blobs]
    for b, e in zip(blobs, exp):
        assert hip_wide.blob_to_kzg_commitment(b) == oracle.blob_to_kzg_commitment(b)
        got = hip_wide.compute_cells_and_kzg_proofs(b)            # direct path, 16-bit monomial table
        assert got[0] == e[0] and got[1] == e[1]
    # FK20 at 13 bits: 65 blobs = 8320 vectors (16 lanes per vector)
    n = 65
    order = [(5 * i + i // 3) % 3 for i in range(n)]
    rc, cells, proofs, st = _cells_batch(hip_wide, b"".join(blobs[k] for k in order), n)
    assert rc == 0 and not any(st.raw)
    for i in range(n):
        assert proofs.raw[i * 6144:(i + 1) * 6144] == b"".join(exp[order[i]][1]), i
        assert cells.raw[i * 262144:(i + 1) * 262144] == b"".join(exp[order[i]][0]), i


def test_wide_520_blob_batch_uses_the_8_lane_form(hip_wide, oracle):
    # 520 blobs = 66,560 vectors: k_msm_small<8> (msm.hip: msm_small_vectors_device), 13-bit digits
    blobs = [rand_blob(132, 0), edge_blob(13, 11)]
    exp = [oracle.compute_cells_and_kzg_proofs(b) for b in blobs]
    n = 520
    order = [(i + i // 7) % 2 for i in range(n)]
    rc, _, proofs, st = _cells_batch(hip_wide, b"".join(blobs[k] for k in order), n, want_cells=False)
    assert rc == 0 and not any(st.raw)
    praw = memoryview(proofs).cast("B")
    for i in range(n):
        assert praw[i * 6144:(i + 1) * 6144] == b"".join(exp[order[i]][1]), i


def test_wide_commitment_batch_of_1024_is_what_the_bench_times(hip_wide, oracle):
    # the headline launch shape (1024 blobs, one workgroup per blob, 16-bit table) against the oracle on the
    # blobs that differ, and against single calls everywhere
    base = [rand_blob(133, i) for i in range(3)]
    vals = _edge_scalars(16)
    base.append(b"".join(vals[(j * 7 + 3) % len(vals)].to_bytes(32, "big") for j in range(4096)))
    exp = [oracle.blob_to_kzg_commitment(b) for b in base]
    ret, outs, st = _batch(hip_wide, [base[(3 * i + i // 5) % 4] for i in range(1024)])
    assert ret == 0 and not any(st)
    assert outs == [exp[(3 * i + i // 5) % 4] for i in range(1024)]
