"""The BASELINE.json configurations at their full sizes, through size-independent properties
(SURVEY.md section 8c/8d): configs[3] verify_blob_kzg_proof_batch over 4096 blobs (the whole batch on
one GPU and as eight 512-blob shards), configs[4] recover_cells_and_kzg_proofs over a 256-row batch
with 64 of 128 cells, and 8192 cells through verify_cell_kzg_proof_batch."""
import ctypes as C

import pytest

from test_gpu_commitment import rand_blob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def material(oracle):
    """Expected values for the 8 base blobs come from the CPU oracle, never from the library under test."""
    blobs = [rand_blob(71, i) for i in range(8)]
    cm = [oracle.blob_to_kzg_commitment(b) for b in blobs]
    pr = [oracle.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    cp = [oracle.compute_cells_and_kzg_proofs(b) for b in blobs]
    return blobs, cm, pr, cp


@pytest.fixture(scope="module")
def hip_pre8():
    """BASELINE configs[4] names precompute=8: load exactly that (src/setup/setup.c:411-422 -> wbits = 8)."""
    from kzg_ctypes import HIP_SO, Kzg
    api = Kzg(HIP_SO, "", precompute=8)
    assert api.s.wbits == 8
    yield api
    api.close()


def _verify(hip, bb, cc, pp, n, off=0):
    f = hip.lib.verify_blob_kzg_proof_batch
    f.restype = C.c_int
    ok = C.c_bool(False)
    rc = f(C.byref(ok), C.c_char_p(bb[off * 131072:(off + n) * 131072]), C.c_char_p(cc[off * 48:(off + n) * 48]),
           C.c_char_p(pp[off * 48:(off + n) * 48]), C.c_uint64(n), hip.sp)
    return rc, ok.value


def test_verify_4096_blobs_whole_and_sharded(hip, material):
    blobs, cm, pr, _ = material
    n = 4096
    bb = b"".join(blobs[i % 8] for i in range(n))
    cc = b"".join(cm[i % 8] for i in range(n))
    pp = b"".join(pr[i % 8] for i in range(n))
    assert _verify(hip, bb, cc, pp, n) == (0, True)
    # eight shards of 512 (one per GPU in configs[3]): the conjunction of the shard verdicts is the verdict
    assert all(_verify(hip, bb, cc, pp, 512, off) == (0, True) for off in range(0, n, 512))
    # one wrong proof anywhere turns the batch, and exactly its shard, false
    bad_at = 3 * 512 + 77
    pp2 = pp[:bad_at * 48] + pr[(bad_at + 1) % 8] + pp[(bad_at + 1) * 48:]
    assert _verify(hip, bb, cc, pp2, n) == (0, False)
    verdicts = [_verify(hip, bb, cc, pp2, 512, off)[1] for off in range(0, n, 512)]
    assert verdicts == [i != 3 for i in range(8)]


def test_recover_256_rows_from_half_the_cells(hip_pre8, material):
    _, _, _, cp = material
    nb = 256
    keep = list(range(1, 128, 2))
    rows = [[cp[b % 8][0][i] for i in keep] for b in range(nb)]
    rc, rp = hip_pre8.recover_cells_and_kzg_proofs_batch(keep, rows)
    for b in range(nb):
        assert rc[b] == cp[b % 8][0], b
        assert rp[b] == cp[b % 8][1], b


def test_verify_8192_cells(hip, material):
    _, cm, _, cp = material
    n = 8192
    rows = [(i // 128) % 8 for i in range(n)]
    cols = [i % 128 for i in range(n)]
    f = hip.lib.verify_cell_kzg_proof_batch
    f.restype = C.c_int
    ccm = b"".join(cm[r] for r in rows)
    idx = (C.c_uint64 * n)(*cols)
    cells = b"".join(cp[r][0][c] for r, c in zip(rows, cols))
    prf = b"".join(cp[r][1][c] for r, c in zip(rows, cols))
    ok = C.c_bool(False)
    assert f(C.byref(ok), ccm, idx, cells, prf, C.c_uint64(n), hip.sp) == 0 and ok.value is True
    k = 5000
    bad = bytearray(cells)
    bad[k * 2048 + 31] ^= 1            # one field element of one cell
    assert f(C.byref(ok), ccm, idx, bytes(bad), prf, C.c_uint64(n), hip.sp) == 0 and ok.value is False
