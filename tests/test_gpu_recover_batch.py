"""GPU parity for the additive batch form of recover_cells_and_kzg_proofs
(ckzg_hip_recover_cells_and_kzg_proofs_batch; per-row semantics of src/eip7594/eip7594.c:177-304):
rows of a block that all hold the same columns are recovered in one call and must equal the
oracle's / the single-call results, row by row."""
import ctypes as C
import random

import pytest

from test_gpu_commitment import rand_blob

pytestmark = pytest.mark.gpu


def _rows(hip, seed, nb):
    blobs = [rand_blob(seed, i) for i in range(nb)]
    return blobs, [hip.compute_cells_and_kzg_proofs(b) for b in blobs]


@pytest.mark.parametrize("pattern", ["even", "first_half", "random_70", "all"])
def test_batch_recover_matches_full_rows(hip, oracle, pattern):
    nb = 3
    blobs, full = _rows(hip, 61, nb)
    keep = {"even": list(range(0, 128, 2)), "first_half": list(range(64)),
            "random_70": sorted(random.Random(5).sample(range(128), 70)), "all": list(range(128))}[pattern]
    rc, rp = hip.recover_cells_and_kzg_proofs_batch(keep, [[full[b][0][i] for i in keep] for b in range(nb)])
    for b in range(nb):
        assert rc[b] == full[b][0]
        assert rp[b] == full[b][1]
    ec, ep = oracle.recover_cells_and_kzg_proofs(keep, [full[0][0][i] for i in keep])
    assert rc[0] == ec and rp[0] == ep


def test_batch_recover_fk20_path_and_outputs_optional(hip_fk20):
    nb = 2
    blobs, full = _rows(hip_fk20, 62, nb)
    keep = list(range(1, 128, 2))
    rows = [[full[b][0][i] for i in keep] for b in range(nb)]
    rc, rp = hip_fk20.recover_cells_and_kzg_proofs_batch(keep, rows)
    c_only, none_p = hip_fk20.recover_cells_and_kzg_proofs_batch(keep, rows, True, False)
    none_c, p_only = hip_fk20.recover_cells_and_kzg_proofs_batch(keep, rows, False, True)
    assert none_p is None and none_c is None
    for b in range(nb):
        assert rc[b] == full[b][0] and rp[b] == full[b][1]
        assert c_only[b] == rc[b] and p_only[b] == rp[b]


def test_batch_recover_flags_the_bad_row(hip):
    nb = 3
    blobs, full = _rows(hip, 63, nb)
    keep = list(range(64, 128))
    rows = [[full[b][0][i] for i in keep] for b in range(nb)]
    bad = bytearray(rows[1][5])
    bad[32:64] = b"\xff" * 32
    rows[1][5] = bytes(bad)
    f = hip.lib.ckzg_hip_recover_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    rc = C.create_string_buffer(nb * 128 * 2048)
    rp = C.create_string_buffer(nb * 128 * 48)
    st = C.create_string_buffer(nb)
    idx = (C.c_uint64 * len(keep))(*keep)
    ret = f(rc, rp, st, idx, b"".join(b"".join(r) for r in rows), C.c_uint64(len(keep)), C.c_uint64(nb), hip.sp)
    assert ret == 1  # C_KZG_BADARGS
    assert list(st.raw) == [0, 1, 0]
    for b in (0, 2):
        assert rc.raw[b * 128 * 2048:(b + 1) * 128 * 2048] == b"".join(full[b][0])
        assert rp.raw[b * 128 * 48:(b + 1) * 128 * 48] == b"".join(full[b][1])


def test_batch_recover_argument_checks(hip):
    blobs, full = _rows(hip, 64, 1)
    f = hip.lib.ckzg_hip_recover_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    rc = C.create_string_buffer(128 * 2048)

    def call(keep):
        idx = (C.c_uint64 * len(keep))(*keep)
        data = b"".join(full[0][0][i % 128] for i in keep)
        return f(rc, None, None, idx, data, C.c_uint64(len(keep)), C.c_uint64(1), hip.sp)

    assert call(list(range(63))) == 1                      # fewer than half the cells
    assert call(list(range(63)) + [62]) == 1               # not strictly ascending
    assert call(list(range(63)) + [128]) == 1              # index out of range
    assert call(list(range(64))) == 0
    idx = (C.c_uint64 * 64)(*range(64))
    assert f(None, None, None, idx, b"", C.c_uint64(64), C.c_uint64(1), hip.sp) == 1   # no output requested
    assert f(rc, None, None, idx, b"", C.c_uint64(64), C.c_uint64(0), hip.sp) == 0     # empty batch
