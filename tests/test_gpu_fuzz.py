"""Differential fuzzing of the C-ABI against the CPU oracle over malformed and edge inputs (hypothesis), the
counterpart of the reference's fuzz targets (fuzz/fuzz_targets/fuzz_verify_cell_kzg_proof_batch.rs,
fuzz_recover_cells_and_kzg_proofs.rs, fuzz_verify_blob_kzg_proof_batch.rs, fuzz_compute_kzg_proof.rs,
fuzz_blob_to_kzg_commitment.rs): both sides go through the same ctypes wrapper, so for every generated input the
product must fail exactly when the oracle fails (same C_KZG_RET) and otherwise return the same bytes / verdict.
Generated: list lengths, duplicate / unsorted / out-of-range cell indices, non-canonical field elements at random
positions, invalid point encodings (flag bits, x >= p, x not on the curve, a curve point outside the subgroup) at
random positions inside batches that take the GPU paths.  Deterministic (derandomize) so that a failure reproduces."""
import random

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from kzg_ctypes import KzgError
from test_gpu_commitment import R, rand_blob

pytestmark = pytest.mark.gpu

P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
W8192 = pow(7, (R - 1) // 8192, R)
FUZZ = dict(deadline=None, derandomize=True, database=None,
            suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])


def outcome(fn, *args):
    try:
        return ("ok", fn(*args))
    except KzgError as e:
        return ("err", str(e).rsplit(" ", 1)[-1])   # the C_KZG_RET value


def _not_in_subgroup_point():
    """Compressed encoding of a curve point outside G1 (y^2 = x^3 + 4 has cofactor h != 1): the first small x works."""
    x = 4
    while True:
        x += 1
        y2 = (pow(x, 3, P_MOD) + 4) % P_MOD
        y = pow(y2, (P_MOD + 1) // 4, P_MOD)
        if y * y % P_MOD != y2:
            continue
        enc = bytearray(x.to_bytes(48, "big"))
        enc[0] |= 0x80 | (0x20 if y > P_MOD - y else 0)
        return bytes(enc)   # (a random curve point is in G1 with probability 1/h ~ 2^-125)


@pytest.fixture(scope="module")
def material(oracle):
    blobs = [rand_blob(191, i) for i in range(4)]
    cm = [oracle.blob_to_kzg_commitment(b) for b in blobs]
    pr = [oracle.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    cp = [oracle.compute_cells_and_kzg_proofs(b) for b in blobs]
    bad_points = [
        bytes(48),                                        # compression flag not set
        b"\xc0" + bytes(46) + b"\x01",                    # infinity flag with a non-zero x
        b"\xe0" + bytes(47),                              # infinity with the sign flag
        b"\x80" + bytes(47),                              # x = 0: not on the curve (4 is a square? y^2 = 4 -> on curve) -- whatever the oracle says
        (b"\x9f" + b"\xff" * 47),                         # x >= p
        b"\x80" + bytes(46) + b"\x03",                    # x = 3: 31 is not a square mod p -> not on the curve, or is: oracle decides
        _not_in_subgroup_point(),
        b"\xc0" + bytes(47),                              # the valid point at infinity
    ]
    return blobs, cm, pr, cp, bad_points


def _noncanonical(rnd):
    return rnd.choice([R, R + 1, 2 ** 256 - 1, R + rnd.randrange(2 ** 200)]).to_bytes(32, "big")


# ---------------------------------------------------------------------------------------------

@settings(max_examples=40, **FUZZ)
@given(st.data())
def test_fuzz_verify_cell_kzg_proof_batch(hip, oracle, material, data):
    blobs, cm, pr, cp, bad_points = material
    rnd = random.Random(data.draw(st.integers(0, 2 ** 32)))
    n = data.draw(st.sampled_from([0, 1, 2, 3, 17, 64, 65, 130]))
    entries = [(rnd.randrange(4), rnd.randrange(128)) for _ in range(n)]
    if n >= 2 and rnd.random() < 0.5:
        entries[1] = entries[0]                            # a duplicate (allowed by the reference)
    commitments = [cm[b] for b, _ in entries]
    idx = [c for _, c in entries]
    cells = [cp[b][0][c] for b, c in entries]
    proofs = [cp[b][1][c] for b, c in entries]
    for _ in range(data.draw(st.integers(0, 2))):
        if n == 0:
            break
        at = rnd.randrange(n)
        kind = data.draw(st.sampled_from(["index_oob", "index_other", "cell_noncanonical", "cell_value", "proof_swap",
                                          "proof_bad", "commit_bad", "commit_other", "index_huge"]))
        if kind == "index_oob":
            idx[at] = 128 + rnd.randrange(3)
        elif kind == "index_huge":
            idx[at] = 2 ** 64 - 1 - rnd.randrange(2)
        elif kind == "index_other":
            idx[at] = (idx[at] + 1 + rnd.randrange(127)) % 128
        elif kind == "cell_noncanonical":
            c = bytearray(cells[at])
            pos = rnd.randrange(64)
            c[32 * pos:32 * pos + 32] = _noncanonical(rnd)
            cells[at] = bytes(c)
        elif kind == "cell_value":
            c = bytearray(cells[at])
            c[rnd.randrange(2048) | 1] ^= 1 << rnd.randrange(8)   # stays canonical with overwhelming probability
            cells[at] = bytes(c)
        elif kind == "proof_swap":
            proofs[at] = proofs[(at + 1) % n] if n > 1 else pr[0]
        elif kind == "proof_bad":
            proofs[at] = rnd.choice(bad_points)
        elif kind == "commit_bad":
            commitments[at] = rnd.choice(bad_points)
        else:
            commitments[at] = cm[(entries[at][0] + 1) % 4]
    got = outcome(hip.verify_cell_kzg_proof_batch, commitments, idx, cells, proofs)
    exp = outcome(oracle.verify_cell_kzg_proof_batch, commitments, idx, cells, proofs)
    assert got == exp


@settings(max_examples=14, **FUZZ)
@given(st.data())
def test_fuzz_recover_cells_and_kzg_proofs(hip, oracle, material, data):
    blobs, cm, pr, cp, _ = material
    rnd = random.Random(data.draw(st.integers(0, 2 ** 32)))
    b = rnd.randrange(4)
    k = data.draw(st.sampled_from([0, 1, 63, 64, 65, 100, 127, 128]))
    idx = sorted(rnd.sample(range(128), k))
    cells = [cp[b][0][i] for i in idx]
    kind = data.draw(st.sampled_from(["none", "none", "unsorted", "duplicate", "oob", "noncanonical", "wrong_cell", "too_many"]))
    if kind == "unsorted" and k >= 2:
        i = rnd.randrange(k - 1)
        idx[i], idx[i + 1] = idx[i + 1], idx[i]
        cells[i], cells[i + 1] = cells[i + 1], cells[i]
    elif kind == "duplicate" and k >= 2:
        idx[1] = idx[0]
    elif kind == "oob" and k >= 1:
        idx[-1] = 128 + rnd.randrange(2)
    elif kind == "noncanonical" and k >= 1:
        at = rnd.randrange(k)
        c = bytearray(cells[at])
        pos = rnd.randrange(64)
        c[32 * pos:32 * pos + 32] = _noncanonical(rnd)
        cells[at] = bytes(c)
    elif kind == "wrong_cell" and k >= 1:
        # a cell of ANOTHER blob: the inputs are no longer a codeword; for k > 64 the reference still "recovers"
        # something (it trusts its input) -- whatever it returns, the product must return the same
        at = rnd.randrange(k)
        cells[at] = cp[(b + 1) % 4][0][idx[at] % 128]
    elif kind == "too_many":
        idx = idx + [127] * (129 - len(idx))
        cells = cells + [cp[b][0][127]] * (129 - len(cells))
    got = outcome(hip.recover_cells_and_kzg_proofs, idx, cells)
    exp = outcome(oracle.recover_cells_and_kzg_proofs, idx, cells)
    assert got[0] == exp[0]
    if got[0] == "err":
        assert got == exp
    else:
        assert got[1][0] == exp[1][0] and got[1][1] == exp[1][1]


@settings(max_examples=30, **FUZZ)
@given(st.data())
def test_fuzz_verify_blob_kzg_proof_batch(hip, oracle, material, data):
    blobs, cm, pr, cp, bad_points = material
    rnd = random.Random(data.draw(st.integers(0, 2 ** 32)))
    n = data.draw(st.sampled_from([0, 1, 2, 3, 4, 5, 16, 40]))   # <= 3: host path; >= 4: GPU validation + ladders
    order = [rnd.randrange(4) for _ in range(n)]
    bl = [blobs[k] for k in order]
    cc = [cm[k] for k in order]
    pp = [pr[k] for k in order]
    for _ in range(data.draw(st.integers(0, 2))):
        if n == 0:
            break
        at = rnd.randrange(n)
        kind = data.draw(st.sampled_from(["proof_other", "proof_bad", "commit_bad", "commit_other", "blob_noncanonical", "blob_value"]))
        if kind == "proof_other":
            pp[at] = pr[(order[at] + 1) % 4]
        elif kind == "proof_bad":
            pp[at] = rnd.choice(bad_points)
        elif kind == "commit_bad":
            cc[at] = rnd.choice(bad_points)
        elif kind == "commit_other":
            cc[at] = cm[(order[at] + 2) % 4]
        elif kind == "blob_noncanonical":
            b = bytearray(bl[at])
            pos = rnd.randrange(4096)
            b[32 * pos:32 * pos + 32] = _noncanonical(rnd)
            bl[at] = bytes(b)
        else:
            b = bytearray(bl[at])
            b[32 * rnd.randrange(4096) + 31] ^= 1
            bl[at] = bytes(b)
    got = outcome(hip.verify_blob_kzg_proof_batch, bl, cc, pp)
    exp = outcome(oracle.verify_blob_kzg_proof_batch, bl, cc, pp)
    assert got == exp
    if n >= 1:
        got1 = outcome(hip.verify_blob_kzg_proof, bl[0], cc[0], pp[0])
        assert got1 == outcome(oracle.verify_blob_kzg_proof, bl[0], cc[0], pp[0])


@settings(max_examples=20, **FUZZ)
@given(st.data())
def test_fuzz_compute_kzg_proof_and_commitment(hip, oracle, material, data):
    blobs = material[0]
    rnd = random.Random(data.draw(st.integers(0, 2 ** 32)))
    # a blob from a handful of values (long runs, zeros, r - 1), optionally with one non-canonical element
    vals = [0, 1, R - 1, rnd.randrange(R), rnd.randrange(2 ** 64)]
    shape = data.draw(st.sampled_from(["random_blob", "few_values", "all_zero", "one_hot"]))
    if shape == "random_blob":
        blob = bytearray(blobs[rnd.randrange(4)])
    elif shape == "few_values":
        blob = bytearray(b"".join(vals[(j * 7 + j // 64) % len(vals)].to_bytes(32, "big") for j in range(4096)))
    elif shape == "all_zero":
        blob = bytearray(131072)
    else:
        blob = bytearray(131072)
        pos = rnd.randrange(4096)
        blob[32 * pos:32 * pos + 32] = rnd.randrange(R).to_bytes(32, "big")
    if data.draw(st.booleans()) and data.draw(st.booleans()):
        pos = rnd.randrange(4096)
        blob[32 * pos:32 * pos + 32] = _noncanonical(rnd)
    blob = bytes(blob)
    zk = data.draw(st.sampled_from(["zero", "one", "r_minus_1", "r", "max", "domain", "domain_brp", "random"]))
    z = {"zero": 0, "one": 1, "r_minus_1": R - 1, "r": R, "max": 2 ** 256 - 1,
         "domain": pow(W8192, 2 * rnd.randrange(4096), R),          # a 4096-th root of unity: the special case of eip4844.c:458-481
         "domain_brp": pow(W8192, 2 * int(format(rnd.randrange(4096), "012b")[::-1], 2), R),
         "random": rnd.randrange(R)}[zk].to_bytes(32, "big")
    assert outcome(hip.blob_to_kzg_commitment, blob) == outcome(oracle.blob_to_kzg_commitment, blob)
    assert outcome(hip.compute_kzg_proof, blob, z) == outcome(oracle.compute_kzg_proof, blob, z)
    c = outcome(oracle.blob_to_kzg_commitment, blob)
    if c[0] == "ok":
        assert outcome(hip.compute_blob_kzg_proof, blob, c[1]) == outcome(oracle.compute_blob_kzg_proof, blob, c[1])


@settings(max_examples=12, **FUZZ)
@given(st.data())
def test_fuzz_batch_entry_points_against_single_calls(hip, material, data):
    """The additive batch forms over sizes that straddle their chunk schedules (64 / 192 / 256-blob staging chunks of
    the commitment batch, the 64-blob latency path and 256-blob sub-chunks of cells+proofs), with non-canonical blobs
    at generated positions: per-blob status and outputs must equal the one-blob calls'."""
    import ctypes as C
    blobs = material[0]
    rnd = random.Random(data.draw(st.integers(0, 2 ** 32)))
    n = data.draw(st.sampled_from([1, 2, 63, 64, 65, 191, 256, 257, 320, 511, 513, 1025]))
    order = [rnd.randrange(4) for _ in range(n)]
    bad_at = sorted(set(rnd.randrange(n) for _ in range(data.draw(st.integers(0, 3)))))
    raw = bytearray(b"".join(blobs[k] for k in order))
    for at in bad_at:
        pos = at * 131072 + 32 * rnd.randrange(4096)
        raw[pos:pos + 32] = _noncanonical(rnd)
    raw = bytes(raw)
    single_c = [hip.blob_to_kzg_commitment(b) for b in blobs]
    out = C.create_string_buffer(48 * n)
    status = C.create_string_buffer(n)
    f = hip.lib.ckzg_hip_blob_to_kzg_commitment_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    rc = f(out, status, raw, n, C.addressof(hip.s))
    assert rc == (1 if bad_at else 0)
    assert [i for i, v in enumerate(status.raw) if v] == bad_at
    for i in range(n):
        if i not in bad_at:
            assert out.raw[48 * i:48 * i + 48] == single_c[order[i]], i
    if n <= 320:   # cells + proofs: 268 KB of output per blob
        g = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
        g.restype = C.c_int
        g.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
        proofs = C.create_string_buffer(n * 6144)
        want_cells = data.draw(st.booleans())
        cells = C.create_string_buffer(n * 262144) if want_cells else None
        rc = g(cells, proofs, status, raw, n, C.addressof(hip.s))
        assert rc == (1 if bad_at else 0)
        assert [i for i, v in enumerate(status.raw) if v] == bad_at
        cp = material[3]
        praw = proofs.raw
        craw = cells.raw if want_cells else None
        for i in range(n):
            if i not in bad_at:
                assert praw[6144 * i:6144 * (i + 1)] == b"".join(cp[order[i]][1]), i
                if want_cells and i % 7 == 0:
                    assert craw[262144 * i:262144 * (i + 1)] == b"".join(cp[order[i]][0]), i
