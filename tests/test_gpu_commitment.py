"""GPU parity for blob_to_kzg_commitment (src/eip4844/eip4844.c:264-280) through the C-ABI:
golden vectors (tests/golden, from /root/reference/tests/blob_to_kzg_commitment), seeded random
blobs against the CPU oracle, batch == one-by-one, and a linearity property at full batch size."""
import ctypes as C
import hashlib
import os

import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def rand_blob(seed, i):
    # field element j = 0x00 || SHA256(seed|i|j)[0:31]  (canonical by construction)
    out = bytearray()
    for j in range(4096):
        h = hashlib.sha256(b"%d|%d|%d" % (seed, i, j)).digest()
        out += b"\x00" + h[:31]
    return bytes(out)


@pytest.mark.parametrize("name", G.case_names("blob_to_kzg_commitment"))
def test_golden(hip, name):
    got, exp = G.run_case(hip, "blob_to_kzg_commitment", name)
    assert got == exp


def test_random_vs_oracle(hip, oracle):
    for i in range(3):
        b = rand_blob(0xC4B64844, i)
        assert hip.blob_to_kzg_commitment(b) == oracle.blob_to_kzg_commitment(b)


def test_full_range_scalars_vs_oracle(hip, oracle):
    # 255-bit scalars, including r-1, 0, 1 and values with every digit pattern
    vals = [0, 1, R - 1, R - 2, 2 ** 254, (2 ** 255 - 19) % R, (R - 1) // 2, 0x8000800080008000 << 128]
    blob = bytearray()
    for j in range(4096):
        v = vals[j % len(vals)] if j < 64 else int.from_bytes(hashlib.sha256(b"x%d" % j).digest(), "big") % R
        blob += v.to_bytes(32, "big")
    blob = bytes(blob)
    assert hip.blob_to_kzg_commitment(blob) == oracle.blob_to_kzg_commitment(blob)


def _batch(hip, blobs):
    n = len(blobs)
    out = C.create_string_buffer(48 * n)
    status = C.create_string_buffer(n)
    f = hip.lib.ckzg_hip_blob_to_kzg_commitment_batch
    f.restype = C.c_int
    ret = f(out, status, b"".join(blobs), C.c_uint64(n), hip.sp)
    return ret, [out.raw[48 * i:48 * i + 48] for i in range(n)], list(status.raw)


def test_batch_matches_single_and_flags_bad_blobs(hip):
    blobs = [rand_blob(7, i) for i in range(5)]
    bad = bytearray(blobs[2])
    bad[32 * 2111:32 * 2112] = b"\xff" * 32
    blobs[2] = bytes(bad)
    ret, outs, status = _batch(hip, blobs)
    assert ret == 1  # C_KZG_BADARGS, as the one-blob call would for blob 2
    assert status == [0, 0, 1, 0, 0]
    for i in (0, 1, 3, 4):
        assert outs[i] == hip.blob_to_kzg_commitment(blobs[i])


def test_batch_linearity_property(hip, oracle):
    # commit is linear: C(a) + C(b) == C(a+b); checked with the oracle's group law on a batch large
    # enough to exercise the multi-workgroup path
    n = 64
    blobs = [rand_blob(11, i) for i in range(n)]
    sums = []
    for i in range(0, n, 2):
        s = bytearray()
        for j in range(4096):
            a = int.from_bytes(blobs[i][32 * j:32 * j + 32], "big")
            b = int.from_bytes(blobs[i + 1][32 * j:32 * j + 32], "big")
            s += ((a + b) % R).to_bytes(32, "big")
        sums.append(bytes(s))
    ret, outs, status = _batch(hip, blobs + sums)
    assert ret == 0 and not any(status)
    o = oracle.lib
    for k in range(n // 2):
        pa, pb, ps = (C.create_string_buffer(144) for _ in range(3))
        aa = C.create_string_buffer(96)
        for src, dst in ((outs[2 * k], pa), (outs[2 * k + 1], pb), (outs[n + k], ps)):
            assert o.og1_uncompress(aa, src) == 0
            o.og1_from_affine(dst, aa)
        o.og1_add(pa, pa, pb)
        o.og1_equal.restype = C.c_bool
        assert o.og1_equal(pa, ps)
