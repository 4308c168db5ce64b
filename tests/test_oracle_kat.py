"""Known answers embedded in the reference's own unit tests (/root/reference/src/test/tests.c),
restated as data and checked against the oracle on the CPU."""
import ctypes as C

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def _fr(o, v):
    b = C.create_string_buffer(32)
    o.lib.ofr_from_u64(b, C.c_uint64(v))
    return b


def test_fr_div_kat(oracle):
    # tests.c:217-232: 2345 / 54321
    o = oracle
    q = C.create_string_buffer(32)
    o.lib.ofr_div(q, _fr(o, 2345), _fr(o, 54321))
    out = C.create_string_buffer(32)
    o.lib.ofr_to_bytes(out, q)
    assert out.raw.hex() == "264d23155705ca938a1f22117681ea9759f348cb177a07ffe0813de67e85c684"
    assert int(out.raw.hex(), 16) == 2345 * pow(54321, -1, R) % R


def test_roots_of_unity_kat(oracle):
    # tests.c:1621-1641: the 2^13-th root is 7^((r-1)/8192); roots_of_unity[0] = roots_of_unity[8192] = 1
    s = oracle.s
    arr = (C.c_uint8 * (8193 * 32)).from_address(s.roots_of_unity)
    out = C.create_string_buffer(32)
    oracle.lib.ofr_to_bytes(out, bytes(arr[32:64]))
    assert int(out.raw.hex(), 16) == pow(7, (R - 1) // 8192, R)
    oracle.lib.ofr_to_bytes(out, bytes(arr[8192 * 32:8193 * 32]))
    assert int(out.raw.hex(), 16) == 1


def test_commitment_kats(oracle):
    # tests.c:477-497: the all-zero blob commits to the point at infinity
    assert oracle.blob_to_kzg_commitment(bytes(131072)).hex() == "c0" + "00" * 47
    # tests.c:499-530: blob {fe0 = 14629a..55ad, rest 0}
    fe = bytes.fromhex("14629a3a39f7b854e6aa49aa2edb450267eac2c14bb2d4f97a0b81a3f57055ad")
    got = oracle.blob_to_kzg_commitment(fe + bytes(131072 - 32)).hex()
    assert got.startswith("91a5e1c1") and got.endswith("e7ac")


def test_pippenger_equals_naive(oracle):
    # tests.c:929-946 (property): g1_lincomb_fast == g1_lincomb_naive on 128 random points
    import random
    o = oracle.lib
    rnd = random.Random(3)
    n = 128
    pts = C.create_string_buffer(144 * n)
    sc = C.create_string_buffer(32 * n)
    g = (C.c_uint8 * 144).in_dll(o, "OG1_GENERATOR")
    for i in range(n):
        k = rnd.randrange(R)
        kk = (C.c_uint64 * 4)(*[(k >> (64 * j)) & (2 ** 64 - 1) for j in range(4)])
        p = C.create_string_buffer(144)
        o.og1_mul_raw(p, g, kk, 255)
        pts[144 * i:144 * (i + 1)] = p.raw
        v = rnd.randrange(R).to_bytes(32, "big")
        f = C.create_string_buffer(32)
        o.ofr_from_bytes_reduce(f, v)
        sc[32 * i:32 * (i + 1)] = f.raw
    a, b = C.create_string_buffer(144), C.create_string_buffer(144)
    assert o.okzg_g1_lincomb_fast(a, pts, sc, C.c_size_t(n)) == 0
    o.okzg_g1_lincomb_naive(b, pts, sc, C.c_size_t(n))
    o.og1_equal.restype = C.c_bool
    assert o.og1_equal(a, b)
