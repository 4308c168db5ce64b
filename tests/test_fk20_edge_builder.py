"""CPU checks of tests/fk20_edge.py (the blob builder behind the FK20 edge-digit GPU tests): the blob it returns
really carries the prescribed FK20 scalars (recomputed from the definition, src/eip7594/fk20.c:55-78,199-209), and
the edge values it is fed really are digit strings that sit on +-2^(c-1) in every window of the product's own
recoding (g1_28.hpp: glv_split_signed + recode_signed_128, through the host shim)."""
import ctypes as C

import fk20_edge as E
from conftest import SHIM_SO

R = E.R
LAMBDA = 0xd201000000010000 ** 2 - 1


def _digits(h, k, wbits):
    kk = (C.c_uint32 * 8)(*[(k >> (32 * i)) & 0xffffffff for i in range(8)])
    m1, m2 = (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
    n1, n2 = C.c_int(), C.c_int()
    h.hs_glv_split_signed(kk, m1, C.byref(n1), m2, C.byref(n2))
    nwh = 127 // wbits + 1
    out = []
    for m, neg in ((m1, n1.value), (m2, n2.value)):
        dg = (C.c_int16 * nwh)()
        h.hs_recode_signed_128(dg, m, neg, wbits, nwh)
        out.append([int(d) for d in dg])
    return out


def test_builder_prescribes_the_scalars():
    want = lambda i, f: (i * 7919 + f * 104729 + 12345) ** 5 % R   # noqa: E731
    blob, poly = E.blob_with_fk20_scalars(want)
    assert len(blob) == 131072
    s = E.fk20_scalars_of(poly)
    assert all(s[i][f] == want(i, f) for i in range(64) for f in range(63))
    # the blob is the evaluation form of poly over the bit-reversed domain: spot-check two positions by Horner
    for pos in (0, 1, 2049):
        x = pow(E.W4096, E._brp(pos, 12), R)
        acc = 0
        for c in reversed(poly):
            acc = (acc * x + c) % R
        assert acc == int.from_bytes(blob[32 * pos:32 * pos + 32], "big")


def test_edge_values_put_every_window_on_the_half():
    h = C.CDLL(SHIM_SO)
    for wbits in (10, 13, 15, 16):
        half, twin = 1 << (wbits - 1), 127 // wbits + 1
        m = sum(half << (wbits * w) for w in range(0, twin - 1, 2))
        for sa in (1, -1):
            for sb in (1, -1):
                d1, d2 = _digits(h, (sa * m + LAMBDA * sb * m) % R, wbits)
                for dg in (d1, d2):
                    # windows 0, 2, 4, ... carry +-half: the recoding's edge (a digit of exactly 2^(c-1))
                    assert all(abs(dg[w]) == half for w in range(0, twin - 1, 2)), (wbits, sa, sb, dg)
