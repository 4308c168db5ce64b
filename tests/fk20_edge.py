"""Blobs whose FK20 fixed-base MSM scalars are chosen values (test infrastructure, pure Python integers).

compute_fk20_cell_proofs (src/eip7594/fk20.c:139-286) multiplies the setup columns by the size-128 FFT of the
circulant vectors c_i (circulant_coeffs_stride, fk20.c:55-78: c_i[0] = p[4095 - i], c_i[128 - k] = p[4095 - i - 64 k]
for k = 1..62, zero elsewhere), i = 0..63.  The product path folds the 1/128 of the later inverse G1 FFT into those
scalars (fk20.hip: fk20_run), so the scalar that meets table column f, row i is  s_i[f] = FFT(c_i)[f] / 128.

A c_i has 63 free entries, so 63 of its 128 frequencies can be prescribed: `blob_with_fk20_scalars` solves the
63 x 63 Vandermonde-like system once (it is the same for every row), fills the polynomial coefficients and returns
the blob (evaluation form over the bit-reversed 4096-domain, src/eip4844/eip4844.c:200-229).  The result is checked
by recomputing every prescribed scalar from the blob's own coefficients."""
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
W8192 = pow(7, (R - 1) // 8192, R)     # src/setup/setup.c:57-83 (primitive root 7)
W4096 = pow(W8192, 2, R)
W128 = pow(W8192, 64, R)
POS = [0] + [128 - k for k in range(1, 63)]   # support of a circulant vector, in the order k = 0, 1..62
FREQS = list(range(63))                        # the prescribed frequencies


def _brp(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2)


def _inverse_matrix(m):
    n = len(m)
    a = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(m)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % R)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], R - 2, R)
        a[c] = [x * inv % R for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % R for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


_MINV = None


def _solver():
    global _MINV
    if _MINV is None:
        _MINV = _inverse_matrix([[pow(W128, f * p, R) for p in POS] for f in FREQS])
    return _MINV


def _ntt(v, w):
    """out[k] = sum_j v[j] w^(jk), natural order in and out (fr_fft, src/eip7594/fft.c:70-115)."""
    n = len(v)
    if n == 1:
        return v[:]
    e, o = _ntt(v[0::2], w * w % R), _ntt(v[1::2], w * w % R)
    out = [0] * n
    t = 1
    for k in range(n // 2):
        x = t * o[k] % R
        out[k] = (e[k] + x) % R
        out[k + n // 2] = (e[k] - x) % R
        t = t * w % R
    return out


def fk20_scalars_of(poly):
    """s_i[f] for every row i < 64 and frequency f < 128, straight from the definition."""
    inv128 = pow(128, R - 2, R)
    out = []
    for i in range(64):
        c = [0] * 128
        c[0] = poly[4095 - i]
        for k in range(1, 63):
            c[128 - k] = poly[4095 - i - 64 * k]
        out.append([x * inv128 % R for x in _ntt(c, W128)])
    return out


def blob_with_fk20_scalars(target, low=None):
    """target(i, f) -> the scalar wanted at row i < 64, frequency f < 63.  low: the 64 coefficients
    that no circulant vector reads -- p[0..63] -- (default 1..64).  Returns (blob bytes, poly)."""
    minv = _solver()
    poly = [0] * 4096
    for j in range(64):
        poly[j] = (low[j] if low else j + 1) % R
    for i in range(64):
        t = [target(i, f) * 128 % R for f in FREQS]
        c = [sum(m * x for m, x in zip(row, t)) % R for row in minv]
        poly[4095 - i] = c[0]
        for k in range(1, 63):
            poly[4095 - i - 64 * k] = c[k]
    got = fk20_scalars_of(poly)
    for i in range(64):
        for f in FREQS:
            assert got[i][f] == target(i, f) % R, (i, f)
    ev = _ntt(poly, W4096)
    blob = b"".join(ev[_brp(i, 12)].to_bytes(32, "big") for i in range(4096))
    return blob, poly
