// fr_inv.hpp -- inversion in Fr by the same safegcd divstep iteration as fp28_inv.hpp, on 9 signed
// 30-bit limbs (255-bit modulus).  The barycentric evaluation inverts one product per lane; the
// Fermat ladder there (255 squarings + ~127 products) was four fifths of the kernel.  Bound: 738
// divsteps suffice for 255-bit inputs, i.e. at most 25 batches of 30; |d|, |e| grow by at most r per
// batch and 9 x 30 bits hold 2^14 r.
#pragma once
#include "fp28_inv.hpp"

namespace ckzg {

HD void fr_update_fg30(int32_t *f, int32_t *g, const DivstepMatrix &t) {
    const int64_t M = (1 << 30) - 1;
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = limb32((int32_t)(cf & M));
        g[i - 1] = limb32((int32_t)(cg & M));
        cf >>= 30;
        cg >>= 30;
    }
    f[8] = limb32((int32_t)cf);
    g[8] = limb32((int32_t)cg);
}

HD void fr_update_de30(int32_t *d, int32_t *e, const DivstepMatrix &t) {
    const int64_t M = (1 << 30) - 1;
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    const int32_t md = (int32_t)(((0u - (uint32_t)cd) * (uint32_t)FR30_RINV) & (uint32_t)M);
    const int32_t me = (int32_t)(((0u - (uint32_t)ce) * (uint32_t)FR30_RINV) & (uint32_t)M);
    cd += (int64_t)FR30_R[0] * md;
    ce += (int64_t)FR30_R[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)FR30_R[i] * md;
        ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)FR30_R[i] * me;
        d[i - 1] = limb32((int32_t)(cd & M));
        e[i - 1] = limb32((int32_t)(ce & M));
        cd >>= 30;
        ce >>= 30;
    }
    d[8] = limb32((int32_t)cd);
    e[8] = limb32((int32_t)ce);
}

// 1/a for a in Montgomery form (radix 2^256), result in Montgomery form; 0 for a == 0
HDNI inline Fr fr_inv_safegcd(const Fr &a) {
    if (a.is_zero()) return Fr::zero();
    int32_t f[9], g[9], d[9], e[9];
    for (int i = 0; i < 9; i++) {
        int bit = 30 * i, j = bit >> 5, sh = bit & 31;
        uint32_t v = j < 8 ? a.l[j] >> sh : 0u;
        if (sh > 2 && j + 1 < 8) v |= a.l[j + 1] << (32 - sh);
        g[i] = (int32_t)(v & 0x3fffffffu);
        f[i] = FR30_R[i];
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t eta = -1;
    for (int it = 0; it < 26; it++) {
        DivstepMatrix t;
        eta = divsteps30(eta, (uint32_t)f[0], (uint32_t)g[0], t);
        fr_update_de30(d, e, t);
        fr_update_fg30(f, g, t);
        int32_t nz = 0;
        for (int i = 0; i < 9; i++) nz |= g[i];
        if (nz == 0) break;
    }
    // f = +-1; result = f * d, made positive by adding 32r (|d| < 27r)
    const bool negate = f[8] < 0;
    int64_t c = 0;
    uint32_t w[9];
    for (int i = 0; i < 9; i++) {
        c += (int64_t)FR30_32R[i] + (negate ? -(int64_t)d[i] : (int64_t)d[i]);
        w[i] = (uint32_t)(c & 0x3fffffff);
        c >>= 30;
    }
    w[8] += (uint32_t)(c << 30);
    // 9 x 30 -> 32-bit words: y = lo + hi * 2^256 with hi < 2^5 (y < 59r < 2^261)
    uint32_t raw[9];
    for (int k = 0; k < 9; k++) {
        int bit = 32 * k, i = bit / 30, sh = bit - 30 * i;
        uint64_t v = (uint64_t)w[i] >> sh;
        if (i + 1 < 9) v |= (uint64_t)w[i + 1] << (30 - sh);
        raw[k] = (uint32_t)v;
    }
    // y = 1/(a R) as an integer; the Montgomery form of 1/a is y R^2 = mul(lo, R^3) + mul(hi R, R^3)
    Fr lo, r3;
    for (int k = 0; k < 8; k++) {
        lo.l[k] = raw[k];
        r3.l[k] = FR_R3[k];
    }
    Fr res = mul(lo, r3);
    if (raw[8]) res = add(res, mul(fr_from_u64(raw[8]), r3));
    return res;
}

}  // namespace ckzg
