// g1.hpp -- BLS12-381 G1 group law (y^2 = x^3 + 4 over Fp) for device and host.
//
// Three coordinate systems, each used where it is cheapest on a VALU-bound machine:
//   * G1Affine (x, y)           : table entries and bases; 96 B; infinity = (0, 0)
//   * G1XYZZ (X, Y, ZZ, ZZZ)    : MSM accumulators (mixed add 8M+2S, full add 12M+2S);
//                                 x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0
//   * G1Jac (X, Y, Z)           : doubling chains (2M+5S) and the layout of the reference's g1_t
//                                 (= blst_p1, src/common/ec.h:26); infinity <=> Z == 0
// All group laws below are complete (identity, P+P and P+(-P) handled), because inputs are
// untrusted: the reference's g1_add is blst_p1_add_or_double (src/common/ec.c:29).
#pragma once
#include "field.hpp"

namespace ckzg {

struct G1Affine {
    Fp x, y;
    HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    HD static G1Affine inf() { return {Fp::zero(), Fp::zero()}; }
};

struct G1Jac {
    Fp x, y, z;
    HD bool is_inf() const { return z.is_zero(); }
    HD static G1Jac inf() { return {Fp::zero(), Fp::zero(), Fp::zero()}; }
};

struct G1XYZZ {
    Fp x, y, zz, zzz;
    HD bool is_inf() const { return zz.is_zero(); }
    HD static G1XYZZ inf() { return {Fp::zero(), Fp::zero(), Fp::zero(), Fp::zero()}; }
};

HD G1XYZZ xyzz_from_affine(const G1Affine &p) {
    if (p.is_inf()) return G1XYZZ::inf();
    return {p.x, p.y, Fp::one(), Fp::one()};
}

HD G1XYZZ xyzz_from_jac(const G1Jac &p) {
    Fp zz = sqr(p.z);
    return {p.x, p.y, zz, mul(zz, p.z)};
}

// (X*ZZ^2, Y*ZZZ^2, ZZZ) is the same point in Jacobian coordinates
HD G1Jac jac_from_xyzz(const G1XYZZ &p) {
    Fp zz2 = sqr(p.zz), zzz2 = sqr(p.zzz);
    return {mul(p.x, zz2), mul(p.y, zzz2), p.zzz};
}

HD G1Jac jac_from_affine(const G1Affine &p) {
    if (p.is_inf()) return G1Jac::inf();
    return {p.x, p.y, Fp::one()};
}

HD G1Affine affine_neg(const G1Affine &p) { return {p.x, neg(p.y)}; }

// ---- XYZZ ---------------------------------------------------------------------------------

// dbl-2008-s-1 with a = 0 (6M+3S... written for clarity, not minimal)
HD G1XYZZ xyzz_dbl(const G1XYZZ &p) {
    Fp u = dbl(p.y);
    Fp v = sqr(u);
    Fp w = mul(u, v);
    Fp s = mul(p.x, v);
    Fp x2 = sqr(p.x);
    Fp m = add(dbl(x2), x2);
    Fp x3 = sub(sqr(m), dbl(s));
    Fp y3 = sub(mul(m, sub(s, x3)), mul(w, p.y));
    return {x3, y3, mul(v, p.zz), mul(w, p.zzz)};
}

// doubling of an affine point into XYZZ (mdbl-2008-s-1)
HD G1XYZZ xyzz_dbl_affine(const G1Affine &p) {
    Fp u = dbl(p.y);
    Fp v = sqr(u);
    Fp w = mul(u, v);
    Fp s = mul(p.x, v);
    Fp x2 = sqr(p.x);
    Fp m = add(dbl(x2), x2);
    Fp x3 = sub(sqr(m), dbl(s));
    Fp y3 = sub(mul(m, sub(s, x3)), mul(w, p.y));
    return {x3, y3, v, w};
}

// acc += p (madd-2008-s), complete.  p must not be infinity-encoded unless checked by caller;
// the check is included here for safety.
HD void xyzz_madd(G1XYZZ &acc, const G1Affine &p) {
    if (p.is_inf()) return;
    if (acc.is_inf()) {
        acc = {p.x, p.y, Fp::one(), Fp::one()};
        return;
    }
    Fp u2 = mul(p.x, acc.zz);
    Fp s2 = mul(p.y, acc.zzz);
    Fp pp = sub(u2, acc.x);
    Fp r = sub(s2, acc.y);
    if (pp.is_zero()) {
        if (r.is_zero()) {
            acc = xyzz_dbl_affine(p);
        } else {
            acc = G1XYZZ::inf();
        }
        return;
    }
    Fp p2 = sqr(pp);
    Fp p3 = mul(pp, p2);
    Fp q = mul(acc.x, p2);
    Fp x3 = sub(sub(sqr(r), p3), dbl(q));
    Fp y3 = sub(mul(r, sub(q, x3)), mul(acc.y, p3));
    acc.x = x3;
    acc.y = y3;
    acc.zz = mul(acc.zz, p2);
    acc.zzz = mul(acc.zzz, p3);
}

// a + b (add-2008-s), complete
HD G1XYZZ xyzz_add(const G1XYZZ &a, const G1XYZZ &b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    Fp u1 = mul(a.x, b.zz);
    Fp u2 = mul(b.x, a.zz);
    Fp s1 = mul(a.y, b.zzz);
    Fp s2 = mul(b.y, a.zzz);
    Fp pp = sub(u2, u1);
    Fp r = sub(s2, s1);
    if (pp.is_zero()) {
        if (r.is_zero()) return xyzz_dbl(a);
        return G1XYZZ::inf();
    }
    Fp p2 = sqr(pp);
    Fp p3 = mul(pp, p2);
    Fp q = mul(u1, p2);
    Fp x3 = sub(sub(sqr(r), p3), dbl(q));
    Fp y3 = sub(mul(r, sub(q, x3)), mul(s1, p3));
    return {x3, y3, mul(mul(a.zz, b.zz), p2), mul(mul(a.zzz, b.zzz), p3)};
}

HD G1XYZZ xyzz_neg(const G1XYZZ &a) { return {a.x, neg(a.y), a.zz, a.zzz}; }

// ---- Jacobian -----------------------------------------------------------------------------

// dbl-2009-l (a = 0): 2M + 5S
HD G1Jac jac_dbl(const G1Jac &p) {
    Fp a = sqr(p.x);
    Fp b = sqr(p.y);
    Fp c = sqr(b);
    Fp t = sub(sub(sqr(add(p.x, b)), a), c);
    Fp d = dbl(t);
    Fp e = add(dbl(a), a);
    Fp f = sqr(e);
    Fp x3 = sub(f, dbl(d));
    Fp c8 = dbl(dbl(dbl(c)));
    Fp y3 = sub(mul(e, sub(d, x3)), c8);
    Fp z3 = dbl(mul(p.y, p.z));
    return {x3, y3, z3};
}

// add-2007-bl, complete
HD G1Jac jac_add(const G1Jac &p, const G1Jac &q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    Fp z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    Fp u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    Fp s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    Fp h = sub(u2, u1);
    Fp rr = sub(s2, s1);
    if (h.is_zero()) {
        if (rr.is_zero()) return jac_dbl(p);
        return G1Jac::inf();
    }
    rr = dbl(rr);
    Fp i = sqr(dbl(h));
    Fp j = mul(h, i);
    Fp v = mul(u1, i);
    Fp x3 = sub(sub(sqr(rr), j), dbl(v));
    Fp y3 = sub(mul(rr, sub(v, x3)), dbl(mul(s1, j)));
    Fp z3 = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
    return {x3, y3, z3};
}

// madd-2007-bl, complete
HD G1Jac jac_madd(const G1Jac &p, const G1Affine &q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return {q.x, q.y, Fp::one()};
    Fp z1z1 = sqr(p.z);
    Fp u2 = mul(q.x, z1z1);
    Fp s2 = mul(mul(q.y, p.z), z1z1);
    Fp h = sub(u2, p.x);
    Fp rr = sub(s2, p.y);
    if (h.is_zero()) {
        if (rr.is_zero()) return jac_dbl(p);
        return G1Jac::inf();
    }
    rr = dbl(rr);
    Fp hh = sqr(h);
    Fp i = dbl(dbl(hh));
    Fp j = mul(h, i);
    Fp v = mul(p.x, i);
    Fp x3 = sub(sub(sqr(rr), j), dbl(v));
    Fp y3 = sub(mul(rr, sub(v, x3)), dbl(mul(p.y, j)));
    Fp z3 = sub(sub(sqr(add(p.z, h)), z1z1), hh);
    return {x3, y3, z3};
}

HD G1Jac jac_neg(const G1Jac &p) { return {p.x, neg(p.y), p.z}; }

// [k]P, k an nbits-bit little-endian u32 array (left-to-right double-and-add)
HDNI inline G1Jac jac_mul(const G1Jac &p, const uint32_t *k, int nbits) {
    G1Jac acc = G1Jac::inf();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = jac_add(acc, p);
    }
    return acc;
}

HDNI inline G1Affine jac_to_affine(const G1Jac &p) {
    if (p.is_inf()) return G1Affine::inf();
    Fp zi = fp_inv(p.z);
    Fp zi2 = sqr(zi);
    return {mul(p.x, zi2), mul(p.y, mul(zi2, zi))};
}

HDNI inline G1Affine xyzz_to_affine(const G1XYZZ &p) {
    if (p.is_inf()) return G1Affine::inf();
    Fp t = fp_inv(p.zzz);  // 1/z^3
    Fp u = mul(p.zz, t);   // 1/z
    return {mul(p.x, sqr(u)), mul(p.y, t)};
}

// ---- ZCash serialisation (src/common/bytes.c:42-44,81-95 via blst_p1_compress/uncompress) ----

HD bool fp_is_lex_largest(const Fp &a) {
    uint32_t raw[12], half[12];
    to_raw<FpParams>(raw, a);
#pragma unroll
    for (int i = 0; i < 12; i++) half[i] = FP_P_MINUS1_HALF[i];
    return !limbs_geq<12>(half, raw);  // raw > (p-1)/2
}

HD void fp_to_be48(uint8_t *out, const Fp &a) {
    uint32_t raw[12];
    to_raw<FpParams>(raw, a);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint32_t v = raw[11 - i];
        out[4 * i] = (uint8_t)(v >> 24);
        out[4 * i + 1] = (uint8_t)(v >> 16);
        out[4 * i + 2] = (uint8_t)(v >> 8);
        out[4 * i + 3] = (uint8_t)v;
    }
}

HD void g1_compress_affine(uint8_t *out, const G1Affine &p) {
    if (p.is_inf()) {
#pragma unroll
        for (int i = 0; i < 48; i++) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    fp_to_be48(out, p.x);
    out[0] |= 0x80;
    if (fp_is_lex_largest(p.y)) out[0] |= 0x20;
}

// 0 ok, 1 bad encoding, 2 not on curve.  No subgroup check.
HDNI inline int g1_uncompress(G1Affine &out, const uint8_t *in) {
    uint8_t b0 = in[0];
    if (!(b0 & 0x80)) return 1;
    if (b0 & 0x40) {
        if (b0 & 0x3f) return 1;
        for (int i = 1; i < 48; i++) {
            if (in[i]) return 1;
        }
        out = G1Affine::inf();
        return 0;
    }
    uint32_t raw[12], m[12];
    for (int i = 0; i < 12; i++) {
        const uint8_t *p = in + 4 * (11 - i);
        raw[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
    raw[11] &= 0x1fffffffu;
    mod_limbs<FpParams>(m);
    if (limbs_geq<12>(raw, m)) return 1;
    Fp x = from_raw<FpParams>(raw);
    Fp four = dbl(dbl(Fp::one()));
    Fp rhs = add(mul(sqr(x), x), four);
    uint32_t e[12];
    for (int i = 0; i < 12; i++) e[i] = FP_SQRT_EXP[i];
    Fp y = pow_limbs(rhs, e, 381);
    if (sqr(y) != rhs) return 2;
    bool want_largest = (b0 & 0x20) != 0;
    if (fp_is_lex_largest(y) != want_largest) y = neg(y);
    out = {x, y};
    return 0;
}

}  // namespace ckzg
