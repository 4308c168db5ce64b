// host_pairing.hpp -- host-only part of the BLS12-381 stack the C-ABI needs outside the GPU hot
// path: the Fp2/Fp6/Fp12 tower, G2, the two-pairing product check and SHA-256.  These back
// pairings_verify (src/common/utils.c:172-196), g2_mul/g2_sub (src/eip4844/eip4844.c:114-130),
// blst_p2_uncompress at setup (src/setup/setup.c:467-477) and blst_sha256
// (src/eip4844/eip4844.c:176).  They run once or twice per API call and are not worth a kernel;
// the MSM/FFT work that dominates every call is in the .hip files.
//
// Tower: Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(1+u)), Fp12 = Fp6[w]/(w^2-v).
#pragma once
#include <cstring>
#include "g1.hpp"
#include "g1_28.hpp"

namespace ckzg {
namespace host {

struct Fp2 {
    Fp c0, c1;
    static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2 &o) const { return c0 == o.c0 && c1 == o.c1; }
};

inline Fp2 add(const Fp2 &a, const Fp2 &b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
inline Fp2 sub(const Fp2 &a, const Fp2 &b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
inline Fp2 neg(const Fp2 &a) { return {neg(a.c0), neg(a.c1)}; }
inline Fp2 dbl(const Fp2 &a) { return add(a, a); }
#if defined(__SIZEOF_INT128__)
// ---- double-width helpers for lazily reduced Fp2 products (host only, 64-bit limbs) ----
struct FpWide {
    uint64_t w[12];
};
inline void fp_load64(uint64_t x[6], const Fp &a) { __builtin_memcpy(x, a.l, 48); }
// 6 x 6 -> 12 limbs, no reduction: product scanning (Comba) with a three-word column accumulator
inline void fp_mul_wide(FpWide &r, const uint64_t a[6], const uint64_t b[6]) {
    typedef unsigned __int128 u128;
    uint64_t acc0 = 0, acc1 = 0, acc2 = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int i = (k < 6 ? 0 : k - 5); i <= (k < 6 ? k : 5); i++) {
            u128 p = (u128)a[i] * b[k - i];
            u128 s = (u128)acc0 + (uint64_t)p;
            acc0 = (uint64_t)s;
            s = (u128)acc1 + (uint64_t)(p >> 64) + (uint64_t)(s >> 64);
            acc1 = (uint64_t)s;
            acc2 += (uint64_t)(s >> 64);
        }
        r.w[k] = acc0;
        acc0 = acc1;
        acc1 = acc2;
        acc2 = 0;
    }
    r.w[11] = acc0;
}
inline void wide_add(FpWide &r, const FpWide &a, const FpWide &b) {
    typedef unsigned __int128 u128;
    u128 c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        c += (u128)a.w[i] + b.w[i];
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
}
inline void wide_sub(FpWide &r, const FpWide &a, const FpWide &b) {  // a >= b required
    typedef unsigned __int128 u128;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u128 d = (u128)a.w[i] - b.w[i] - br;
        r.w[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
}
inline const FpWide &fp_p_squared() {
    static const FpWide p2 = []() {
        uint64_t m[6];
        for (int i = 0; i < 6; i++) m[i] = (uint64_t)FP_P[2 * i] | ((uint64_t)FP_P[2 * i + 1] << 32);
        FpWide r;
        fp_mul_wide(r, m, m);
        return r;
    }();
    return p2;
}
// Montgomery reduction of t < p * 2^384: t / 2^384 mod p, fully reduced
inline Fp fp_redc(const FpWide &tin) {
    typedef unsigned __int128 u128;
    uint64_t t[13], m[6];
    __builtin_memcpy(t, tin.w, sizeof tin.w);
    t[12] = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) m[i] = (uint64_t)FP_P[2 * i] | ((uint64_t)FP_P[2 * i + 1] << 32);
    constexpr uint64_t m0 = (uint64_t)FP_P[0] | ((uint64_t)FP_P[1] << 32);
    constexpr uint64_t inv32 = (uint64_t)0 - (uint64_t)FP_NINV32;
    constexpr uint64_t ninv = (uint64_t)0 - inv32 * (2 - m0 * inv32);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint64_t q = t[i] * ninv;
        u128 c = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            c += (u128)q * m[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
#pragma unroll
        for (int k = i + 6; k < 13; k++) {
            c += t[k];
            t[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    uint64_t s[6], br = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)t[6 + i] - m[i] - br;
        s[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 6; i++) s[i] = br ? t[6 + i] : s[i];
    __builtin_memcpy(r.l, s, 48);
    return r;
}
inline Fp2 mul(const Fp2 &a, const Fp2 &b) {
    // Karatsuba with lazy reduction: three double-width products, two Montgomery reductions.
    // a0 + a1 < 2p needs no reduction (2p < 2^382), its product with b0 + b1 is < 4p^2 < p 2^384.
    typedef unsigned __int128 u128;
    uint64_t a0[6], a1[6], b0[6], b1[6], sa[6], sb[6];
    fp_load64(a0, a.c0);
    fp_load64(a1, a.c1);
    fp_load64(b0, b.c0);
    fp_load64(b1, b.c1);
    u128 ca = 0, cb = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ca += (u128)a0[i] + a1[i];
        sa[i] = (uint64_t)ca;
        ca >>= 64;
        cb += (u128)b0[i] + b1[i];
        sb[i] = (uint64_t)cb;
        cb >>= 64;
    }
    FpWide t0, t1, t2, u;
    fp_mul_wide(t0, a0, b0);
    fp_mul_wide(t1, a1, b1);
    fp_mul_wide(t2, sa, sb);
    wide_sub(t2, t2, t0);             // a0 b1 + a1 b0 + a1 b1
    wide_sub(t2, t2, t1);             // a0 b1 + a1 b0            (< 2p^2)
    wide_add(u, t0, fp_p_squared());  // a0 b0 + p^2
    wide_sub(u, u, t1);               // a0 b0 - a1 b1 + p^2      (in (0, 2p^2))
    return {fp_redc(u), fp_redc(t2)};
}
#else
inline Fp2 mul(const Fp2 &a, const Fp2 &b) {
    // Karatsuba: 3 base-field products
    Fp t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    Fp t2 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return {sub(t0, t1), sub(sub(t2, t0), t1)};
}
#endif
#if defined(__SIZEOF_INT128__)
inline Fp2 sqr(const Fp2 &a) {
    // (a0 + a1)(a0 - a1 + p) and 2 a0 a1 as double-width products: both < 4p^2 < p 2^384
    typedef unsigned __int128 u128;
    uint64_t a0[6], a1[6], s[6], d[6], m[6];
    fp_load64(a0, a.c0);
    fp_load64(a1, a.c1);
#pragma unroll
    for (int i = 0; i < 6; i++) m[i] = (uint64_t)FP_P[2 * i] | ((uint64_t)FP_P[2 * i + 1] << 32);
    u128 c = 0;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        c += (u128)a0[i] + a1[i];
        s[i] = (uint64_t)c;
        c >>= 64;
    }
    c = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {  // a0 + p - a1 > 0
        c += (u128)a0[i] + m[i];
        uint64_t lo = (uint64_t)c;
        c >>= 64;
        u128 t = (u128)lo - a1[i] - br;
        d[i] = (uint64_t)t;
        br = (uint64_t)(t >> 64) & 1;
        // a borrow out of limb i is repaid from the carry chain's next limb via br
    }
    // the final carry and borrow cancel: a0 + p - a1 < 2p < 2^384
    FpWide t0, t1;
    fp_mul_wide(t0, s, d);
    fp_mul_wide(t1, a0, a1);
    wide_add(t1, t1, t1);
    return {fp_redc(t0), fp_redc(t1)};
}
#else
inline Fp2 sqr(const Fp2 &a) {
    Fp m = mul(a.c0, a.c1);
    return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(m)};
}
#endif
inline Fp2 mul_fp(const Fp2 &a, const Fp &k) { return {mul(a.c0, k), mul(a.c1, k)}; }
inline Fp2 mul_xi(const Fp2 &a) { return {sub(a.c0, a.c1), add(a.c0, a.c1)}; }  // * (1+u)
inline Fp2 inv(const Fp2 &a) {
    Fp n = fp_inv(add(sqr(a.c0), sqr(a.c1)));
    return {mul(a.c0, n), neg(mul(a.c1, n))};
}

inline bool fp_sqrt(Fp &out, const Fp &a) {
    uint32_t e[12];
    for (int i = 0; i < 12; i++) e[i] = FP_SQRT_EXP[i];
    Fp s = pow_limbs(a, e, 381);
    out = s;
    return sqr(s) == a;
}

// norm method; false if a is a non-residue in Fp2
inline bool fp2_sqrt(Fp2 &out, const Fp2 &a) {
    Fp2 cand;
    if (a.c1.is_zero()) {
        Fp s;
        if (fp_sqrt(s, a.c0)) {
            cand = {s, Fp::zero()};
        } else {
            if (!fp_sqrt(s, neg(a.c0))) return false;
            cand = {Fp::zero(), s};
        }
    } else {
        Fp s, x0;
        if (!fp_sqrt(s, add(sqr(a.c0), sqr(a.c1)))) return false;
        Fp half = fp_inv(dbl(Fp::one()));
        if (!fp_sqrt(x0, mul(add(a.c0, s), half))) {
            if (!fp_sqrt(x0, mul(sub(a.c0, s), half))) return false;
        }
        cand = {x0, mul(a.c1, fp_inv(dbl(x0)))};
    }
    if (!(sqr(cand) == a)) return false;
    out = cand;
    return true;
}

struct Fp6 {
    Fp2 c0, c1, c2;
};
inline Fp6 add(const Fp6 &a, const Fp6 &b) { return {add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2)}; }
inline Fp6 sub(const Fp6 &a, const Fp6 &b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2)}; }
inline Fp6 neg(const Fp6 &a) { return {neg(a.c0), neg(a.c1), neg(a.c2)}; }
inline Fp6 mul(const Fp6 &a, const Fp6 &b) {
    Fp2 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1), v2 = mul(a.c2, b.c2);
    // Toom/Karatsuba-style cross terms
    Fp2 t12 = sub(sub(mul(add(a.c1, a.c2), add(b.c1, b.c2)), v1), v2);  // a1b2 + a2b1
    Fp2 t01 = sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1);  // a0b1 + a1b0
    Fp2 t02 = sub(sub(mul(add(a.c0, a.c2), add(b.c0, b.c2)), v0), v2);  // a0b2 + a2b0
    return {add(v0, mul_xi(t12)), add(t01, mul_xi(v2)), add(t02, v1)};
}
inline Fp6 mul_v(const Fp6 &a) { return {mul_xi(a.c2), a.c0, a.c1}; }
inline Fp6 inv(const Fp6 &a) {
    Fp2 t0 = sub(sqr(a.c0), mul_xi(mul(a.c1, a.c2)));
    Fp2 t1 = sub(mul_xi(sqr(a.c2)), mul(a.c0, a.c1));
    Fp2 t2 = sub(sqr(a.c1), mul(a.c0, a.c2));
    Fp2 d = add(mul(a.c0, t0), mul_xi(add(mul(a.c2, t1), mul(a.c1, t2))));
    Fp2 di = inv(d);
    return {mul(t0, di), mul(t1, di), mul(t2, di)};
}

struct Fp12 {
    Fp6 c0, c1;
    static Fp12 one() {
        Fp12 r;
        std::memset(&r, 0, sizeof r);
        r.c0.c0.c0 = Fp::one();
        return r;
    }
    bool is_one() const {
        Fp12 o = one();
        return std::memcmp(this, &o, sizeof o) == 0;
    }
};
inline Fp12 mul(const Fp12 &a, const Fp12 &b) {
    Fp6 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
    Fp6 c1 = sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1);
    return {add(v0, mul_v(v1)), c1};
}
// complex squaring: 2 Fp6 products instead of 3
inline Fp12 sqr(const Fp12 &a) {
    Fp6 v0 = mul(a.c0, a.c1);
    Fp6 t = mul(add(a.c0, a.c1), add(a.c0, mul_v(a.c1)));
    return {sub(sub(t, v0), mul_v(v0)), add(v0, v0)};
}
// a * (b0 + b1 v): 5 Fp2 products
inline Fp6 mul_sparse01(const Fp6 &a, const Fp2 &b0, const Fp2 &b1) {
    Fp2 m0 = mul(a.c0, b0), m1 = mul(a.c1, b1);
    Fp2 cross = sub(sub(mul(add(a.c0, a.c1), add(b0, b1)), m0), m1);  // a0 b1 + a1 b0
    return {add(m0, mul_xi(mul(a.c2, b1))), cross, add(m1, mul(a.c2, b0))};
}
// a * (k v) for k in Fp
inline Fp6 mul_sparse1_fp(const Fp6 &a, const Fp &k) {
    return {mul_xi(mul_fp(a.c2, k)), mul_fp(a.c0, k), mul_fp(a.c1, k)};
}
inline Fp12 conj(const Fp12 &a) { return {a.c0, neg(a.c1)}; }
inline Fp12 inv(const Fp12 &a) {
    Fp6 d = inv(sub(mul(a.c0, a.c0), mul_v(mul(a.c1, a.c1))));
    return {mul(a.c0, d), neg(mul(a.c1, d))};
}

// ---- G2: y^2 = x^3 + 4(1+u), Jacobian; same layout as blst_p2 (the reference's g2_t) ----

struct G2Affine {
    Fp2 x, y;
    bool is_inf() const { return x.is_zero() && y.is_zero(); }
};
struct G2Jac {
    Fp2 x, y, z;
    bool is_inf() const { return z.is_zero(); }
    static G2Jac inf() { return {Fp2::zero(), Fp2::zero(), Fp2::zero()}; }
};

inline G2Jac g2_generator() {
    G2Jac g;
    for (int i = 0; i < 12; i++) {
        g.x.c0.l[i] = G2_GEN_X0[i];
        g.x.c1.l[i] = G2_GEN_X1[i];
        g.y.c0.l[i] = G2_GEN_Y0[i];
        g.y.c1.l[i] = G2_GEN_Y1[i];
    }
    g.z = Fp2::one();
    return g;
}

inline G1Jac g1_generator() {
    G1Jac g;
    for (int i = 0; i < 12; i++) {
        g.x.l[i] = G1_GEN_X[i];
        g.y.l[i] = G1_GEN_Y[i];
    }
    g.z = Fp::one();
    return g;
}

inline G2Jac g2_dbl(const G2Jac &p) {
    Fp2 a = sqr(p.x), b = sqr(p.y), c = sqr(b);
    Fp2 d = dbl(sub(sub(sqr(add(p.x, b)), a), c));
    Fp2 e = add(dbl(a), a);
    Fp2 x3 = sub(sqr(e), dbl(d));
    Fp2 y3 = sub(mul(e, sub(d, x3)), dbl(dbl(dbl(c))));
    return {x3, y3, dbl(mul(p.y, p.z))};
}

inline G2Jac g2_add(const G2Jac &p, const G2Jac &q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    Fp2 z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    Fp2 u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    Fp2 s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    Fp2 h = sub(u2, u1), rr = sub(s2, s1);
    if (h.is_zero()) return rr.is_zero() ? g2_dbl(p) : G2Jac::inf();
    rr = dbl(rr);
    Fp2 i = sqr(dbl(h));
    Fp2 j = mul(h, i), v = mul(u1, i);
    Fp2 x3 = sub(sub(sqr(rr), j), dbl(v));
    Fp2 y3 = sub(mul(rr, sub(v, x3)), dbl(mul(s1, j)));
    Fp2 z3 = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
    return {x3, y3, z3};
}

inline G2Jac g2_neg(const G2Jac &p) { return {p.x, neg(p.y), p.z}; }

inline G2Jac g2_mul(const G2Jac &p, const uint32_t *k, int nbits) {
    G2Jac acc = G2Jac::inf();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = g2_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = g2_add(acc, p);
    }
    return acc;
}

inline G2Affine g2_to_affine(const G2Jac &p) {
    if (p.is_inf()) return {Fp2::zero(), Fp2::zero()};
    Fp2 zi = inv(p.z);
    Fp2 zi2 = sqr(zi);
    return {mul(p.x, zi2), mul(p.y, mul(zi2, zi))};
}

inline bool fp2_is_lex_largest(const Fp2 &a) {
    return a.c1.is_zero() ? fp_is_lex_largest(a.c0) : fp_is_lex_largest(a.c1);
}

inline bool fp_from_be48(Fp &out, const uint8_t *in, bool mask_flags) {
    uint32_t raw[12], m[12];
    for (int i = 0; i < 12; i++) {
        const uint8_t *p = in + 4 * (11 - i);
        raw[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
    if (mask_flags) raw[11] &= 0x1fffffffu;
    mod_limbs<FpParams>(m);
    if (limbs_geq<12>(raw, m)) return false;
    out = from_raw<FpParams>(raw);
    return true;
}

// 96-byte ZCash encoding: x.c1 (with flag bits) || x.c0.  0 ok, 1 bad encoding, 2 not on curve
inline int g2_uncompress(G2Affine &out, const uint8_t *in) {
    uint8_t b0 = in[0];
    if (!(b0 & 0x80)) return 1;
    if (b0 & 0x40) {
        if (b0 & 0x3f) return 1;
        for (int i = 1; i < 96; i++) {
            if (in[i]) return 1;
        }
        out = {Fp2::zero(), Fp2::zero()};
        return 0;
    }
    Fp2 x, y;
    if (!fp_from_be48(x.c1, in, true)) return 1;
    if (!fp_from_be48(x.c0, in + 48, false)) return 1;
    Fp four = dbl(dbl(Fp::one()));
    Fp2 rhs = add(mul(sqr(x), x), Fp2{four, four});
    if (!fp2_sqrt(y, rhs)) return 2;
    if (fp2_is_lex_largest(y) != ((b0 & 0x20) != 0)) y = neg(y);
    out = {x, y};
    return 0;
}

// ---- pairing product check ----------------------------------------------------------------

// line through the twist point (xt, yt) with slope lam, evaluated at the G1 point p, scaled into
// Fp12 as (lam*xt - yt) + (-lam*xp) v + yp v*w  (the scale factor lies in Fp4 and dies in the
// final exponentiation)
inline Fp12 line_eval(const Fp2 &lam, const Fp2 &xt, const Fp2 &yt, const G1Affine &p) {
    Fp12 l;
    std::memset(&l, 0, sizeof l);
    l.c0.c0 = sub(mul(lam, xt), yt);
    l.c0.c1 = neg(mul_fp(lam, p.x));
    l.c1.c1.c0 = p.y;
    return l;
}

inline Fp12 miller_loop(const G2Affine &q, const G1Affine &p) {
    Fp12 f = Fp12::one();
    if (p.is_inf() || q.is_inf()) return f;
    Fp2 tx = q.x, ty = q.y;
    const uint64_t xabs = BLS_X_ABS;
    for (int i = 62; i >= 0; i--) {
        f = mul(f, f);
        Fp2 x2 = sqr(tx);
        Fp2 lam = mul(add(dbl(x2), x2), inv(dbl(ty)));
        f = mul(f, line_eval(lam, tx, ty, p));
        Fp2 x3 = sub(sub(sqr(lam), tx), tx);
        Fp2 y3 = sub(mul(lam, sub(tx, x3)), ty);
        tx = x3;
        ty = y3;
        if ((xabs >> i) & 1) {
            lam = mul(sub(q.y, ty), inv(sub(q.x, tx)));
            f = mul(f, line_eval(lam, tx, ty, p));
            x3 = sub(sub(sqr(lam), tx), q.x);
            y3 = sub(mul(lam, sub(tx, x3)), ty);
            tx = x3;
            ty = y3;
        }
    }
    return f;  // conjugation for x < 0 omitted: applied to both factors of the product or neither
}

// f^(p^k), k = 1..3: with f = sum a_i w^i (a_i in Fp2, w^6 = 1+u), the map is
// a_i -> conj^k(a_i) * (1+u)^(i (p^k - 1)/6).  In the (c0, c1) layout a_0,a_2,a_4 are c0's and
// a_1,a_3,a_5 are c1's coefficients.
inline Fp2 frob_gamma(int k, int i) {
    Fp2 g;
    for (int j = 0; j < 12; j++) {
        g.c0.l[j] = FROB_GAMMA[k - 1][i - 1][0][j];
        g.c1.l[j] = FROB_GAMMA[k - 1][i - 1][1][j];
    }
    return g;
}

inline Fp12 frobenius(const Fp12 &f, int k) {
    auto cj = [k](const Fp2 &a) { return (k & 1) ? Fp2{a.c0, neg(a.c1)} : a; };
    Fp12 r;
    r.c0.c0 = cj(f.c0.c0);
    r.c1.c0 = mul(cj(f.c1.c0), frob_gamma(k, 1));
    r.c0.c1 = mul(cj(f.c0.c1), frob_gamma(k, 2));
    r.c1.c1 = mul(cj(f.c1.c1), frob_gamma(k, 3));
    r.c0.c2 = mul(cj(f.c0.c2), frob_gamma(k, 4));
    r.c1.c2 = mul(cj(f.c1.c2), frob_gamma(k, 5));
    return r;
}

// Squaring in the cyclotomic subgroup (Granger-Scott): with Fp12 seen as three Fp4 = Fp2[y]/(y^2 - xi)
// components (z0,z1), (z2,z3), (z4,z5), only the three Fp4 squares are needed: 9 Fp2 products
// instead of 18.  Valid only for elements of norm 1 (after the easy part of the final exponentiation).
inline Fp12 cyclotomic_sqr(const Fp12 &f) {
    Fp2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
    auto fp4_sqr = [](Fp2 &r0, Fp2 &r1, const Fp2 &a, const Fp2 &b) {
        Fp2 ab = mul(a, b);
        r0 = sub(sub(mul(add(a, b), add(a, mul_xi(b))), ab), mul_xi(ab));  // a^2 + xi b^2
        r1 = dbl(ab);
    };
    Fp2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(t0, t1, z0, z1);
    fp4_sqr(t2, t3, z2, z3);
    fp4_sqr(t4, t5, z4, z5);
    auto three_minus_two = [](const Fp2 &t, const Fp2 &z) { Fp2 d = sub(t, z); return add(dbl(d), t); };  // 3t - 2z
    auto three_plus_two = [](const Fp2 &t, const Fp2 &z) { Fp2 d = add(t, z); return add(dbl(d), t); };    // 3t + 2z
    Fp12 r;
    r.c0.c0 = three_minus_two(t0, z0);
    r.c1.c1 = three_plus_two(t1, z1);
    r.c1.c0 = three_plus_two(mul_xi(t5), z2);
    r.c0.c2 = three_minus_two(t4, z3);
    r.c0.c1 = three_minus_two(t2, z4);
    r.c1.c2 = three_plus_two(t3, z5);
    return r;
}

// g^x for the (negative) BLS parameter x, g in the cyclotomic subgroup (inverse = conjugate)
inline Fp12 pow_x(const Fp12 &g) {
    const uint64_t xabs = BLS_X_ABS;
    Fp12 acc = g;
    for (int i = 62; i >= 0; i--) {
        acc = cyclotomic_sqr(acc);
        if ((xabs >> i) & 1) acc = mul(acc, g);
    }
    return conj(acc);
}

// f^((p^12-1)/r * 3).  Easy part (p^6-1)(p^2+1); hard part through
//   3 (p^4 - p^2 + 1)/r = l0 + l1 p + l2 p^2 + l3 p^3,
//   l3 = (x-1)^2, l2 = l3 x, l1 = l2 x - l3, l0 = l1 x + 3
// (identity asserted in tools/gen_constants.py).  The extra factor 3 is harmless for an
// "== 1" test: the result has order dividing r, and r is prime to 3.
inline Fp12 final_exp(const Fp12 &f) {
    Fp12 a = mul(conj(f), inv(f));    // f^(p^6-1): now unitary
    a = mul(frobenius(a, 2), a);      // ^(p^2+1): now in the cyclotomic subgroup
    Fp12 t = mul(pow_x(a), conj(a));  // a^(x-1)
    Fp12 y3 = mul(pow_x(t), conj(t)); // a^((x-1)^2)
    Fp12 y2 = pow_x(y3);
    Fp12 y1 = mul(pow_x(y2), conj(y3));
    Fp12 y0 = mul(pow_x(y1), mul(cyclotomic_sqr(a), a));
    return mul(mul(y0, frobenius(y1, 1)), mul(frobenius(y2, 2), frobenius(y3, 3)));
}

// ---- G1 helpers of the host-side (small-batch) verification path ----

// [|x|]P for the BLS parameter |x| = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16
inline G1Jac jac_mul_bls_x(const G1Jac &p) {
    G1Jac acc = p;
    for (int b = 62; b >= 0; b--) {
        acc = jac_dbl(acc);
        if (b == 62 || b == 60 || b == 57 || b == 48 || b == 16) acc = jac_add(acc, p);
    }
    return acc;
}

// P in G1  <=>  [x^2]P = (beta^2 X, -Y): the endomorphism test of g1_28.hpp (g1_28_in_subgroup, where
// its exactness is argued) on the host's Jacobian arithmetic; 126 doublings + 10 additions
inline bool g1_in_subgroup_host(const G1Affine &a) {
    if (a.is_inf()) return true;
    G1Jac q = jac_mul_bls_x(jac_mul_bls_x(jac_from_affine(a)));
    if (q.is_inf()) return false;
    Fp b2;
    for (int i = 0; i < 12; i++) b2.l[i] = FP_BETA_LAMBDA2_MONT[i];
    Fp z2 = sqr(q.z);
    if (q.x != mul(mul(a.x, b2), z2)) return false;
    return add(q.y, mul(a.y, mul(z2, q.z))).is_zero();
}

// [k]P for P in G1 through the GLV split (g1_28.hpp: glv_split): k = k1 + k2 lambda with 128-bit halves,
// phi(P) = (beta X, Y, Z); joint double-and-add over (P, phi(P), P + phi(P)): 128 doublings and ~96
// additions instead of 255 and ~128
inline G1Jac g1_mul_glv_host(const G1Jac &p, const uint32_t *k) {
    if (p.is_inf()) return p;
    uint32_t k1[4], k2[4];
    glv_split(k, k1, k2);
    Fp b1;
    for (int i = 0; i < 12; i++) b1.l[i] = FP_BETA_LAMBDA_MONT[i];
    const G1Jac q = {mul(p.x, b1), p.y, p.z};
    const G1Jac pq = jac_add(p, q);
    G1Jac acc = G1Jac::inf();
    for (int i = 127; i >= 0; i--) {
        acc = jac_dbl(acc);
        const uint32_t b = ((k1[i >> 5] >> (i & 31)) & 1u) | (((k2[i >> 5] >> (i & 31)) & 1u) << 1);
        if (b == 1) acc = jac_add(acc, p);
        else if (b == 2) acc = jac_add(acc, q);
        else if (b == 3) acc = jac_add(acc, pq);
    }
    return acc;
}

// ---- fixed-argument pairing: the G2 inputs of every verification equation are one of three
// setup constants ([1]_2, [s]_2, [s^64]_2), so the slope and intercept of each of the 68 line
// functions are computed once at load time; a pairing then needs no G2 arithmetic and no
// inversions, and the two Miller loops of a product check share their squarings. ----

constexpr int MILLER_STEPS = 68;  // 63 doublings + 5 additions for |x| = 0xd201000000010000

struct G2Prepared {
    Fp2 lam[MILLER_STEPS];  // slope of the line at each step
    Fp2 c[MILLER_STEPS];    // lam * x_T - y_T
    bool inf = true;
};

inline void g2_prepare(G2Prepared &out, const G2Affine &q) {
    out.inf = q.is_inf();
    if (out.inf) return;
    Fp2 tx = q.x, ty = q.y;
    const uint64_t xabs = BLS_X_ABS;
    int n = 0;
    for (int i = 62; i >= 0; i--) {
        Fp2 x2 = sqr(tx);
        Fp2 lam = mul(add(dbl(x2), x2), inv(dbl(ty)));
        out.lam[n] = lam;
        out.c[n++] = sub(mul(lam, tx), ty);
        Fp2 x3 = sub(sub(sqr(lam), tx), tx);
        Fp2 y3 = sub(mul(lam, sub(tx, x3)), ty);
        tx = x3;
        ty = y3;
        if ((xabs >> i) & 1) {
            lam = mul(sub(q.y, ty), inv(sub(q.x, tx)));
            out.lam[n] = lam;
            out.c[n++] = sub(mul(lam, tx), ty);
            x3 = sub(sub(sqr(lam), tx), q.x);
            y3 = sub(mul(lam, sub(tx, x3)), ty);
            tx = x3;
            ty = y3;
        }
    }
}

// f * (c + (-lam*xp) v + yp v w): the line value is sparse, multiply it out directly
inline Fp12 mul_by_prepared_line(const Fp12 &f, const Fp2 &lam, const Fp2 &c, const G1Affine &p) {
    // l = (A + B v) + (yp v) w with A = c, B = -lam xp:  f l = (f0 l0 + v f1 l1) + (f0 l1 + f1 l0) w,
    // Karatsuba over w: 2 sparse Fp6 products of 5 Fp2 products + one by an Fp multiple of v
    const Fp2 B = neg(mul_fp(lam, p.x));
    Fp6 t0 = mul_sparse01(f.c0, c, B);
    Fp6 t1 = mul_sparse1_fp(f.c1, p.y);
    Fp2 By = B;
    By.c0 = add(By.c0, p.y);  // l0 + l1 = A + (B + yp) v
    Fp6 t2 = mul_sparse01(add(f.c0, f.c1), c, By);
    return {add(t0, mul_v(t1)), sub(sub(t2, t0), t1)};
}

// The Miller-loop value of e(p1, Q1) * e(p2, Q2) (before the final exponentiation); an infinite argument makes
// its factor 1.  A caller whose second pair is known early can run that loop ahead of time
// (miller_product_prepared(p2, q2, inf, q2)) and multiply the two values: the same product.
inline Fp12 miller_product_prepared(const G1Affine &p1, const G2Prepared &q1, const G1Affine &p2,
                                    const G2Prepared &q2) {
    const bool use1 = !p1.is_inf() && !q1.inf, use2 = !p2.is_inf() && !q2.inf;
    Fp12 f = Fp12::one();
    const uint64_t xabs = BLS_X_ABS;
    int n = 0;
    for (int i = 62; i >= 0; i--) {
        f = sqr(f);
        if (use1) f = mul_by_prepared_line(f, q1.lam[n], q1.c[n], p1);
        if (use2) f = mul_by_prepared_line(f, q2.lam[n], q2.c[n], p2);
        n++;
        if ((xabs >> i) & 1) {
            if (use1) f = mul_by_prepared_line(f, q1.lam[n], q1.c[n], p1);
            if (use2) f = mul_by_prepared_line(f, q2.lam[n], q2.c[n], p2);
            n++;
        }
    }
    return f;
}

// e(p1, Q1) * e(p2, Q2) == 1 ?
inline bool pairing_product_is_one(const G1Affine &p1, const G2Prepared &q1, const G1Affine &p2,
                                   const G2Prepared &q2) {
    return final_exp(miller_product_prepared(p1, q1, p2, q2)).is_one();
}

// ---- [k]G1 for the generator from a table: 64 windows of 4 bits, entries j * 16^w * G (j = 1..15) with Z = 1,
// so a 255-bit multiple is at most 64 additions and no doubling (the generic GLV ladder: 128 + ~96).  Built on
// first use (~1 ms). ----
struct G1GenTable {
    G1Jac e[64][15];
};
inline const G1GenTable &g1_gen_table() {
    static const G1GenTable *tbl = []() {
        G1GenTable *t = new G1GenTable;
        G1Jac base = g1_generator();
        for (int w = 0; w < 64; w++) {
            G1Jac cur = base;
            for (int j = 0; j < 15; j++) {
                t->e[w][j] = cur;
                cur = jac_add(cur, base);
            }
            base = cur;                                             // 16 * base
        }
        // all 960 entries to Z = 1 with ONE inversion (Montgomery's trick); no entry is at infinity
        G1Jac *flat = &t->e[0][0];
        const int N = 64 * 15;
        Fp *prefix = new Fp[N];
        Fp acc = Fp::one();
        for (int i = 0; i < N; i++) {
            prefix[i] = acc;
            acc = mul(acc, flat[i].z);
        }
        Fp inv = fp_inv(acc);
        for (int i = N - 1; i >= 0; i--) {
            const Fp zi = mul(inv, prefix[i]);
            inv = mul(inv, flat[i].z);
            const Fp zi2 = sqr(zi);
            flat[i].x = mul(flat[i].x, zi2);
            flat[i].y = mul(flat[i].y, mul(zi2, zi));
            flat[i].z = Fp::one();
        }
        delete[] prefix;
        return t;
    }();
    return *tbl;
}
// k: canonical little-endian limbs of a scalar < r
inline G1Jac g1_gen_mul(const uint32_t *k) {
    const G1GenTable &t = g1_gen_table();
    G1Jac acc = G1Jac::inf();
    for (int w = 0; w < 64; w++) {
        const uint32_t d = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (d) acc = jac_add(acc, t.e[w][d - 1]);
    }
    return acc;
}

// e(a1, a2) == e(b1, b2)
inline bool pairings_verify(const G1Jac &a1, const G2Jac &a2, const G1Jac &b1, const G2Jac &b2) {
    G1Affine pa = jac_to_affine(jac_neg(a1)), pb = jac_to_affine(b1);
    G2Affine qa = g2_to_affine(a2), qb = g2_to_affine(b2);
    Fp12 f = mul(miller_loop(qa, pa), miller_loop(qb, pb));
    return final_exp(f).is_one();
}

// ---- SHA-256 ------------------------------------------------------------------------------

inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

inline void sha256_block(uint32_t h[8], const uint8_t *blk) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) {
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
               ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    }
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// x86 SHA extensions: four rounds per instruction pair, ~4x the portable loop.  Compiled only into
// the host pass and chosen at run time from CPUID (leaf 7, EBX bit 29).
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#define CKZG_HAVE_SHANI 1
}  // namespace host
}  // namespace ckzg
#include <cpuid.h>
#include <immintrin.h>
namespace ckzg {
namespace host {
inline bool cpu_has_sha_ni() {
    static const int has = []() {
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return 0;
        unsigned a1 = 0, b1 = 0, c1 = 0, d1 = 0;
        if (!__get_cpuid(1, &a1, &b1, &c1, &d1)) return 0;
        const bool sse41 = (c1 >> 19) & 1u, ssse3 = (c1 >> 9) & 1u;
        return (int)(((b >> 29) & 1u) && sse41 && ssse3);
    }();
    return has != 0;
}

__attribute__((target("sha,sse4.1,ssse3"))) inline void sha256_blocks_shani(uint32_t h[8], const uint8_t *p, size_t nblocks) {
    alignas(16) static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    // the instructions want the state as (A,B,E,F) and (C,D,G,H)
    __m128i t = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&h[0]), 0xB1);   // C D A B
    __m128i s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&h[4]), 0x1B);  // E F G H
    __m128i s0 = _mm_alignr_epi8(t, s1, 8);                                          // A B E F
    s1 = _mm_blend_epi16(s1, t, 0xF0);                                               // C D G H
    while (nblocks--) {
        const __m128i save0 = s0, save1 = s1;
        __m128i m[4];
        for (int i = 0; i < 16; i++) {
            if (i < 4) {
                m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * i)), bswap);
            } else {
                // W[4i..4i+3] = msg2(msg1(W[4i-16..], W[4i-12..]) + W[4i-7..4i-4], W[4i-4..4i-1])
                __m128i x = _mm_sha256msg1_epu32(m[i & 3], m[(i + 1) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(m[(i + 3) & 3], m[(i + 2) & 3], 4));
                m[i & 3] = _mm_sha256msg2_epu32(x, m[(i + 3) & 3]);
            }
            __m128i wk = _mm_add_epi32(m[i & 3], _mm_load_si128((const __m128i *)&K[4 * i]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
            s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(wk, 0x0E));
        }
        s0 = _mm_add_epi32(s0, save0);
        s1 = _mm_add_epi32(s1, save1);
        p += 64;
    }
    t = _mm_shuffle_epi32(s0, 0x1B);                 // F E B A
    s1 = _mm_shuffle_epi32(s1, 0xB1);                // D C H G
    _mm_storeu_si128((__m128i *)&h[0], _mm_blend_epi16(t, s1, 0xF0));  // A B C D (memory order)
    _mm_storeu_si128((__m128i *)&h[4], _mm_alignr_epi8(s1, t, 8));     // E F G H
}
#endif

inline void sha256_blocks(uint32_t h[8], const uint8_t *p, size_t nblocks) {
#ifdef CKZG_HAVE_SHANI
    if (cpu_has_sha_ni()) {
        sha256_blocks_shani(h, p, nblocks);
        return;
    }
#endif
    for (size_t i = 0; i < nblocks; i++) sha256_block(h, p + 64 * i);
}

// incremental interface so transcripts need not be copied into one buffer
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    void update(const uint8_t *p, size_t n) {
        total += n;
        if (fill) {
            size_t take = 64 - fill < n ? 64 - fill : n;
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { sha256_blocks(h, buf, 1); fill = 0; }
        }
        if (n >= 64) { size_t nb = n / 64; sha256_blocks(h, p, nb); p += 64 * nb; n -= 64 * nb; }
        if (n) { std::memcpy(buf, p, n); fill = n; }
    }
    void finish(uint8_t out[32]) {
        uint64_t bits = total * 8;
        uint8_t pad[72] = {0x80};
        size_t padlen = (fill < 56) ? 56 - fill : 120 - fill;
        uint8_t lenb[8];
        for (int i = 0; i < 8; i++) lenb[7 - i] = (uint8_t)(bits >> (8 * i));
        update(pad, padlen);
        update(lenb, 8);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
            out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
        }
    }
};

}  // namespace host
}  // namespace ckzg
