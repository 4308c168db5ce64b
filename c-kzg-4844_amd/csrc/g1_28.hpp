// g1_28.hpp -- XYZZ mixed addition on the 28-bit-limb field (fp28.hpp): the inner operation of
// the MSM accumulate kernel.  Same madd-2008-s formulas as g1.hpp::xyzz_madd; the bounds in the
// comments are what the F28<limb, value> types prove at compile time.
#pragma once
#include "fp28.hpp"
#include "fp28_inv.hpp"
#include "g1.hpp"

namespace ckzg {

struct XYZZ28 {
    F28<1, 10> x;
    F28<1, 6> y;
    F28<1, 2> zz, zzz;
};

// doubling of an affine point (mdbl-2008-s-1), only reached when acc == pt
HDNI inline void xyzz28_dbl_affine(XYZZ28 &acc, const F28<1, 1> &x2, const F28<4, 2> &y2) {
    auto u = add(y2, y2);                       // <8,4>
    auto un = norm(u);                          // <1,4>
    auto v = sqr(un);                           // <1,2>
    auto w = mul(un, v);                        // <1,2>
    auto s = mul(x2, v);                        // <1,2>
    auto x2s = sqr(x2);                         // <1,2>
    auto m = add(add(x2s, x2s), x2s);           // <3,6>
    auto mm = sqr(m);                           // <1,2>
    acc.x = norm(sub(mm, add(s, s)));           // sub(<1,2>,<2,4>) = <5,10> -> <1,10>
    auto d = sub(s, acc.x);                     // <4,18>
    auto m1 = mul(m, d);                        // 14*12+15 ok, 6*18 ok
    auto m2 = mul(w, y2);                       // <1,2>
    acc.y = norm(sub(m1, m2));                  // <4,6> -> <1,6>
    acc.zz = v;
    acc.zzz = w;
}

// acc += (x2, y2); `inf` is the accumulator's infinity flag (kept outside the coordinates because
// a lazily reduced ZZ cannot be tested for zero cheaply).  The point must not be infinity.
HD void xyzz28_madd(XYZZ28 &acc, bool &inf, const F28<1, 1> &x2, const F28<4, 2> &y2) {
    if (inf) {
        acc.x = widen<1, 10>(x2);
        acc.y = widen<1, 6>(norm(y2));
        acc.zz = widen<1, 2>(f28_one());
        acc.zzz = acc.zz;
        inf = false;
        return;
    }
    auto u2 = mul(x2, acc.zz);                  // <1,2>
    auto s2 = mul(y2, acc.zzz);                 // <1,2>
    auto p = sub(u2, acc.x);                    // <4,18>
    auto r = sub(s2, acc.y);                    // <4,10>
    auto pp = sqr(p);                           // 14*16+15 = 239 <= 255; 18*18 <= 2500
    if (is_zero(pp)) {
        // same x: either the same point (double) or its negative (result is infinity)
        auto rr = mul(r, f28_one());
        if (is_zero(rr)) {
            xyzz28_dbl_affine(acc, x2, y2);
        } else {
            inf = true;
        }
        return;
    }
    auto ppp = mul(p, pp);                      // <1,2>
    auto q = mul(acc.x, pp);                    // <1,2>
    auto rr = sqr(r);                           // <1,2>
    auto t1 = add(ppp, add(q, q));              // <3,6>
    acc.x = norm(sub(rr, t1));                  // <6,10> -> <1,10>
    auto d = sub(q, acc.x);                     // <4,18>
    auto m1 = mul(r, d);                        // 14*16+15 ok; 10*18 ok
    auto m2 = mul(acc.y, ppp);                  // <1,2>
    acc.y = norm(sub(m1, m2));                  // <4,6> -> <1,6>
    acc.zz = mul(acc.zz, pp);
    acc.zzz = mul(acc.zzz, ppp);
}

// The accumulate kernels' form of the mixed addition: one field reduction fewer per addition.
// The accumulator keeps W = +-Y (`yneg` says which).  With t = -+y2 chosen so that T = t*ZZZ has the
// opposite sign, Rw = T + W = -+R is an addition, and the new W3 = Rw*(Q - X3) + W*PPP is a SUM of two
// products (one shared Montgomery reduction, mul_add2) that equals -+Y3: the stored sign flips at
// every addition, which costs nothing because the table point's sign is chosen per lane anyway.
// (x2, y2) is the table entry, fully reduced; `dneg` asks for its negative.  Not infinity.
HD void xyzz28_madd_alt(XYZZ28 &acc, bool &inf, bool &yneg, const F28<1, 1> &x2, const F28<1, 1> &y2, bool dneg) {
    if (inf) yneg = false;
    // yneg == false: stored +Y, want t = -(+-y2); yneg == true: stored -Y, want t = +(+-y2)
    const F28<4, 2> t = cneg_reduced(y2, dneg != !yneg);
    if (inf) {
        acc.x = widen<1, 10>(x2);
        acc.y = widen<1, 6>(norm(t));           // = -(the point's y): stored sign is "negative"
        acc.zz = widen<1, 2>(f28_one());
        acc.zzz = acc.zz;
        inf = false;
        yneg = true;
        return;
    }
    auto u2 = mul(x2, acc.zz);                  // <1,2>
    auto tt = mul(t, acc.zzz);                  // <1,2>   -+S2
    auto p = sub(u2, acc.x);                    // <4,18>
    auto r = add(tt, acc.y);                    // <2,8>    -+R
    auto pp = sqr(p);                           // <1,2>
    if (is_zero(pp)) {
        // same x: either the same point (double) or its negative (result is infinity)
        if (is_zero(mul(r, f28_one()))) {
            xyzz28_dbl_affine(acc, x2, cneg_reduced(y2, dneg));
        } else {
            inf = true;
        }
        yneg = false;
        return;
    }
    auto ppp = mul(p, pp);                      // <1,2>
    auto q = mul(acc.x, pp);                    // <1,2>
    auto rr = sqr(r);                           // 15*4+15 ok; 64 ok
    auto t1 = add(ppp, add(q, q));              // <3,6>
    acc.x = norm(sub(rr, t1));                  // <6,10> -> <1,10>
    auto d = sub(q, acc.x);                     // <4,18>
    // limbs 14*(2*4 + 1*1)+15 = 141 ok; values 8*18 + 6*2 = 156 ok
    acc.y = widen<1, 6>(mul_add2(r, d, acc.y, ppp));
    acc.zz = mul(acc.zz, pp);
    acc.zzz = mul(acc.zzz, ppp);
    yneg = !yneg;
}

// back to the plain representation (acc.y = +Y) after a run of xyzz28_madd_alt
HD void xyzz28_fix_sign(XYZZ28 &acc, bool inf, bool &yneg) {
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    auto n = norm(sub(zero, acc.y));            // <4,8> -> <1,8>: 8p - y
    auto m = mul(n, f28_one());                 // back under the <1,6> bound of the struct
    const bool flip = yneg && !inf;
#pragma unroll
    for (int j = 0; j < 14; j++) acc.y.l[j] = flip ? m.l[j] : acc.y.l[j];
    yneg = false;
}

// general doubling (dbl-2008-s-1)
HD void xyzz28_dbl(XYZZ28 &a) {
    auto u = add(a.y, a.y);                     // <2,12>
    auto v = sqr(u);                            // 14*4+15 ok; 144 ok
    auto w = mul(u, v);
    auto s = mul(a.x, v);
    auto x2 = sqr(a.x);                         // 100 ok
    auto m = add(add(x2, x2), x2);              // <3,6>
    auto mm = sqr(m);
    auto x3 = norm(sub(mm, add(s, s)));         // <1,10>
    auto d = sub(s, x3);                        // <4,18>
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    auto yn = sub(zero, a.y);                   // <4,8>  = -y
    // m*d - w*y with one reduction: limbs 14*(3*4 + 1*4)+15 = 239 ok; values 6*18 + 2*8 ok
    auto y3 = widen<1, 6>(mul_add2(m, d, w, yn));
    a.zz = mul(v, a.zz);
    a.zzz = mul(w, a.zzz);
    a.x = x3;
    a.y = y3;
}

// a += b (add-2008-s), complete; ainf/binf are the infinity flags
HD void xyzz28_add(XYZZ28 &a, bool &ainf, const XYZZ28 &b, bool binf) {
    if (binf) return;
    if (ainf) {
        a = b;
        ainf = false;
        return;
    }
    auto u1 = mul(a.x, b.zz);
    auto u2 = mul(b.x, a.zz);
    auto s1 = mul(a.y, b.zzz);
    auto s2 = mul(b.y, a.zzz);
    auto p = sub(u2, u1);                       // <4,6>
    auto r = sub(s2, s1);                       // <4,6>
    auto pp = sqr(p);
    if (is_zero(pp)) {
        if (is_zero(mul(r, f28_one()))) {
            xyzz28_dbl(a);
        } else {
            ainf = true;
        }
        return;
    }
    auto ppp = mul(p, pp);
    auto q = mul(u1, pp);
    auto rr = sqr(r);
    auto x3 = norm(sub(rr, add(ppp, add(q, q))));   // <1,10>
    auto d = sub(q, x3);                            // <4,18>
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    auto s1n = sub(zero, s1);                       // <4,4>  = -s1
    // r*d - s1*ppp with one reduction: limbs 14*(1*4 + 4*1)+15 = 127 ok; values 6*18 + 4*2 ok
    auto y3 = widen<1, 6>(mul_add2(norm(r), d, s1n, ppp));
    a.zz = mul(mul(a.zz, b.zz), pp);
    a.zzz = mul(mul(a.zzz, b.zzz), ppp);
    a.x = x3;
    a.y = y3;
}

// into the 28-bit domain from fully reduced 2^384-domain coordinates
HD XYZZ28 xyzz28_from_xyzz(const G1XYZZ &p, bool &inf) {
    XYZZ28 r;
    inf = p.is_inf();
    r.x = widen<1, 10>(f28_from_fp(p.x));
    r.y = widen<1, 6>(f28_from_fp(p.y));
    r.zz = f28_from_fp(p.zz);
    r.zzz = f28_from_fp(p.zzz);
    return r;
}

// -a (one multiplication by -1: keeps the coordinate inside its value bound)
HD XYZZ28 xyzz28_neg(const XYZZ28 &a) {
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    auto minus_one = sub(zero, f28_one());      // <4,2>
    XYZZ28 r = a;
    r.y = widen<1, 6>(mul(a.y, minus_one));
    return r;
}

// [k]P for a 255-bit k (canonical little-endian u32 limbs), fixed 4-bit windows: every lane runs
// the same 4 doublings + 1 table addition per window, so lanes with different scalars stay in step.
HDNI inline void xyzz28_mul_w4(XYZZ28 &out, bool &out_inf, const XYZZ28 &p, bool p_inf, const uint32_t *k) {
    XYZZ28 tbl[15];
    uint32_t tinf = 0;  // bit i: (i+1)*P is infinity (possible for untrusted points of small order)
    XYZZ28 acc;
    bool inf = true;
    if (!p_inf) {
        tbl[0] = p;
        tbl[1] = p;
        xyzz28_dbl(tbl[1]);  // E(Fp) has odd order: doubling a finite point never gives infinity
        for (int i = 2; i < 15; i++) {
            tbl[i] = tbl[i - 1];
            bool ti = ((tinf >> (i - 1)) & 1u) != 0;
            xyzz28_add(tbl[i], ti, p, false);
            if (ti) tinf |= 1u << i;
        }
        for (int w = 63; w >= 0; w--) {
            if (!inf) {
                xyzz28_dbl(acc);
                xyzz28_dbl(acc);
                xyzz28_dbl(acc);
                xyzz28_dbl(acc);
            }
            uint32_t d = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
            if (d) xyzz28_add(acc, inf, tbl[d - 1], ((tinf >> (d - 1)) & 1u) != 0);
        }
    }
    out = acc;
    out_inf = inf;
}

// GLV: lambda = x^2 - 1 satisfies lambda^2 + lambda + 1 = r exactly, so k2 = k / lambda and
// k1 = k mod lambda (plain integer division, both < 2^128, both non-negative) give
// k = k1 + k2*lambda.  Binary long division on 32-bit limbs: setup-time work for the ~130 twiddles.
HDNI inline void glv_split(const uint32_t *k, uint32_t *k1, uint32_t *k2) {
    uint32_t rem[5] = {0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 255; i >= 0; i--) {
        for (int j = 4; j > 0; j--) rem[j] = (rem[j] << 1) | (rem[j - 1] >> 31);
        rem[0] = (rem[0] << 1) | ((k[i >> 5] >> (i & 31)) & 1u);
        uint32_t t[5], br = 0;
        for (int j = 0; j < 5; j++) {
            uint64_t d = (uint64_t)rem[j] - (j < 4 ? FR_LAMBDA[j] : 0u) - br;
            t[j] = (uint32_t)d;
            br = (uint32_t)(d >> 32) & 1u;
        }
        if (!br) {
            for (int j = 0; j < 5; j++) rem[j] = t[j];
            q[i >> 5] |= 1u << (i & 31);
        }
    }
    for (int j = 0; j < 4; j++) {
        k1[j] = rem[j];
        k2[j] = q[j];  // k < r = lambda^2 + lambda + 1  =>  k2 <= lambda + 1 < 2^128
    }
}

// Balanced GLV split for the fixed-base tables: k = s1*m1 + lambda * s2*m2 (mod r) with signs s1, s2 and
// magnitudes m1, m2 <= (lambda + 3)/2 < 0.68 * 2^127, so that a signed c-bit recoding of either half fits
// floor(127/c) + 1 windows without a carry out of the top one (msm.hip).  Straight-line code, one per
// blob field element: k2 = floor(k / lambda) by a Barrett product with G = floor(2^383 / lambda)
// (estimate is k2 or k2 - 1, fixed by one conditional subtraction), then the two centring steps
//   k1 > lambda/2      ->  (k1 - lambda, k2 + 1)              [same value]
//   k2 > (lambda+1)/2  ->  (k1 - 1, k2 - (lambda + 1))        [value - r, since lambda^2 + lambda + 1 = r]
// k: canonical (< r) little-endian u32 limbs.
HD void glv_split_signed(const uint32_t *k, uint32_t *m1, bool &neg1, uint32_t *m2, bool &neg2) {
    // high part of k * G: only limbs 11..15 of the 512-bit product are needed, but the carries into them
    // come from every column, so all 64 partial products are accumulated (column sums in 64 + 32 bits)
    uint32_t prod[16];
    {
        uint64_t lo = 0;
        uint32_t hi = 0;  // column accumulator: hi:lo
#pragma unroll
        for (int c = 0; c < 15; c++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int j = c - i;
                if (j < 0 || j > 7) continue;
                const uint64_t t = (uint64_t)k[i] * FR_LAMBDA_RECIP383[j];
                lo += t;
                hi += lo < t ? 1u : 0u;
            }
            prod[c] = (uint32_t)lo;
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
        prod[15] = (uint32_t)lo;
    }
    uint32_t q[5];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = (prod[11 + i] >> 31) | (prod[12 + i] << 1);
    q[4] = prod[15] >> 31;  // 0: k < 2^255 and G < 2^256
    // rem = k - q * lambda, low 160 bits (the true value is in [0, 2 lambda))
    uint32_t ql[5];
    {
        uint64_t lo = 0;
        uint32_t hi = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = c - i;
                if (j < 0 || j > 3) continue;
                const uint64_t t = (uint64_t)q[i] * FR_LAMBDA[j];
                lo += t;
                hi += lo < t ? 1u : 0u;
            }
            ql[c] = (uint32_t)lo;
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
    }
    uint32_t rem[5];
    {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint64_t d = (uint64_t)k[i] - ql[i] - br;
            rem[i] = (uint32_t)d;
            br = (uint32_t)(d >> 32) & 1u;
        }
    }
    auto geq_lambda = [](const uint32_t *v) {  // v (5 limbs) >= lambda
        if (v[4]) return true;
        for (int i = 3; i >= 0; i--) {
            if (v[i] != FR_LAMBDA[i]) return v[i] > FR_LAMBDA[i];
        }
        return true;
    };
    if (geq_lambda(rem)) {
        uint32_t br = 0, c = 1;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint64_t d = (uint64_t)rem[i] - (i < 4 ? FR_LAMBDA[i] : 0u) - br;
            rem[i] = (uint32_t)d;
            br = (uint32_t)(d >> 32) & 1u;
            const uint64_t a = (uint64_t)q[i] + c;
            q[i] = (uint32_t)a;
            c = (uint32_t)(a >> 32);
        }
    }
    // now k = rem + lambda * q with 0 <= rem < lambda, 0 <= q <= lambda + 1
    // centre k1: 2*rem > lambda ?
    uint32_t dbl[5];
#pragma unroll
    for (int i = 0; i < 5; i++) dbl[i] = (rem[i] << 1) | (i ? rem[i - 1] >> 31 : 0u);
    bool n1 = false;
    {
        bool gt = dbl[4] != 0;
        if (!gt) {
            gt = false;
            for (int i = 3; i >= 0; i--) {
                if (dbl[i] != FR_LAMBDA[i]) {
                    gt = dbl[i] > FR_LAMBDA[i];
                    break;
                }
            }
        }
        if (gt) {
            // rem <- lambda - rem (magnitude of the negative value), q <- q + 1
            uint32_t br = 0, c = 1;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const uint64_t d = (uint64_t)(i < 4 ? FR_LAMBDA[i] : 0u) - rem[i] - br;
                rem[i] = (uint32_t)d;
                br = (uint32_t)(d >> 32) & 1u;
                const uint64_t a = (uint64_t)q[i] + c;
                q[i] = (uint32_t)a;
                c = (uint32_t)(a >> 32);
            }
            n1 = true;
        }
    }
    // centre k2: 2*q > lambda + 1 ?   (lambda + 1 = x^2: limb 0 of lambda is 0xffffffff, so add with carry)
    uint32_t lp1[5];
    {
        uint32_t c = 1;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint64_t a = (uint64_t)(i < 4 ? FR_LAMBDA[i] : 0u) + c;
            lp1[i] = (uint32_t)a;
            c = (uint32_t)(a >> 32);
        }
    }
    bool n2 = false;
    {
        uint32_t d2[6];
#pragma unroll
        for (int i = 0; i < 5; i++) d2[i] = (q[i] << 1) | (i ? q[i - 1] >> 31 : 0u);
        d2[5] = q[4] >> 31;
        bool gt = d2[5] != 0;
        if (!gt) {
            for (int i = 4; i >= 0; i--) {
                if (d2[i] != lp1[i]) {
                    gt = d2[i] > lp1[i];
                    break;
                }
            }
        }
        if (gt) {
            // q <- (lambda + 1) - q (magnitude), k1 <- k1 - 1
            uint32_t br = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const uint64_t d = (uint64_t)lp1[i] - q[i] - br;
                q[i] = (uint32_t)d;
                br = (uint32_t)(d >> 32) & 1u;
            }
            n2 = true;
            const bool zero = (rem[0] | rem[1] | rem[2] | rem[3] | rem[4]) == 0;
            if (n1 || zero) {  // magnitude grows: -(m) - 1 = -(m + 1);  0 - 1 = -(1)
                uint32_t c = 1;
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const uint64_t a = (uint64_t)rem[i] + c;
                    rem[i] = (uint32_t)a;
                    c = (uint32_t)(a >> 32);
                }
                n1 = true;
            } else {
                uint32_t br2 = 1;
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const uint64_t d = (uint64_t)rem[i] - br2;
                    rem[i] = (uint32_t)d;
                    br2 = (uint32_t)(d >> 32) & 1u;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m1[i] = rem[i];
        m2[i] = q[i];
    }
    neg1 = n1;
    neg2 = n2;
}

// Signed base-2^wbits digits of the signed half-scalar (neg ? -m : m), m < 2^(wbits*nwh - 1), digit w
// written at dst[w * stride].  Stored digits lie in [-2^(wbits-1), 2^(wbits-1) - 1] (int16_t up to
// wbits = 16): a non-negative value rounds a window of exactly 2^(wbits-1) down to -2^(wbits-1) with a
// carry, a negative one keeps +2^(wbits-1) and stores its negation.
HD void recode_signed_128(int16_t *dst, size_t stride, const uint32_t *m, bool neg, int wbits, int nwh) {
    uint32_t s[5] = {m[0], m[1], m[2], m[3], 0u};
    const uint32_t mask = (1u << wbits) - 1u, half = 1u << (wbits - 1);
    uint32_t carry = 0;
    for (int w = 0; w < nwh; w++) {
        int d = (int)((s[0] & mask) + carry);
#pragma unroll
        for (int k = 0; k < 4; k++) s[k] = (s[k] >> wbits) | (s[k + 1] << (32 - wbits));
        s[4] >>= wbits;
        carry = 0;
        if (neg ? (uint32_t)d > half : (uint32_t)d >= half) {
            d -= (int)(mask + 1u);
            carry = 1;
        }
        dst[(size_t)w * stride] = (int16_t)(neg ? -d : d);
    }
}

// Width-4 non-adjacent form of a 128-bit k: digits in {0, +-1, +-3, +-5, +-7}, digit i has weight 2^i,
// at most one non-zero digit in any 4 consecutive positions (density 1/5).  out has GLV_NAF_LEN entries.
constexpr int GLV_NAF_LEN = 132;
HDNI inline void wnaf4_128(int8_t *out, const uint32_t *k) {
    uint32_t v[5] = {k[0], k[1], k[2], k[3], 0};
    for (int i = 0; i < GLV_NAF_LEN; i++) {
        int d = 0;
        if (v[0] & 1u) {
            d = (int)(v[0] & 15u);
            if (d >= 8) d -= 16;
            // v -= d
            if (d > 0) {
                uint64_t br = (uint64_t)d;
                for (int j = 0; j < 5 && br; j++) {
                    uint64_t t = (uint64_t)v[j] - br;
                    v[j] = (uint32_t)t;
                    br = (t >> 32) & 1u;
                }
            } else {
                uint64_t c = (uint64_t)(-d);
                for (int j = 0; j < 5 && c; j++) {
                    uint64_t t = (uint64_t)v[j] + c;
                    v[j] = (uint32_t)t;
                    c = t >> 32;
                }
            }
        }
        out[i] = (int8_t)d;
        for (int j = 0; j < 4; j++) v[j] = (v[j] >> 1) | (v[j + 1] << 31);
        v[4] >>= 1;
    }
}

// ---- Jacobian coordinates on the 28-bit-limb field, for doubling-heavy ladders ----
// A doubling costs 3M + 4S here against 6M + 3S in XYZZ (2,380 vs 3,059 multiply-adds); an addition
// with a table entry that caches Z^2 and Z^3 costs the same as the XYZZ addition.  So the scalar
// ladders (G1 FFT twiddles, subgroup test, variable-base sums) run in Jacobian form and convert at
// the ends: (X, Y, ZZ, ZZZ) -> (X*ZZ, Y*ZZZ, ZZ) is a valid Jacobian triple (Z = ZZ), and
// (X, Y, Z) -> (X, Y, Z^2, Z^3) the way back.
struct JAC28 {
    F28<1, 34> x, y;
    F28<2, 4> z;
};
// table entry: the point plus Z^2, Z^3; y may be a lazily negated value
struct JACT28 {
    F28<1, 34> x;
    F28<1, 64> y;
    F28<2, 4> z;
    F28<1, 2> zz, zzz;
};

HD JAC28 jac28_from_xyzz(const XYZZ28 &p) {
    JAC28 r;
    r.x = widen<1, 34>(mul(p.x, p.zz));
    r.y = widen<1, 34>(mul(p.y, p.zzz));
    r.z = widen<2, 4>(p.zz);
    return r;
}
HD XYZZ28 jac28_to_xyzz(const JAC28 &p) {
    XYZZ28 r;
    r.zz = sqr(p.z);
    r.zzz = mul(p.z, r.zz);
    r.x = widen<1, 10>(mul(p.x, f28_one()));   // back under the XYZZ28 bounds
    r.y = widen<1, 6>(mul(p.y, f28_one()));
    return r;
}
HD JACT28 jac28_table_entry(const JAC28 &p) {
    JACT28 t;
    t.x = p.x;
    t.y = widen<1, 64>(p.y);
    t.z = p.z;
    t.zz = sqr(p.z);
    t.zzz = mul(p.z, t.zz);
    return t;
}
HD JACT28 jact28_neg(const JACT28 &t) {
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    JACT28 r = t;
    // only ever applied to entries whose y is still under the <1,34> bound of a JAC28
    F28<1, 34> y;
#pragma unroll
    for (int j = 0; j < 14; j++) y.l[j] = t.y.l[j];
    r.y = norm(sub(zero, y));                   // <4,64> -> <1,64>
    return r;
}

// dbl-2009-l for a = 0 with D = 4 X Y^2 taken as a product (keeps the value bounds small)
HD void jac28_dbl(JAC28 &a) {
    auto A = sqr(a.x);                          // 34^2 = 1156 ok
    auto B = sqr(a.y);
    auto C = sqr(B);
    auto XB = mul(a.x, B);                      // 34*2 ok
    auto XB2 = add(XB, XB);
    auto D = add(XB2, XB2);                     // <4,8>   = 4 X Y^2
    auto E = add(add(A, A), A);                 // <3,6>   = 3 X^2
    auto F = sqr(E);                            // 15*9+15 ok
    auto X3 = norm(sub(F, add(D, D)));          // sub(<1,2>,<8,16>) = <11,34> -> <1,34>
    auto dx = norm(sub(D, X3));                 // <7,72> -> <1,72>
    auto C2 = add(C, C);
    auto C4 = add(C2, C2);
    auto C8 = add(C4, C4);                      // <8,16>
    auto Y3 = norm(sub(mul(E, dx), C8));        // mul: 14*3+15 ok, 6*72 ok; sub -> <11,34> -> <1,34>
    auto YZ = mul(a.y, a.z);                    // 14*2+15 ok, 34*4 ok
    a.x = X3;
    a.y = Y3;
    a.z = add(YZ, YZ);                          // <2,4>
}

// a += b (add-2007-bl with the second operand's Z^2, Z^3 cached), complete; b is finite
HD void jac28_add(JAC28 &a, bool &ainf, const JACT28 &b) {
    if (ainf) {
        a.x = b.x;
        a.y = widen<1, 34>(mul(b.y, f28_one()));
        a.z = b.z;
        ainf = false;
        return;
    }
    auto z1z1 = sqr(a.z);                       // 15*4+15 ok, 16 ok
    auto u1 = mul(a.x, b.zz);                   // 34*2 ok
    auto u2 = mul(b.x, z1z1);
    auto s1 = mul(a.y, b.zzz);
    auto s2 = mul(b.y, mul(a.z, z1z1));         // inner 14*2+15 ok, 4*2 ok; outer 64*2 ok
    auto h = sub(u2, u1);                       // <4,6>
    auto r = sub(s2, s1);                       // <4,6>
    auto hh = sqr(h);                           // 15*16+15 = 255 ok, 36 ok
    if (is_zero(hh)) {
        if (is_zero(mul(r, f28_one()))) {
            jac28_dbl(a);
        } else {
            ainf = true;
        }
        return;
    }
    auto hhh = mul(h, hh);
    auto v = mul(u1, hh);
    auto rr = sqr(r);
    auto x3 = norm(sub(rr, add(hhh, add(v, v))));   // <6,10> -> <1,10>
    auto dv = sub(v, x3);                           // <4,18>
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    auto s1n = sub(zero, s1);                       // <4,4>
    // r*dv - s1*hhh with one reduction: limbs 14*(1*4 + 4*1)+15 = 127 ok; values 6*18 + 4*2 ok
    auto y3 = mul_add2(norm(r), dv, s1n, hhh);
    auto z3 = mul(mul(a.z, b.z), h);                // 14*4+15 ok, 16 ok; 14*4+15 ok, 2*6 ok
    a.x = widen<1, 34>(x3);
    a.y = widen<1, 34>(y3);
    a.z = widen<2, 4>(z3);
}

// ---- the NAF ladder's "effectively affine" table (one-lane form of the G1 FFT twiddle multiplication) ----
// P, 3P, 5P, 7P are built with co-Z arithmetic (Meloni; Longa-Miri): a co-Z addition of two points that share Z
// returns the sum AND its first operand over the sum's new Z, and any other point over the old Z follows for two
// multiplications (X * dX^2, Y * dX^3, both factors are by-products of the addition).  With all four multiples
// over one Zc, the map (X, Y, Z) -> (X, Y, Z/Zc) is an isomorphism onto the curve y^2 = x^3 + 4 Zc^6, on which
// the table is AFFINE; doubling and addition for a = 0 never use the constant term, so the whole ladder runs
// there with mixed additions (3S + 6M + one two-product reduction, 10.5 products instead of 14 + 1 for phi)
// and the result comes home by one multiplication of Z with Zc.  Table: 37 + 6 products instead of 61.
struct JE28 {        // accumulator on the isomorphic curve (Jacobian)
    F28<1, 20> x, y;
    F28<2, 4> z;
};
struct EAT28 {       // table entry: x, beta*x (the phi image), y
    F28<1, 20> x, bx, y;
};

// dbl-2009-l for a = 0, as jac28_dbl, with the tighter subtraction constants the mixed addition's squarings need
HD void je28_dbl(JE28 &a) {
    auto A = sqr(a.x);                          // 20^2 = 400 ok
    auto B = sqr(a.y);
    auto C = sqr(B);
    auto XB = mul(a.x, B);
    auto XB2 = add(XB, XB);
    auto D = add(XB2, XB2);                     // <4,8>
    auto E = add(add(A, A), A);                 // <3,6>
    auto F = sqr(E);
    auto X3 = norm(sub_k<17>(F, add(D, D)));    // <1,2> - <8,16> -> <11,19> -> <1,19>
    auto dx = norm(sub_k<20>(D, X3));           // <7,28> -> <1,28>
    auto C2 = add(C, C);
    auto C4 = add(C2, C2);
    auto C8 = add(C4, C4);                      // <8,16>
    auto Y3 = norm(sub_k<17>(mul(E, dx), C8));  // mul: 14*3+15, 6*28 ok; -> <1,19>
    auto YZ = mul(a.y, a.z);                    // 14*2+15 ok, 20*4 ok
    a.x = widen<1, 20>(X3);
    a.y = widen<1, 20>(Y3);
    a.z = add(YZ, YZ);
}

// a += (x2, neg ? -y2 : y2), the point affine on the curve a lives on; complete
HD void je28_madd(JE28 &a, bool &ainf, const F28<1, 20> &x2, const F28<1, 20> &y2in, bool neg) {
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    F28<1, 41> y2;
    {
        auto ny = norm(sub_k<21>(zero, y2in));   // <3,21> -> <1,21>
#pragma unroll
        for (int j = 0; j < 14; j++) y2.l[j] = neg ? ny.l[j] : y2in.l[j];
    }
    if (ainf) {
        a.x = x2;
        a.y = widen<1, 20>(mul(y2, f28_one()));
        a.z = widen<2, 4>(f28_one());
        ainf = false;
        return;
    }
    auto z1z1 = sqr(a.z);
    auto u2 = mul(x2, z1z1);
    auto s2 = mul(y2, mul(a.z, z1z1));          // 41*2 ok
    auto h = sub_k<21>(u2, a.x);                // <4,23>
    auto r = sub_k<21>(s2, a.y);                // <4,23>
    auto hh = sqr(h);                           // 15*16+15 = 255 ok, 529 ok
    if (is_zero(hh)) {
        if (is_zero(mul(r, f28_one()))) {
            a.x = x2;                           // acc == the table point: double it
            a.y = widen<1, 20>(mul(y2, f28_one()));
            a.z = widen<2, 4>(f28_one());
            je28_dbl(a);
        } else {
            ainf = true;
        }
        return;
    }
    auto hhh = mul(h, hh);
    auto v = mul(a.x, hh);
    auto rr = sqr(r);
    auto x3 = norm(sub(rr, add(hhh, add(v, v))));   // <1,2> - <3,6> -> <6,10> -> <1,10>
    auto dv = sub(v, x3);                           // <4,18>
    auto s1n = sub_k<21>(zero, a.y);                // <3,21>
    // r*dv - Y1*hhh with one reduction: limbs 14*(1*4 + 3*1)+15 ok; values 23*18 + 21*2 ok
    auto y3 = mul_add2(norm(r), dv, s1n, hhh);
    auto z3 = mul(a.z, h);                          // 14*8+15 ok, 4*23 ok
    a.x = widen<1, 20>(x3);
    a.y = widen<1, 20>(y3);
    a.z = widen<2, 4>(z3);
}

// One co-Z addition (X1,Y1) + (X2,Y2), both over the same Z: returns the sum in (x3, y3), replaces (X1,Y1) by
// the same point over the new Z3 = Z * dX and (X2,Y2) likewise; c = dX^2 and w = dX^3 rescale any further point.
struct CoZ28 {
    F28<1, 20> x, y;
};
HD void coz28_addu(CoZ28 &sum, CoZ28 &p1, CoZ28 &p2, F28<1, 2> &z, F28<1, 2> &c, F28<4, 6> &w) {
    auto dX = sub_k<21>(p2.x, p1.x);                // <4,41>
    auto dY = sub_k<21>(p2.y, p1.y);
    c = sqr(dX);                                    // 15*16+15 ok, 41^2 = 1681 ok
    auto W1 = mul(p1.x, c);
    auto W2 = mul(p2.x, c);
    auto D = sqr(dY);
    w = sub(W2, W1);                                // <4,6>  = dX^3
    auto A1 = mul(p1.y, w);                         // 14*4+15 ok, 120 ok
    auto A2 = mul(p2.y, w);
    auto X3 = norm(sub(D, add(W1, W2)));            // <1,2> - <2,4> -> <5,10> -> <1,10>
    auto Y3 = norm(sub(mul(dY, sub(W1, X3)), A1));  // inner sub <4,18>: 14*16+15 ok, 41*18 ok; -> <4,6> -> <1,6>
    z = mul(z, dX);                                 // 14*4+15 ok, 82 ok
    sum.x = widen<1, 20>(X3);
    sum.y = widen<1, 20>(Y3);
    p1.x = widen<1, 20>(W1);
    p1.y = widen<1, 20>(A1);
    p2.x = widen<1, 20>(W2);
    p2.y = widen<1, 20>(A2);
}
HD void coz28_rescale(CoZ28 &q, const F28<1, 2> &c, const F28<4, 6> &w) {
    q.x = widen<1, 20>(mul(q.x, c));
    q.y = widen<1, 20>(mul(q.y, w));
}

// tbl[m] = (2m+1)P for a finite P of the prime-order subgroup (so no step is exceptional), zc = the common Z
HDNI inline void eat28_build(EAT28 (&tbl)[4], F28<1, 2> &zc, const XYZZ28 &p) {
    // P as a Jacobian triple: (X, Y, Z) = (x*zz, y*zzz, zz)
    auto X1 = mul(p.x, p.zz);
    auto Y1 = mul(p.y, p.zzz);
    // doubling that also returns P over 2P's Z2 = 2 Y1 Z1:  P = (4 X1 Y1^2, 8 Y1^4, Z2)
    auto A = sqr(X1);
    auto B = sqr(Y1);
    auto C = sqr(B);
    auto XB = mul(X1, B);
    auto XB2 = add(XB, XB);
    auto D = add(XB2, XB2);                         // <4,8>
    auto E = add(add(A, A), A);                     // <3,6>
    auto F = sqr(E);
    auto X2 = norm(sub_k<17>(F, add(D, D)));        // <1,19>
    auto dx = norm(sub_k<20>(D, X2));               // <1,28>
    auto C2 = add(C, C);
    auto C4 = add(C2, C2);
    auto C8 = add(C4, C4);                          // <8,16>
    auto Y2 = norm(sub_k<17>(mul(E, dx), C8));      // <1,19>
    auto YZ = mul(Y1, p.zz);
    F28<1, 2> z = mul(add(YZ, YZ), f28_one());      // Z2, back under <1,2>
    CoZ28 t1, t2, t3, t5, t7;
    t1.x = widen<1, 20>(norm(D));
    t1.y = widen<1, 20>(norm(C8));
    t2.x = widen<1, 20>(X2);
    t2.y = widen<1, 20>(Y2);
    F28<1, 2> c;
    F28<4, 6> w;
    coz28_addu(t3, t2, t1, z, c, w);                // 3P = 2P + P; 2P and P follow
    coz28_addu(t5, t2, t3, z, c, w);                // 5P = 2P + 3P
    coz28_rescale(t1, c, w);
    coz28_addu(t7, t2, t5, z, c, w);                // 7P = 2P + 5P
    coz28_rescale(t1, c, w);
    coz28_rescale(t3, c, w);
    const F28<1, 1> beta = f28_const<1, 1>(FP28_BETA_LAMBDA);
    const CoZ28 *src[4] = {&t1, &t3, &t5, &t7};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        tbl[m].x = src[m]->x;
        tbl[m].y = src[m]->y;
        tbl[m].bx = widen<1, 20>(mul(src[m]->x, beta));
    }
    zc = z;
}

// [k]P = [k1]P + [k2]phi(P) with both halves given in width-4 NAF (wnaf4_128): 131 doublings at most
// and ~52 mixed additions from the effectively affine table {P, 3P, 5P, 7P} (x, beta*x, y; signs on the fly).
// The schedule depends on the digits, so it is meant for callers whose lanes share the
// scalar: the G1 FFT stage kernels, where a wave works on one twiddle.  Only for points of the
// prime-order subgroup.
HDNI inline void xyzz28_mul_glv_naf(XYZZ28 &out, bool &out_inf, const XYZZ28 &p, bool p_inf, const int8_t *naf1,
                                    const int8_t *naf2) {
    bool inf = true;
    if (!p_inf) {
        EAT28 tbl[4];
        F28<1, 2> zc;
        eat28_build(tbl, zc, p);
        JE28 acc;
        for (int i = GLV_NAF_LEN - 1; i >= 0; i--) {
            if (!inf) je28_dbl(acc);
            const int d1 = naf1[i], d2 = naf2[i];
            if (d1) {
                const EAT28 &e = tbl[(d1 > 0 ? d1 : -d1) >> 1];
                je28_madd(acc, inf, e.x, e.y, d1 < 0);
            }
            if (d2) {
                const EAT28 &e = tbl[(d2 > 0 ? d2 : -d2) >> 1];
                je28_madd(acc, inf, e.bx, e.y, d2 < 0);
            }
        }
        if (!inf) {
            JAC28 j;                                  // home: Z * Zc
            j.x = widen<1, 34>(acc.x);
            j.y = widen<1, 34>(acc.y);
            j.z = widen<2, 4>(mul(acc.z, zc));
            out = jac28_to_xyzz(j);
        }
    }
    out_inf = inf;
}

// [k]P = [k1]P + [k2]phi(P), phi(X, Y, Z) = (beta*X, Y, Z) = [lambda]P for P in G1, for lanes with
// DIFFERENT scalars: a uniform 4-bit window schedule (4 doublings, then one table addition per half) so
// that the lanes stay in step; 128 doublings instead of 256.  glv = {k1[4], k2[4]} (glv_split).
// Jacobian coordinates inside.  Only for points of the prime-order subgroup (the endomorphism is
// [lambda] only there, and every multiple 1..15 of such a point is finite).
HDNI inline void xyzz28_mul_glv_w4(XYZZ28 &out, bool &out_inf, const XYZZ28 &p, bool p_inf, const uint32_t *glv) {
    JACT28 tbl[15];
    JAC28 acc;
    bool inf = true;
    if (!p_inf) {
        const F28<1, 1> beta = f28_const<1, 1>(FP28_BETA_LAMBDA);
        JAC28 cur = jac28_from_xyzz(p);
        tbl[0] = jac28_table_entry(cur);
        for (int i = 1; i < 15; i++) {
            bool ci = false;
            jac28_add(cur, ci, tbl[0]);
            tbl[i] = jac28_table_entry(cur);
        }
        for (int w = 31; w >= 0; w--) {
            if (!inf) {
                jac28_dbl(acc);
                jac28_dbl(acc);
                jac28_dbl(acc);
                jac28_dbl(acc);
            }
            uint32_t d1 = (glv[w >> 3] >> ((w & 7) * 4)) & 15u;
            if (d1) jac28_add(acc, inf, tbl[d1 - 1]);
            uint32_t d2 = (glv[4 + (w >> 3)] >> ((w & 7) * 4)) & 15u;
            if (d2) {
                JACT28 e = tbl[d2 - 1];
                e.x = widen<1, 34>(mul(e.x, beta));
                jac28_add(acc, inf, e);
            }
        }
    }
    if (!inf) out = jac28_to_xyzz(acc);
    out_inf = inf;
}

// [k]P for a 128-bit k (one GLV half), uniform 4-bit windows, Jacobian inside: 128 doublings, 32 table
// additions.  The variable-base sums give each GLV half its own lane (the second one on phi(P)) instead
// of interleaving both in one ladder: the lanes are idle anyway and the dependent chain is what costs.
HDNI inline void xyzz28_mul_w4_128(XYZZ28 &out, bool &out_inf, const XYZZ28 &p, bool p_inf, const uint32_t *k) {
    JACT28 tbl[15];
    JAC28 acc;
    bool inf = true;
    if (!p_inf) {
        JAC28 cur = jac28_from_xyzz(p);
        tbl[0] = jac28_table_entry(cur);
        for (int i = 1; i < 15; i++) {
            bool ci = false;
            jac28_add(cur, ci, tbl[0]);
            tbl[i] = jac28_table_entry(cur);
        }
        for (int w = 31; w >= 0; w--) {
            if (!inf) {
                jac28_dbl(acc);
                jac28_dbl(acc);
                jac28_dbl(acc);
                jac28_dbl(acc);
            }
            uint32_t d = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
            if (d) jac28_add(acc, inf, tbl[d - 1]);
        }
    }
    if (!inf) out = jac28_to_xyzz(acc);
    out_inf = inf;
}

// a^e for a public exponent e (little-endian limbs, nbits bits) by a 4-bit sliding window: nbits
// squarings and ~nbits/5 multiplications (odd powers a, a^3, ..., a^15 precomputed).
HDNI inline F28<1, 2> f28_pow_public(const F28<1, 2> &a, const uint32_t *e, int nbits) {
    F28<1, 2> odd[8];
    odd[0] = a;
    F28<1, 2> a2 = sqr(a);
    for (int i = 1; i < 8; i++) odd[i] = mul(odd[i - 1], a2);
    F28<1, 2> acc = widen<1, 2>(f28_one());
    int i = nbits - 1;
    while (i >= 0) {
        if (!((e[i >> 5] >> (i & 31)) & 1u)) {
            acc = sqr(acc);
            i--;
            continue;
        }
        // longest window [i, i-w+1], w <= 4, that ends in a set bit
        int w = i >= 3 ? 4 : i + 1;
        uint32_t bits = 0;
        for (int k = 0; k < w; k++) bits = (bits << 1) | ((e[(i - k) >> 5] >> ((i - k) & 31)) & 1u);
        while (!(bits & 1u)) {
            bits >>= 1;
            w--;
        }
        for (int k = 0; k < w; k++) acc = sqr(acc);
        acc = mul(acc, odd[bits >> 1]);
        i -= w;
    }
    return acc;
}

// a^(p-2): 381 squarings and ~80 multiplications.  Kept as the independent cross-check of the
// safegcd inverse (tests/test_host_arith.py).
HDNI inline F28<1, 2> f28_inv_fermat(const F28<1, 2> &a) {
    uint32_t e[12];
    for (int i = 0; i < 12; i++) e[i] = FP_INV_EXP[i];
    return f28_pow_public(a, e, 381);
}

// candidate square root a^((p+1)/4) (p = 3 mod 4); the caller checks the square
HDNI inline F28<1, 2> f28_sqrt_candidate(const F28<1, 2> &a) {
    uint32_t e[12];
    for (int i = 0; i < 12; i++) e[i] = FP_SQRT_EXP[i];
    return f28_pow_public(a, e, 381);
}

// a == b mod p for lazily reduced operands
template <int LA, int VA, int LB, int VB>
HD bool f28_equal(const F28<LA, VA> &a, const F28<LB, VB> &b) {
    return is_zero(mul(sub(a, b), f28_one()));
}

// y with y^2 = x^3 + 4, or false if x is not the abscissa of a curve point
HDNI inline bool g1_28_solve_y(F28<1, 2> &y, const F28<1, 2> &x) {
    auto rhs = mul(add(mul(sqr(x), x), f28_const<1, 1>(FP28_FOUR)), f28_one());  // <1,2>
    y = f28_sqrt_candidate(rhs);
    return f28_equal(sqr(y), rhs);
}

// [|x|]P for the BLS parameter |x| = 0xd201000000010000 = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16:
// 63 doublings and 5 additions, the same for every lane (Jacobian coordinates)
HDNI inline void jac28_mul_bls_x(JAC28 &out, bool &out_inf, const JAC28 &p, bool p_inf) {
    JAC28 acc = p;
    bool inf = p_inf;
    if (!p_inf) {
        const JACT28 pt = jac28_table_entry(p);
        for (int b = 62; b >= 0; b--) {
            if (!inf) jac28_dbl(acc);
            if (b == 62 || b == 60 || b == 57 || b == 48 || b == 16) jac28_add(acc, inf, pt);
        }
    }
    out = acc;
    out_inf = inf;
}

// Subgroup test for a finite curve point (x, y): P is in G1 iff phi2(P) = [-x^2]P with
// phi2(X, Y) = (beta^2 X, Y), i.e. iff [x^2]P = (beta^2 X, -Y).  Exact: E(Fp) = G1 x T with the
// exponent of T dividing x - 1, so on T the right side is -T' while phi2^2 + phi2 + 1 = 0 forces
// phi2(T') = -T' only for T' = 0.  126 doublings + 10 additions instead of a 255-bit ladder by r
// (the reference's blst performs an endomorphism-based test of the same kind).
HDNI inline bool g1_28_in_subgroup(const F28<1, 2> &x, const F28<1, 2> &y) {
    JAC28 p, q1, q;
    p.x = widen<1, 34>(x);
    p.y = widen<1, 34>(y);
    p.z = widen<2, 4>(f28_one());
    bool i1, i2;
    jac28_mul_bls_x(q1, i1, p, false);
    jac28_mul_bls_x(q, i2, q1, i1);
    if (i2) return false;
    auto zz = sqr(q.z);
    auto bx = mul(x, f28_const<1, 1>(FP28_BETA_LAMBDA2));
    if (!f28_equal(q.x, mul(bx, zz))) return false;
    // q.y == -y * z^3  <=>  q.y + y * z^3 == 0
    return is_zero(mul(add(q.y, mul(y, mul(q.z, zz))), f28_one()));
}

// the inversion used by the kernels: safegcd (fp28_inv.hpp), ~12x fewer instructions than the ladder
HD F28<1, 2> f28_inv(const F28<1, 2> &a) { return f28_inv_safegcd(a); }

// 1/a for the 12-limb representation through the safegcd inverse (host glue and kernels that still hold Fp values)
HDNI inline Fp fp_inv_safegcd(const Fp &a) { return f28_to_fp(f28_inv(f28_from_fp(a))); }
HDNI inline G1Affine jac_to_affine_fast(const G1Jac &p) {
    if (p.is_inf()) return G1Affine::inf();
    Fp zi = fp_inv_safegcd(p.z);
    Fp zi2 = sqr(zi);
    return {mul(p.x, zi2), mul(p.y, mul(zi2, zi))};
}

// affine coordinates (fully reduced, 2^384 domain) of a point in the 28-bit domain
HDNI inline G1Affine xyzz28_to_affine(const XYZZ28 &a, bool inf) {
    if (inf) return G1Affine::inf();
    auto t = f28_inv(a.zzz);   // 1/z^3
    auto u = mul(a.zz, t);     // 1/z
    return {f28_to_fp(mul(a.x, sqr(u))), f28_to_fp(mul(a.y, t))};
}

// out of the 28-bit domain: fully reduced coordinates in the host/LDS representation
HD G1XYZZ xyzz28_to_xyzz(const XYZZ28 &a, bool inf) {
    if (inf) return G1XYZZ::inf();
    return {f28_to_fp(a.x), f28_to_fp(a.y), f28_to_fp(a.zz), f28_to_fp(a.zzz)};
}

}  // namespace ckzg
