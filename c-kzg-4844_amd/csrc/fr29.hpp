// fr29.hpp -- Fr on 9 limbs of 29 bits, Montgomery radix 2^261: the arithmetic of the barycentric evaluation
// (verify.hip: k_eval_barycentric; evaluate_polynomial_in_evaluation_form, src/eip4844/eip4844.c:192-240).
//
// Why a second Fr: the 8 x 32-bit CIOS product of field.hpp costs two carry instructions per multiply-add (~420 VALU
// instructions).  With 29-bit limbs a 64-bit column accumulator absorbs all 18 partial products of a column of a
// Montgomery product (18 * 2^58 < 2^63) without a carry, as fp28.hpp does for Fp: 162 multiply-adds + 17 shifts.  And
// r = 1 mod 2^32, so -1/r mod 2^29 is -1 and the quotient digit of a column is the negated accumulator.
//
// Values are lazily reduced: limbs are always < 2^29 (the top one holds what is left), the value may exceed r.  A
// product of a < A r and b < B r comes out below (A B r / 2^261 + 1) r = (0.00708 A B + 1) r; every operand below
// stays under 2^258 (17 r), every sum under 2^261.  Interchange with the library's Fr (Montgomery radix 2^256, eight
// 32-bit words, canonical) is a re-packing of the same integer, and the radix travels with the operand:
//   mul29(x 2^256, y 2^261) = x y 2^256      -- a term of the sum comes out in the library's form directly
//   mul29(x 2^256, 2^266)   = x 2^261        -- into this form
#pragma once
#include "fr_inv.hpp"

namespace ckzg {

struct Fr29 {
    uint32_t l[9];
};

constexpr uint32_t M29 = (1u << 29) - 1u;

// the same integer, eight 32-bit words -> nine 29-bit limbs (and back; the value must be < 2^256 on the way back)
HD Fr29 fr29_pack(const uint32_t *w) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, j = bit >> 5, sh = bit & 31;
        uint32_t v = w[j] >> sh;
        if (sh > 3 && j + 1 < 8) v |= w[j + 1] << (32 - sh);
        r.l[i] = v & M29;
    }
    return r;
}
HD void fr29_unpack(uint32_t *w, const Fr29 &a) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, i = bit / 29, sh = bit - 29 * i;
        uint32_t v = a.l[i] >> sh;
        v |= a.l[i + 1] << (29 - sh);
        if (29 - sh + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - sh);
        w[k] = v;
    }
}
HD Fr29 fr29_const(const uint32_t (&c)[9]) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c[i];
    return r;
}

// a b / 2^261 mod r, lazily reduced (see the bounds above)
HD Fr29 fr29_mul_inline(const Fr29 &a, const Fr29 &b) {
    uint64_t acc = 0;
    uint32_t q[9];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * FR29_RMUL[0][k - i];
        q[k] = (0u - (uint32_t)acc) & M29;   // -1/r = -1 mod 2^29
        acc += q[k];                          // r's lowest limb is 1: the column's low 29 bits vanish
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * FR29_RMUL[0][k - i];
        r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
// On the device a product is a real function: the evaluation's two loops of 16 terms only unroll (and keep their
// prefix products in registers) while their bodies are calls.  Eighteen scalar arguments, because two nine-word
// structures by value exceed what the calling convention passes in registers and would travel through scratch.
__device__ __noinline__ Fr29 fr29_mul_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                           uint32_t a6, uint32_t a7, uint32_t a8, uint32_t b0, uint32_t b1, uint32_t b2,
                                           uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8) {
    const Fr29 a = {{a0, a1, a2, a3, a4, a5, a6, a7, a8}}, b = {{b0, b1, b2, b3, b4, b5, b6, b7, b8}};
    return fr29_mul_inline(a, b);
}
__device__ __forceinline__ Fr29 fr29_mul(const Fr29 &a, const Fr29 &b) {
    return fr29_mul_regs(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], b.l[0], b.l[1], b.l[2],
                         b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8]);
}
#else
HD Fr29 fr29_mul(const Fr29 &a, const Fr29 &b) { return fr29_mul_inline(a, b); }
#endif

// carries of a sum of up to eight lazily reduced values, limb by limb
HD void fr29_carry(Fr29 &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.l[i + 1] += a.l[i] >> 29;
        a.l[i] &= M29;
    }
}
HD Fr29 fr29_add(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    fr29_carry(r);
    return r;
}
// a - b for b < 2^K r: a + (2^K r - b)
template <int K>
HD Fr29 fr29_sub_below(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t v = (int32_t)(a.l[i] + FR29_RMUL[K][i]) - (int32_t)b.l[i] + c;
        r.l[i] = i < 8 ? ((uint32_t)v & M29) : (uint32_t)v;
        c = v >> 29;
    }
    return r;
}
HD Fr29 fr29_sub_canonical(const Fr29 &a, const Fr29 &b) { return fr29_sub_below<0>(a, b); }
HD bool fr29_equal(const Fr29 &a, const Fr29 &b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) d |= a.l[i] ^ b.l[i];
    return d == 0;
}
// a < 2^(K+1) r  ->  the canonical representative (conditional subtractions of 2^K r, ..., 2 r, r)
template <int K>
HD Fr29 fr29_canonical(Fr29 a) {
#pragma unroll
    for (int k = K; k >= 0; k--) {
        Fr29 s;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int32_t v = (int32_t)a.l[i] - (int32_t)FR29_RMUL[k][i] + c;
            s.l[i] = i < 8 ? ((uint32_t)v & M29) : (uint32_t)v;
            c = v >> 29;
        }
        const bool keep = (int32_t)s.l[8] < 0;   // a < 2^k r
#pragma unroll
        for (int i = 0; i < 9; i++) a.l[i] = keep ? a.l[i] : s.l[i];
    }
    return a;
}

// library form (canonical, radix 2^256) -> this form (canonical, radix 2^261), and back
HD Fr29 fr29_from_fr(const Fr &x) { return fr29_canonical<0>(fr29_mul(fr29_pack(x.l), fr29_const(FR29_2POW266))); }
HD Fr fr29_to_fr(const Fr29 &x) {
    Fr r;
    fr29_unpack(r.l, fr29_canonical<0>(fr29_mul(x, fr29_const(FR29_2POW256))));
    return r;
}

// 2^261 / x for a canonical x 2^261 != 0 (safegcd, fr_inv.hpp's iteration on the same 30-bit limbs); below 2 r
HDNI inline Fr29 fr29_inv(const Fr29 &a) {
    int32_t f[9], g[9], d[9], e[9];
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, j = bit / 29, sh = bit - 29 * j;
        uint32_t v = a.l[j] >> sh;
        if (j + 1 < 9) v |= a.l[j + 1] << (29 - sh);
        if (29 - sh + 29 < 30 && j + 2 < 9) v |= a.l[j + 2] << (58 - sh);
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
    // f = +-1; f * d made positive by adding 32 r (|d| < 27 r): y = 1/(x 2^261) as an integer below 59 r < 2^261
    const bool negate = f[8] < 0;
    int64_t c = 0;
    uint32_t w[9];
    for (int i = 0; i < 9; i++) {
        c += (int64_t)FR30_32R[i] + (negate ? -(int64_t)d[i] : (int64_t)d[i]);
        w[i] = (uint32_t)(c & 0x3fffffff);
        c >>= 30;
    }
    w[8] += (uint32_t)(c << 30);
    Fr29 y;
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, i = bit / 30, sh = bit - 30 * i;
        uint32_t v = w[i] >> sh;
        if (i + 1 < 9 && 30 - sh < 29) v |= w[i + 1] << (30 - sh);
        y.l[j] = j == 8 ? v : (v & M29);
    }
    return fr29_mul(y, fr29_const(FR29_R3));   // y 2^783 / 2^261 = 2^261 / x
}

// ------------------------------------------------------------------------------------------
// The pieces of the barycentric evaluation (k_eval_barycentric; replayed on the host by tests/test_host_arith.py
// through host_shim.cpp).  A thread owns PER terms i = first + k * stride of
//   sum_i p_i w_i / (z - w_i)
// roots29: the domain in this form (canonical, radix 2^261); poly: the library's Fr (canonical, radix 2^256).
// ------------------------------------------------------------------------------------------
namespace ev29 {

constexpr int PER = 8;       // terms per thread
constexpr int WAVES = 8;     // waves of a workgroup: 8 x 64 x PER = 4096

// prefix products of the thread's denominators (Montgomery's trick); returns the index of a domain point equal to
// z, or -1.  pre[k] = prod_{j<k} (z - w_j), acc = the whole product (< 1.1 r).
HD int forward(Fr29 *pre, Fr29 &acc, const Fr29 &z, const Fr29 *roots29, int first, int stride) {
    int hit = -1;
    acc = fr29_const(FR29_ONE);
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = first + k * stride;
        const Fr29 root = roots29[i];
        if (fr29_equal(z, root)) hit = i;
        pre[k] = acc;
        acc = fr29_mul(acc, fr29_sub_canonical(z, root));
    }
    return hit;
}

// One lane of the first wave: the WAVES products get(w) (lane l of every wave) -> put(w, 2^261 / get(w)), with ONE
// inversion.  A wave executes an inversion's instructions whether one lane needs it or 64: what has to be spread
// is the number of waves that run one.  park / parked keep the prefix products outside the registers.
template <class Get, class Park, class Parked, class Put, class Inv>
HD void invert_across(Get get, Park park, Parked parked, Put put, Inv invert) {
    Fr29 t = get(0);
    for (int w = 1; w < WAVES; w++) {
        park(w, t);                      // prod_{v < w} get(v)
        t = fr29_mul(t, get(w));
    }
    Fr29 inv = invert(fr29_canonical<0>(t));
    for (int w = WAVES - 1; w >= 1; w--) {
        const Fr29 g = get(w);
        put(w, fr29_mul(inv, parked(w)));
        inv = fr29_mul(inv, g);
    }
    put(0, inv);
}

// inv = 2^261 / acc on entry.  Returns the thread's part of the sum in the LIBRARY's radix (below 2^5 r, limbs
// carried); di_out (may be null) receives 1/(z - w_i) in this form, canonical, packed in eight words per term.
HD Fr29 backward(const Fr29 *pre, Fr29 inv, const Fr29 &z, const Fr29 *roots29, const Fr *poly, int first, int stride,
                 uint32_t *di_out) {
    Fr29 sum;
#pragma unroll
    for (int i = 0; i < 9; i++) sum.l[i] = 0;
#pragma unroll
    for (int k = PER - 1; k >= 0; k--) {
        const int i = first + k * stride;
        const Fr29 root = roots29[i];
        const Fr29 di = fr29_mul(inv, pre[k]);             // 1/(z - w_i)
        if (k) inv = fr29_mul(inv, fr29_sub_canonical(z, root));
        if (di_out) fr29_unpack(di_out + (size_t)i * 8, fr29_canonical<0>(di));
        const Fr29 t = fr29_mul(fr29_pack(poly[i].l), fr29_mul(di, root));   // p_i w_i / (z - w_i), radix 2^256
#pragma unroll
        for (int j = 0; j < 9; j++) sum.l[j] += t.l[j];
        if ((k & 3) == 0) fr29_carry(sum);
    }
    return sum;
}

// the library's Fr from a lazily reduced sum in the library's radix (below 2^6 r)
HD Fr to_fr_radix256(const Fr29 &s) {
    Fr r;
    fr29_unpack(r.l, fr29_canonical<5>(s));
    return r;
}

// (z^4096 - 1)/4096 in this form, and y = sum * that for a sum in the library's form
HD Fr29 vanishing_over_n(const Fr29 &z) {
    Fr29 zn = z;
    for (int k = 0; k < 12; k++) zn = fr29_mul(zn, zn);
    return fr29_mul(fr29_sub_canonical(zn, fr29_const(FR29_ONE)), fr29_const(FR29_INV4096));
}
HD Fr scale(const Fr &sum, const Fr29 &f) {
    Fr r;
    fr29_unpack(r.l, fr29_canonical<0>(fr29_mul(fr29_pack(sum.l), f)));
    return r;
}


// ------------------------------------------------------------------------------------------
// The evaluation WITHOUT inversions (k_eval_tree).  With x_i = z / w_i the Lagrange basis over the roots of unity is
//   L_i(z) = (1/n) sum_{k<n} x_i^k = (1/n) prod_{l < log n} (1 + x_i^(2^l)),
// and in bit-reversed order the leaves under a node j of level l share x^(2^l) = z^(2^l) / brp_roots[j] while the
// two children of a node carry opposite signs of it.  So p(z) = (1/n) * root of the tree
//   parent = (even + odd) + X * (even - odd),     X = z^(2^l) * tab[m],   tab[m] = 1 / brp_roots[2 m]
// -- two products per node, 2 (n - 1) per polynomial (the barycentric sum pays five per term and an inversion), no
// division, and a z inside the domain (eip4844.c:208-215) is not a special case: the identity is polynomial.
// Node values stay in the library's radix (leaves are the polynomial's words as they lie), the X in this form's.
// Bounds: a node of level c + 1 is below 2 B_c + 1.4 with children below B_c: 1, 3.3, 7.7, 16.6 r -> reduced to the
// canonical value at every fourth level; the subtraction adds the multiple 2^k r >= B_c of r.
// ------------------------------------------------------------------------------------------
template <int C, bool FLAT = false>   // C = the children's level; FLAT: the product inlined (no call on the device)
HD Fr29 tree_combine(const Fr29 &e, const Fr29 &o, const Fr29 &x) {
    constexpr int K = (C % 4 == 0) ? 0 : (C % 4 == 1) ? 2 : (C % 4 == 2) ? 3 : 5;
    const Fr29 t = FLAT ? fr29_mul_inline(x, fr29_sub_below<K>(e, o)) : fr29_mul(x, fr29_sub_below<K>(e, o));
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = e.l[i] + o.l[i] + t.l[i];
    fr29_carry(r);
    if ((C + 1) % 4 == 0) return fr29_canonical<5>(r);
    return r;
}
// the canonical value of a node of level L
template <int L>
HD Fr29 tree_canonical(const Fr29 &v) {
    if (L % 4 == 0) return v;
    return fr29_canonical<(L % 4 == 1) ? 1 : (L % 4 == 2) ? 2 : 4>(v);
}
// the node of level L above the leaves [base, base + 2^L) (base a multiple of 2^L); zp[l] = z^(2^l)
template <int L, class Leaf>
HD Fr29 tree_node(Leaf leaf, const Fr29 *tab, const Fr29 *zp, int base) {
    if constexpr (L == 0) {
        return leaf(base);
    } else {
        const Fr29 e = tree_node<L - 1>(leaf, tab, zp, base);
        const Fr29 o = tree_node<L - 1>(leaf, tab, zp, base + (1 << (L - 1)));
        return tree_combine<L - 1>(e, o, fr29_mul(zp[L - 1], tab[base >> L]));
    }
}
// y from the root (library's radix) -- the factor 1/n
HD Fr tree_finish(const Fr29 &root) {
    Fr r;
    fr29_unpack(r.l, fr29_canonical<0>(fr29_mul(root, fr29_const(FR29_INV4096))));
    return r;
}
// The tree over the blob's own bytes (k_eval_tree's BYTES form: blob_to_polynomial, src/eip4844/blob.c:31-38, is
// folded into the evaluation).  A leaf is the canonical integer of a field element -- radix 1, not 2^256 -- and the
// whole tree is linear in the leaves, so the radix is put right once, at the root: the factor 2^256 / n.
// s: the element's eight 32-bit words, least significant first.  An element >= r (bytes_to_bls_field,
// src/common/bytes.c:52-70, rejects it) sets bad and counts as zero.
HD Fr29 tree_leaf_from_words(const uint32_t *s, uint32_t &bad) {
    Fr29 v = fr29_pack(s);
    int32_t t = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) t = ((int32_t)v.l[i] - (int32_t)FR29_RMUL[0][i]) + (t >> 29);
    const bool ge = t >= 0;   // no borrow out of the top limb: v >= r
    bad |= ge ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 9; i++) v.l[i] = ge ? 0u : v.l[i];
    return v;
}
HD Fr tree_finish_from_integers(const Fr29 &root) {
    Fr r;
    fr29_unpack(r.l, fr29_canonical<0>(fr29_mul(root, fr29_const(FR29_INV4096_R256))));
    return r;
}

}  // namespace ev29

}  // namespace ckzg
