// fp28.hpp -- the device-side Fp multiplier the MI355X actually wants: 14 limbs of 28 bits.
//
// Measured on gfx950 (tools/ubench/instr_rates.hip): v_mad_u64_u32 issues at ~4.8 cycles per
// wave-instruction, but every carry-propagating instruction around it (v_add_co/v_addc_co,
// v_lshl_add_u64) costs ~4.3 as well, and the 12 x 32-bit CIOS product needs two of those per
// multiply-add.  With 28-bit limbs a 64-bit column accumulator has 8 spare bits, so a column can
// absorb all 28 partial products of a Montgomery multiplication without a single carry:
//     T[j] += a[j] * b[i]          one v_mad_u64_u32, accumulator updated in place
// 392 multiply-adds + ~100 cheap ops per product instead of 288 + ~950 (1.8x faster, 64 VGPRs).
//
// Values are kept lazily reduced.  F28<LB, VB> carries compile-time bounds: every limb is
// < LB * 2^28 and the value is <= VB * p.  Each operation computes the bounds of its result and
// static_asserts the preconditions that make overflow impossible, so "random tests pass" is not
// what correctness rests on.  Montgomery radix is 2^392 (device-private; tables are converted at
// build time, results converted back before they leave the kernel).
#pragma once
#include "field.hpp"

namespace ckzg {

constexpr uint32_t M28 = 0x0fffffffu;

template <int LB, int VB>
struct F28 {
    uint32_t l[14];
};

namespace f28detail {
struct Limbs14 {
    uint32_t v[14];
};
// k*p with every limb except the top one raised by c*2^28 (and the next limb lowered by c), so
// that a limb-wise "a + m - b" cannot go negative for b's limbs < c*2^28.
constexpr Limbs14 spread_multiple(int k, int c) {
    Limbs14 r{};
    uint64_t carry = 0;
    for (int j = 0; j < 14; j++) {
        uint64_t t = (uint64_t)FP28_P[j] * (uint64_t)k + carry;
        r.v[j] = (uint32_t)(t & M28);
        carry = t >> 28;
    }
    r.v[13] += (uint32_t)(carry << 28);  // k*p < 2^392 for k <= 1024: carry is 0
    for (int j = 0; j < 13; j++) {
        r.v[j] += (uint32_t)c << 28;
        r.v[j + 1] -= (uint32_t)c;
    }
    return r;
}
constexpr int pow2_above(int v) {  // smallest power of two > v
    int k = 1;
    while (k <= v) k <<= 1;
    return k;
}
}  // namespace f28detail

// Montgomery product a*b/2^392 mod p: result limbs normalised (< 2^28), value < 2p.
//
// Column-wise (product-scanning) form: ONE 64-bit accumulator walks the 28 columns of a*b + q*p.  Column k
// receives its a_i*b_(k-i) and q_i*p_(k-i) terms, the first 14 columns each fix their quotient digit
// q_k = -column/p mod 2^28, and the accumulator is shifted down by 28 bits into the next column -- that shift
// is the whole carry handling, and the upper 14 columns leave the result limbs already normalised.
// Measured on gfx950 (tools/ubench/mad_latency.hip): 64-bit shifts and adds cost as much issue time as a
// v_mad_u64_u32 (4.6 vs 4.85 cycles at two waves per SIMD) and a dependent chain of mads runs at the same
// rate as 16 independent ones, so the single accumulator costs nothing and needs 28 fewer live registers
// than the row-wise form's 15 accumulators (rounds 1-2; profiles/r02_fp28_ab.txt).
// acc += a * b: one v_mad_u64_u32.  (LLVM re-associates a column's long sum into two chains -- the a*b terms
// and the q*p terms -- and joins them with a 64-bit add, so the instruction counts of this form and of the
// row-wise one come out equal: 3548 mads + 236 v_lshl_add_u64 + 234 v_lshrrev_b64 per mixed addition.  Forcing a
// single chain with inline assembly removes the 236 adds but makes the compiler pad every asm statement with
// an s_nop and pessimises the ladder kernels: measured 10.77 ms per 1024-blob launch against 10.42 ms
// row-wise and 10.22 ms for this plain form, which wins through its smaller register footprint --
// 191 instead of 209 VGPRs.  tools/ab_fp28.sh, profiles/r02_fp28_ab.txt.)
HD void mad64(uint64_t &acc, uint32_t a, uint32_t b) { acc += (uint64_t)a * b; }
HD void mad64c(uint64_t &acc, uint32_t a, uint32_t c) { acc += (uint64_t)a * c; }

// CKZG_F28_ASM_BLOCKS (device code of the translation units that define it -- msm.hip): the same three
// routines with each column's multiply-adds as one inline-asm block per operand group, which keeps the single
// accumulator chain LLVM otherwise splits (tools/gen_fp28_asm.py explains; fp28_asm_cols.inc is generated).
#if defined(__HIP_DEVICE_COMPILE__) && defined(CKZG_F28_ASM_BLOCKS)
#define CKZG_F28_USE_ASM_BLOCKS 1
#include "fp28_asm_cols.inc"
#endif

template <int LA, int VA, int LB, int VB>
HD F28<1, 2> mul(const F28<LA, VA> &a, const F28<LB, VB> &b) {
    // a column holds <= 14 products a_i*b_j (< LA*LB*2^56) + 14 products q*p_j (< 2^56) + the carry-in
    static_assert(14 * LA * LB + 14 + 1 <= 255, "64-bit column accumulator would overflow");
    // result < a*b/2^392 + p; 2^392/p > 2520
    static_assert(VA * VB <= 2500, "Montgomery product would not be < 2p");
#ifdef CKZG_F28_USE_ASM_BLOCKS
    {
        F28<1, 2> ra;
        f28asm_mul(ra.l, a.l, b.l);
        return ra;
    }
#endif
    uint32_t q[14];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad64(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad64c(acc, q[i], FP28_P[k - i]);
        q[k] = ((uint32_t)acc * (uint32_t)FP28_NINV) & M28;
        mad64c(acc, q[k], FP28_P[0]);
        acc >>= 28;
    }
    F28<1, 2> r;
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) mad64(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 13; i < 14; i++) mad64c(acc, q[i], FP28_P[k - i]);
        r.l[k - 14] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}

// (a*b + c*d)/2^392 mod p with ONE Montgomery reduction: 588 multiply-adds instead of 784 for two
// products; both partial products of a column land in the accumulator before its quotient digit is fixed.
template <int LA, int VA, int LB, int VB, int LC, int VC, int LD, int VD>
HD F28<1, 2> mul_add2(const F28<LA, VA> &a, const F28<LB, VB> &b, const F28<LC, VC> &c, const F28<LD, VD> &d) {
    static_assert(14 * (LA * LB + LC * LD) + 14 + 1 <= 255, "64-bit column accumulator would overflow");
    static_assert(VA * VB + VC * VD <= 2500, "Montgomery result would not be < 2p");
#ifdef CKZG_F28_USE_ASM_BLOCKS
    {
        F28<1, 2> ra;
        f28asm_mul_add2(ra.l, a.l, b.l, c.l, d.l);
        return ra;
    }
#endif
    uint32_t q[14];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad64(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i <= k; i++) mad64(acc, c.l[i], d.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad64c(acc, q[i], FP28_P[k - i]);
        q[k] = ((uint32_t)acc * (uint32_t)FP28_NINV) & M28;
        mad64c(acc, q[k], FP28_P[0]);
        acc >>= 28;
    }
    F28<1, 2> r;
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) mad64(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 13; i < 14; i++) mad64(acc, c.l[i], d.l[k - i]);
#pragma unroll
        for (int i = k - 13; i < 14; i++) mad64c(acc, q[i], FP28_P[k - i]);
        r.l[k - 14] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}

// Montgomery square: the 91 cross products are taken once against a doubled operand (105 multiply-adds
// for the product instead of 196), same column walk: 301 vs 392 mads.
template <int LA, int VA>
HD F28<1, 2> sqr(const F28<LA, VA> &a) {
    // column k holds <= 7 doubled cross terms (< 2*LA^2*2^56) + one square + 14 q*p terms + the carry-in
    static_assert(15 * LA * LA + 14 + 1 <= 255, "64-bit column accumulator would overflow");
    static_assert(2 * LA <= 15, "doubled limb would overflow 32 bits");
    static_assert(VA * VA <= 2500, "Montgomery product would not be < 2p");
    uint32_t d[14], q[14];
#pragma unroll
    for (int j = 0; j < 14; j++) d[j] = a.l[j] << 1;
#ifdef CKZG_F28_USE_ASM_BLOCKS
    {
        F28<1, 2> ra;
        f28asm_sqr(ra.l, a.l, d);
        return ra;
    }
#endif
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) mad64(acc, a.l[i], d[k - i]);
        if ((k & 1) == 0) mad64(acc, a.l[k / 2], a.l[k / 2]);
#pragma unroll
        for (int i = 0; i < k; i++) mad64c(acc, q[i], FP28_P[k - i]);
        q[k] = ((uint32_t)acc * (uint32_t)FP28_NINV) & M28;
        mad64c(acc, q[k], FP28_P[0]);
        acc >>= 28;
    }
    F28<1, 2> r;
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; 2 * i < k; i++) mad64(acc, a.l[i], d[k - i]);
        if ((k & 1) == 0) mad64(acc, a.l[k / 2], a.l[k / 2]);
#pragma unroll
        for (int i = k - 13; i < 14; i++) mad64c(acc, q[i], FP28_P[k - i]);
        r.l[k - 14] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}

template <int LA, int VA, int LB, int VB>
HD F28<LA + LB, VA + VB> add(const F28<LA, VA> &a, const F28<LB, VB> &b) {
    static_assert(LA + LB <= 15, "limb would overflow 32 bits");
    F28<LA + LB, VA + VB> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = a.l[j] + b.l[j];
    return r;
}

// a - b + K*p with K the smallest power of two > VB (so the result is positive as an integer)
template <int LA, int VA, int LB, int VB>
HD F28<LA + LB + 2, VA + f28detail::pow2_above(VB)> sub(const F28<LA, VA> &a, const F28<LB, VB> &b) {
    constexpr int K = f28detail::pow2_above(VB);
    constexpr int C = LB + 1;
    static_assert(LA + LB + 2 <= 15, "limb would overflow 32 bits");
    static_assert(K <= 64, "value bound out of range");
    constexpr f28detail::Limbs14 m = f28detail::spread_multiple(K, C);
    // top limb: b < VB*p  =>  b_13 <= VB*(p >> 364) + 1 <= m_13 because K > VB and p>>364 >> C
    static_assert((uint64_t)(K - VB) * (FP28_P[13]) > (uint64_t)C + (uint64_t)VB + 2, "top limb could go negative");
    F28<LA + LB + 2, VA + K> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = a.l[j] + m.v[j] - b.l[j];
    return r;
}

// The same with the multiple K of p chosen by the caller (any K > VB): a tighter value bound than the
// power of two when the result feeds a squaring.
template <int K, int LA, int VA, int LB, int VB>
HD F28<LA + LB + 2, VA + K> sub_k(const F28<LA, VA> &a, const F28<LB, VB> &b) {
    constexpr int C = LB + 1;
    static_assert(LA + LB + 2 <= 15, "limb would overflow 32 bits");
    static_assert(K > VB && K <= 64, "value bound out of range");
    constexpr f28detail::Limbs14 m = f28detail::spread_multiple(K, C);
    static_assert((uint64_t)(K - VB) * (FP28_P[13]) > (uint64_t)C + (uint64_t)VB + 2, "top limb could go negative");
    F28<LA + LB + 2, VA + K> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = a.l[j] + m.v[j] - b.l[j];
    return r;
}

// carry-propagate: limbs back below 2^28 (the value, hence the top limb, is bounded by VB)
template <int LA, int VA>
HD F28<1, VA> norm(const F28<LA, VA> &a) {
    static_assert(VA <= 2500, "top limb bound");
    F28<1, VA> r;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) {
        uint32_t v = a.l[j] + c;
        r.l[j] = v & M28;
        c = v >> 28;
    }
    r.l[13] = a.l[13] + c;
    return r;
}

// value == 0 mod p, for a Montgomery product (normalised limbs, value < 2p: it is 0 or p)
HD bool is_zero(const F28<1, 2> &a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int j = 0; j < 14; j++) {
        z |= a.l[j];
        e |= a.l[j] ^ FP28_P[j];
    }
    return z == 0 || e == 0;
}

template <int L, int V>
HD F28<L, V> f28_const(const uint32_t (&c)[14]) {
    F28<L, V> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = c[j];
    return r;
}

HD F28<1, 1> f28_one() { return f28_const<1, 1>(FP28_ONE); }

// reinterpret with looser (larger) bounds
template <int L2, int V2, int L, int V>
HD F28<L2, V2> widen(const F28<L, V> &a) {
    static_assert(L2 >= L && V2 >= V, "bounds can only be loosened");
    F28<L2, V2> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = a.l[j];
    return r;
}

// neg ? -a : a for a fully reduced a (0 <= a < p), selected per lane.  The negation goes through
// the bounds-checked sub() (2p - a with spread limbs): a hand-rolled "p - a" would underflow the
// top limb whenever a's top limb equals p's.
HD F28<4, 2> cneg_reduced(const F28<1, 1> &a, bool neg) {
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    F28<4, 2> n = sub(zero, a);
    F28<4, 2> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = neg ? n.l[j] : a.l[j];
    return r;
}

// 12 packed 32-bit words (any integer < 2^384) -> 14 limbs of 28 bits
template <int V>
HD F28<1, V> f28_unpack(const uint32_t *w) {
    F28<1, V> r;
#pragma unroll
    for (int j = 0; j < 14; j++) {
        int bit = 28 * j, k = bit >> 5, sh = bit & 31;
        uint32_t lo = w[k] >> sh;
        if (sh > 4 && k + 1 < 12) lo |= w[k + 1] << (32 - sh);
        r.l[j] = (j == 13) ? lo : (lo & M28);
    }
    return r;
}

// inverse of f28_unpack; the value must be < 2^384 and the limbs normalised
template <int V>
HD void f28_pack(uint32_t *w, const F28<1, V> &a) {
    static_assert(V <= 9, "value would not fit 384 bits");
#pragma unroll
    for (int k = 0; k < 12; k++) {
        int bit = 32 * k, j = bit / 28, sh = bit - 28 * j;
        uint32_t v = a.l[j] >> sh;
        v |= a.l[j + 1] << (28 - sh);
        if (28 - sh + 28 < 32 && j + 2 < 14) v |= a.l[j + 2] << (56 - sh);
        w[k] = v;
    }
}

// Conversions to/from the host representation (Fp: 12 x u32, Montgomery radix 2^384, < p)
HD F28<1, 2> f28_from_fp(const Fp &x) {
    return mul(f28_unpack<1>(x.l), f28_const<1, 1>(FP28_FROM384));
}

// the tail of f28_to_fp: a value < 2p that already carries the 2^384-domain factor -> [0, p), packed
HD Fp f28_finish_fp(const F28<1, 2> &t);

// full reduction to [0, p) in the 2^384 domain
template <int LA, int VA>
HD Fp f28_to_fp(const F28<LA, VA> &a) {
    return f28_finish_fp(mul(a, f28_const<1, 1>(FP28_TO384)));
}

HD Fp f28_finish_fp(const F28<1, 2> &t) {
    // conditional subtraction of p in 28-bit limbs
    uint32_t s[14];
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 14; j++) {
        uint32_t d = t.l[j] - FP28_P[j] - br;
        br = d >> 31;  // limbs are < 2^28 (top < 2^19): a negative difference sets bit 31
        s[j] = (j == 13) ? d : (d & M28);
    }
    F28<1, 1> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = br ? t.l[j] : s[j];
    Fp out;
    f28_pack<1>(out.l, r);
    return out;
}

}  // namespace ckzg
