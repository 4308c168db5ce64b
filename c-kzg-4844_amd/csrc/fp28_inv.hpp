// fp28_inv.hpp -- modular inversion in Fp by the Bernstein-Yang "safegcd" divstep iteration
// (variable-time form; nothing here is secret).  A Fermat ladder costs 381 dependent squarings
// (~2*10^5 instructions per lane); 30 divsteps at a time on the low words of (f, g), followed by
// one 2x2-matrix update of the full-width (f, g) and of the cofactors (d, e), costs ~600
// instructions per batch and at most ~40 batches.
//
//   divstep(eta, f, g):  g odd and eta < 0 -> (-eta - 1, g, (g - f)/2)
//                        g odd             -> ( eta - 1, f, (g + f)/2)
//                        g even            -> ( eta - 1, f,  g/2)
// Invariant: f = d*x, g = e*x (mod p) for the input x; when g reaches 0, f = +-1 and d = +-1/x.
// Limbs: 13 signed words of 30 bits; 64-bit signed accumulators (v_mad_i64_i32).  d and e are kept
// exactly divisible by 2^30 at every update by adding a multiple of p, so no power of two is left
// over; each update lets |d|, |e| grow by at most p, which 13 x 30 bits absorb for the <= 40
// batches the iteration can take (768 divsteps suffice for 381-bit inputs; 40*30 = 1200).
#pragma once
#include "fp28.hpp"

namespace ckzg {

struct DivstepMatrix {
    int32_t u, v, q, r;
};

// A limb stored as (int32_t)(c & M) is, to the compiler, still the 64-bit value it was cut from, and its next product
// with a matrix entry becomes a 64 x 64-bit multiplication (v_mad_u64_u32 + two v_mul_lo_u32 + a v_add3_u32 of sign
// corrections).  Passing the limb through an empty asm makes it a plain 32-bit register again, and
// (int64_t)a * b is ONE v_mad_i64_i32: 661 -> ~230 instructions per update of (f, g) and (d, e).
HD int32_t limb32(int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#endif
    return x;
}

// 30 divsteps on the low 30 bits of f and g; returns the new eta.
// Variable-time form (nothing here is secret): a run of zero bits of g is ONE step (count trailing zeros, shift g and
// the matrix column by the run), and an odd g has its bottom min(eta + 1, bits left, 6) bits cancelled at once by the
// multiple w = -g/f mod 2^6 of f -- for odd f, f (f^2 - 2) = -1/f mod 64 -- which is what that many consecutive
// divsteps would add up to.  A batch of 30 divsteps is ~6 trips round this loop instead of 30 round the bit-by-bit
// one: the inversion that closes every commitment, every proof batch and every step of the table builder drops from
// ~31,000 to ~13,000 instructions of one lane.
HD int32_t divsteps30(int32_t eta, uint32_t f, uint32_t g, DivstepMatrix &t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    int left = 30;
    for (;;) {
        // zeros of g, counted only up to the bits that are left (sentinel above them)
        const uint32_t sentinel = g | (0xffffffffu << left);
#if defined(__HIP_DEVICE_COMPILE__)
        const int zeros = (int)__builtin_ctz(sentinel);
#else
        const int zeros = __builtin_ctz(sentinel);
#endif
        g >>= zeros;
        u <<= zeros;
        v <<= zeros;
        eta -= zeros;
        left -= zeros;
        if (left == 0) break;
        // g is odd
        if (eta < 0) {
            eta = -eta;
            uint32_t tmp = f;
            f = g;
            g = 0u - tmp;
            tmp = u;
            u = q;
            q = 0u - tmp;
            tmp = v;
            v = r;
            r = 0u - tmp;
        }
        // no more bits than are left, and no more than eta + 1 (after that many the sign of eta flips)
        int limit = eta + 1 > left ? left : eta + 1;
        if (limit > 6) limit = 6;
        const uint32_t m = (1u << limit) - 1u;
        const uint32_t w = (g * f * (f * f - 2u)) & m;   // -g/f mod 2^limit
        g += f * w;
        q += u * w;
        r += v * w;
    }
    t.u = limb32((int32_t)u);
    t.v = limb32((int32_t)v);
    t.q = limb32((int32_t)q);
    t.r = limb32((int32_t)r);
    return eta;
}

// (f, g) <- t * (f, g) / 2^30, exact
HD void update_fg30(int32_t *f, int32_t *g, const DivstepMatrix &t) {
    const int64_t M = (1 << 30) - 1;
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 13; i++) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = limb32((int32_t)(cf & M));
        g[i - 1] = limb32((int32_t)(cg & M));
        cf >>= 30;
        cg >>= 30;
    }
    f[12] = limb32((int32_t)cf);
    g[12] = limb32((int32_t)cg);
}

// (d, e) <- t * (d, e) / 2^30 mod p: a multiple of p makes the low 30 bits vanish first
HD void update_de30(int32_t *d, int32_t *e, const DivstepMatrix &t) {
    const int64_t M = (1 << 30) - 1;
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    const int32_t md = (int32_t)(((0u - (uint32_t)cd) * (uint32_t)FP30_PINV) & (uint32_t)M);
    const int32_t me = (int32_t)(((0u - (uint32_t)ce) * (uint32_t)FP30_PINV) & (uint32_t)M);
    cd += (int64_t)FP30_P[0] * md;
    ce += (int64_t)FP30_P[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 13; i++) {
        cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)FP30_P[i] * md;
        ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)FP30_P[i] * me;
        d[i - 1] = limb32((int32_t)(cd & M));
        e[i - 1] = limb32((int32_t)(ce & M));
        cd >>= 30;
        ce >>= 30;
    }
    d[12] = limb32((int32_t)cd);
    e[12] = limb32((int32_t)ce);
}

// 1/a in the 2^392 Montgomery domain; 0 for a == 0 (mod p), like the Fermat ladder
HDNI inline F28<1, 2> f28_inv_safegcd(const F28<1, 2> &a) {
    if (is_zero(a)) {
        F28<1, 2> z;
        for (int j = 0; j < 14; j++) z.l[j] = 0;
        return z;
    }
    int32_t f[13], g[13], d[13], e[13];
    // 14 x 28-bit limbs (value < 2p < 2^382) -> 13 x 30-bit limbs
    for (int i = 0; i < 13; i++) {
        int bit = 30 * i, j = bit / 28, sh = bit - 28 * j;
        uint32_t v = a.l[j] >> sh;
        if (j + 1 < 14) v |= a.l[j + 1] << (28 - sh);
        if (28 - sh + 28 < 30 && j + 2 < 14) v |= a.l[j + 2] << (56 - sh);
        g[i] = (int32_t)(v & 0x3fffffffu);
        f[i] = FP30_P[i];
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t eta = -1;
    for (int it = 0; it < 40; it++) {
        DivstepMatrix t;
        eta = divsteps30(eta, (uint32_t)f[0], (uint32_t)g[0], t);
        update_de30(d, e, t);
        update_fg30(f, g, t);
        int32_t nz = 0;
        for (int i = 0; i < 13; i++) nz |= g[i];
        if (nz == 0) break;
    }
    // f = +1 or -1; result = f * d, made positive by adding 64p (|d| < 41p)
    const bool negate = f[12] < 0;
    int64_t c = 0;
    uint32_t w[13];
    for (int i = 0; i < 13; i++) {
        c += (int64_t)FP30_64P[i] + (negate ? -(int64_t)d[i] : (int64_t)d[i]);
        w[i] = (uint32_t)(c & 0x3fffffff);
        c >>= 30;
    }
    w[12] += (uint32_t)(c << 30);  // value < 105p < 2^388: the top word holds what is left
    // 13 x 30 -> 14 x 28
    F28<1, 128> y;
    for (int j = 0; j < 14; j++) {
        int bit = 28 * j, i = bit / 30, sh = bit - 30 * i;
        uint32_t v = w[i] >> sh;
        if (i + 1 < 13 && 30 - sh < 28) v |= w[i + 1] << (30 - sh);
        y.l[j] = (j == 13) ? v : (v & M28);
    }
    // y = 1/(a R) as an integer; Montgomery-multiply by R^3 to land on (1/a) R
    return mul(y, f28_const<1, 1>(FP28_R3));
}

}  // namespace ckzg
