// g1_pipe.hpp -- the twiddle multiplication [k]P of the small-batch G1 FFT as a pipeline of waves (device only).
//
// g1_quad.hpp's left-to-right ladder (xyzz28_mul_glv_naf_quad) is one dependent chain per point: table (15 product
// steps), then per scalar bit a doubling (3 steps) and, at ~2/5 of the bits, a mixed addition (4 steps) INTO THE SAME
// accumulator: ~610 dependent steps for the two 128-bit GLV halves.  A lone wave is already issue-bound on its
// multiply-adds, so the only way to shorten the chain is to take work off it.  Right-to-left evaluation does that:
//     B_i = 2^i P                      (128 doublings, 3 steps each: the part no algorithm can avoid)
//     R1 = sum d1_i B_i,  R2 = sum d2_i B_i      (plain NAF digits +-1 of k1, k2; ~43 additions each)
//     [k]P = R1 + phi(R2),  phi(X, Y, Z) = (beta X, Y, Z)
// The additions no longer touch the doubling chain, so they run on OTHER waves of the same workgroup: the doubler
// publishes B_i through an LDS ring at the bits where a digit is non-zero (the digit strings are those of ONE twiddle
// per workgroup, known to every wave), one adder wave per chain consumes them.  What is left after the last doubling
// is one addition, the hand-over of R2 and R1 + phi(R2): ~400 dependent steps instead of ~610.  All waves keep
// g1_quad.hpp's four-lanes-per-point form (16 points per wave, replicated state).
// Three forms, by what a step's workgroups find on the chip (measured: profiles/r05_fk20_small_ab.txt; where waves land:
// tools/ubench/wave_placement.hip -- the waves of one workgroup never share a SIMD, workgroups go one per compute unit
// until the 256 are taken, a second workgroup on a unit does not pick the idle SIMDs):
//   * three waves, TWO twiddles per workgroup (<= 8 transforms): 168 workgroups per radix-8 step, a compute unit each;
//     the step runs at the doubler's pace (~0.55 ms).
//   * two waves -- one adder for both chains, its two sums parked in LDS between their additions -- in a kernel with
//     the whole register file per wave (9..16 transforms, 336 workgroups, two per unit, a SIMD per wave by
//     construction).  With the usual 256 registers the adder spilled, its additions took ~11 us instead of ~7 and it
//     bounded the step.
//   * three waves, one twiddle (the first form built; 336 workgroups leave units with two workgroups, where a doubler
//     shares its SIMD with somebody's adder: 0.74 ms per step) -- kept for the A/B builds.
//
// Infinity: a quad whose input is the point at infinity runs the doubler on zeros and sits out the additions
// (predicated), so it costs nothing and cannot reach the exceptional-case fallback.
#pragma once
#include "g1_quad.hpp"

namespace ckzg {
namespace quad {

constexpr int NAF2_LEN = 130;     // plain NAF of a value < 2^129
constexpr int PIPE_SLOTS = 16;    // ring entries (events of BOTH chains); the doubler waits when an adder is this far behind

// Plain (width-2) non-adjacent form of a 128-bit k: digits in {0, +-1}, no two adjacent non-zero, density 1/3.
HDNI inline void naf2_128(int8_t *out, const uint32_t *k) {
    uint32_t v[5] = {k[0], k[1], k[2], k[3], 0};
    for (int i = 0; i < NAF2_LEN; i++) {
        int d = 0;
        if (v[0] & 1u) {
            d = 2 - (int)(v[0] & 3u);   // 1 -> +1, 3 -> -1
            if (d > 0) {
                v[0] &= ~1u;            // v -= 1 (v is odd)
            } else {
                uint64_t c = 1;         // v += 1
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

// LDS of one pipeline workgroup: the ring holds B_i as Jacobian X | Y | Z | Z^2 (14 limbs each) per quad; word w of a
// quad's record lives at [w >> 2][quad][w & 3], so the adder's 16-byte reads are conflict-free (the four lanes of a
// quad read one address, the 16 quads consecutive ones).
struct PipeShared {
    uint32_t ring[PIPE_SLOTS][14][16][4];
    uint32_t r2[14][16][4];   // the k2 chain's sum, handed to the k1 chain's wave for R1 + phi(R2)
    uint32_t r1[14][16][4];   // two-wave form only: the k1 chain's sum lives here between its additions (r2 likewise)
    uint32_t r2inf[16];
    uint32_t produced;        // events published by the doubler
    uint32_t consumed[2];     // per adder wave: it needs no event below this index (signed comparison: may run ahead)
    uint32_t r2done;
    uint32_t pinf[16];        // input at infinity, per quad (the doubler knows; the adders need it)
};

// The digit strings of one twiddle as bit masks in scalar registers: a per-bit byte load from memory costs a wave
// ~0.2 us of latency, 2 x 130 times per wave.
struct NafMasks {
    uint64_t nz[2][3], neg[2][3];   // [chain][word]: digit non-zero / negative at bit i
    int top;                        // highest bit with a non-zero digit in either chain (-1: none)
};
// (every index into the mask arrays is a compile-time constant after unrolling: a run-time index -- or a reference
// selected at run time -- keeps the struct in memory, and the compiler then parks one copy PER THREAD in LDS: +20 KB)
__device__ __forceinline__ NafMasks naf_masks(const int8_t *naf1, const int8_t *naf2) {
    NafMasks m;
    m.top = -1;
#pragma unroll
    for (int w = 0; w < 3; w++) {
        uint64_t nz1 = 0, ng1 = 0, nz2 = 0, ng2 = 0;
        for (int b = 0; b < 64; b++) {
            const int i = 64 * w + b;
            if (i >= NAF2_LEN) break;
            const int d1 = naf1[i], d2 = naf2[i];
            const uint64_t bit = 1ull << b;
            if (d1) nz1 |= bit;
            if (d1 < 0) ng1 |= bit;
            if (d2) nz2 |= bit;
            if (d2 < 0) ng2 |= bit;
            if (d1 | d2) m.top = i;
        }
        m.nz[0][w] = nz1;
        m.neg[0][w] = ng1;
        m.nz[1][w] = nz2;
        m.neg[1][w] = ng2;
    }
    return m;
}
__device__ __forceinline__ NafMasks naf_masks_none() {
    NafMasks m;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int w = 0; w < 3; w++) m.nz[c][w] = m.neg[c][w] = 0;
    m.top = -1;
    return m;
}
__device__ __forceinline__ bool mask_bit(const uint64_t (&w)[3], int i) {
    const uint64_t word = i < 64 ? w[0] : (i < 128 ? w[1] : w[2]);
    return ((word >> (i & 63)) & 1ull) != 0;
}
// digit of chain c (a run-time value, uniform over the wave) at bit i non-zero / negative
__device__ __forceinline__ bool naf_nz(const NafMasks &m, int c, int i) { return c ? mask_bit(m.nz[1], i) : mask_bit(m.nz[0], i); }
__device__ __forceinline__ bool naf_neg(const NafMasks &m, int c, int i) { return c ? mask_bit(m.neg[1], i) : mask_bit(m.neg[0], i); }

__device__ __forceinline__ uint32_t pipe_load(uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pipe_store(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// A wait between the waves of ONE workgroup.  They are resident together, so the wait ends unless the protocol is
// wrong -- and if it is, the wave must not occupy the device for ever (round-5 advisor finding; the host side of the
// library has had no unbounded wait since round 6 either: device.hpp).  The longest legitimate wait is one doubling
// chain of the partner wave, tens of microseconds; after 2^24 looks (seconds) the wave traps and the launch fails
// (hipErrorLaunchFailure -> C_KZG_ERROR at the call).  The flag words are read by every lane from one address:
// readfirstlane makes the condition -- and with it the look counter -- scalar, so the bound costs no vector register.
constexpr uint32_t PIPE_SPIN_LIMIT = 1u << 24;
__device__ __forceinline__ uint32_t pipe_load_uniform(uint32_t *p) { return __builtin_amdgcn_readfirstlane(pipe_load(p)); }
template <class Ready>
__device__ __forceinline__ void pipe_wait(Ready &&ready) {
    uint32_t looks = 0;
    while (!ready()) {
        __builtin_amdgcn_s_sleep(1);
        if (++looks > PIPE_SPIN_LIMIT) __builtin_trap();
    }
}

// ---- points between the launches of a small-batch transform: RAW records ----
// 56 limbs of the 28-bit domain (x, y, zz, zzz of an XYZZ28) and the infinity flag, as they stand in registers.  A G1XYZZ
// (12 x 32-bit limbs, 2^384 domain) costs four Montgomery products to read and four plus a final reduction to write --
// per TERM of the sums around the ladders, that was a third of a radix-8 step.
constexpr int RAW_WORDS = 57;

__device__ __forceinline__ XYZZ28 raw_load(const uint32_t *p, bool &inf) {
    XYZZ28 r;
    uint32_t *d = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (int k = 0; k < 56; k++) d[k] = p[k];
    inf = p[56] != 0;
    return r;
}
__device__ __forceinline__ void raw_store(uint32_t *p, const XYZZ28 &v, bool inf) {
    const uint32_t *s = reinterpret_cast<const uint32_t *>(&v);
#pragma unroll
    for (int k = 0; k < 56; k++) p[k] = s[k];
    p[56] = inf ? 1u : 0u;
}

// a <- a + b or a - b: xyzz28_add_quad with the sign folded into the operand selection (xyzz28_neg costs a product)
__device__ __forceinline__ void xyzz28_addsub_quad(XYZZ28 &a, bool &ainf, const XYZZ28 &b, bool binf, bool neg, int ql) {
    if (binf) return;
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    F28<1, 10> by;
    {
        const auto ny = norm(sub(zero, b.y));   // <3,8> -> <1,8>
#pragma unroll
        for (int j = 0; j < 14; j++) by.l[j] = neg ? ny.l[j] : b.y.l[j];
    }
    if (ainf) {
        a = b;
        a.y = widen<1, 6>(mul(by, f28_one()));
        ainf = false;
        return;
    }
    // step 1: X1*ZZ2 | X2*ZZ1 | Y1*ZZZ2 | Y2*ZZZ1
    const auto p1 = mul(qsel(ql, a.x, b.x, widen<1, 10>(a.y), by), qsel(ql, b.zz, a.zz, b.zzz, a.zzz));  // 14+15 ok; 20 ok
    const auto u1 = qread<0>(p1), u2 = qread<1>(p1), s1 = qread<2>(p1), s2 = qread<3>(p1);
    const auto p = sub(u2, u1);   // <4,6>
    const auto r = sub(s2, s1);   // <4,6>
    // step 2: P*P | ZZ1*ZZ2 | ZZZ1*ZZZ2 | R*R
    const auto zz1 = widen<4, 6>(a.zz), zz2 = widen<4, 6>(b.zz), zzz1 = widen<4, 6>(a.zzz), zzz2 = widen<4, 6>(b.zzz);
    const auto p2 = mul(qsel(ql, p, zz1, zzz1, r), qsel(ql, p, zz2, zzz2, r));   // 14*16+15 = 239 ok; 36 ok
    const auto pp = qread<0>(p2), zzab = qread<1>(p2), zzzab = qread<2>(p2), rr = qread<3>(p2);
    if (is_zero(pp)) {  // same x: the complete one-lane routine, on every copy
        xyzz28_add(a, ainf, neg ? xyzz28_neg(b) : b, binf);
        return;
    }
    // step 3: P*PP | U1*PP | ZZ1ZZ2*PP
    const auto p3 = mul(qsel(ql, p, widen<4, 6>(u1), widen<4, 6>(zzab), p), pp);   // 14*4+15 ok; 12 ok
    const auto ppp = qread<0>(p3), q = qread<1>(p3), zz3 = qread<2>(p3);
    const auto x3 = norm(sub(rr, add(ppp, add(q, q))));   // <6,10> -> <1,10>
    const auto d = sub(q, x3);                            // <4,18>
    const auto s1n = sub(zero, s1);                       // <4,4> = -S1
    // step 4: R*(Q - X3) | (-S1)*PPP | ZZZ1ZZZ2*PPP ; Y3 is the sum of the first two
    const auto rn = widen<4, 6>(norm(r)), sn = widen<4, 6>(s1n), zn = widen<4, 6>(zzzab);
    const auto ppp18 = widen<4, 18>(ppp);
    const auto p4 = mul(qsel(ql, rn, sn, zn, rn), qsel(ql, d, ppp18, ppp18, d));   // 239 ok; 108 ok
    const auto y3 = norm(add(qread<0>(p4), qread<1>(p4)));   // <2,4> -> <1,4>
    a.x = x3;
    a.y = widen<1, 6>(y3);
    a.zz = zz3;
    a.zzz = qread<2>(p4);
}

// A wave may serve TWO twiddles -- quads 0..7 the ladders of mA, quads 8..15 those of mB (batches of <= 8 transforms:
// the workgroups of a radix-8 step halve, 336 -> 168, and every three-wave workgroup finds a compute unit to itself).
// The doubling chain does not depend on the scalar at all; it publishes at the bits where ANY of the four digit strings
// is non-zero, and an adder wave's quads take part in an addition only where their own twiddle's digit is non-zero
// (the addition then runs with part of the wave masked off: ~5/9 of the bits per chain instead of 1/3, still well
// under the doubling chain's time).  One twiddle per wave: mA and mB are the same masks.
//
// The doubler wave.  p: the input (replicated in the quad), quad_id 0..15, ql 0..3.
__device__ __forceinline__ void pipe_doubler(PipeShared &sh, const XYZZ28 &p, bool p_inf, const NafMasks &mA, const NafMasks &mB,
                                          int quad_id, int ql, bool one_adder = false) {
    if (ql == 0) sh.pinf[quad_id] = p_inf ? 1u : 0u;
    JAC28 b = jac28_from_xyzz(p);
    F28<1, 2> zz = sqr(p.zz);   // jac28_from_xyzz takes Z = ZZ(p)
    uint32_t ev = 0;
    const int top = mA.top > mB.top ? mA.top : mB.top;
    for (int i = 0; i <= top; i++) {
        if (mask_bit(mA.nz[0], i) || mask_bit(mA.nz[1], i) || mask_bit(mB.nz[0], i) || mask_bit(mB.nz[1], i)) {   // uniform over the wave
            pipe_wait([&]() {   // room in the ring: both adders are past event ev - PIPE_SLOTS
                const int c0 = (int)pipe_load_uniform(&sh.consumed[0]), c1 = one_adder ? c0 : (int)pipe_load_uniform(&sh.consumed[1]);
                return (int)ev - (c0 < c1 ? c0 : c1) < PIPE_SLOTS;
            });
            // lane ql stores element ql of the record: X | Y | Z | Z^2
            const auto e = qsel(ql, widen<2, 34>(b.x), widen<2, 34>(b.y), widen<2, 34>(b.z), widen<2, 34>(zz));
            uint32_t(*slot)[16][4] = sh.ring[ev % PIPE_SLOTS];
#pragma unroll
            for (int j = 0; j < 14; j++) {
                const int w = ql * 14 + j;   // (the element's base is a runtime value: plain 4-byte stores)
                slot[w >> 2][quad_id][w & 3] = e.l[j];
            }
            ev++;
            if ((threadIdx.x & 63) == 0) pipe_store(&sh.produced, ev);
        }
        if (i < top) jac28_dbl_quad_zz(b, zz, ql);
    }
}

// a <- a + (+-b) for b = (bx, by, bz) Jacobian with bzz = bz^2; zz = Z(a)^2 in and out.  Four product steps (b's Z^3
// is made in the first).  The partial sums of a NAF never meet their next term (|sum| < 2^i), so the exceptional case
// is for the final R1 + phi(R2) only; it takes the complete one-lane routine on every copy.
__device__ __forceinline__ void jac28_add_quad_pipe(JAC28 &a, F28<1, 2> &zz, bool &ainf, const F28<1, 34> &bx,
                                                    const F28<1, 34> &by_in, const F28<2, 4> &bz, const F28<1, 2> &bzz, bool neg,
                                                    int ql) {
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    F28<1, 64> by;
    {
        const auto ny = norm(sub(zero, by_in));   // <3,64> -> <1,64>
#pragma unroll
        for (int j = 0; j < 14; j++) by.l[j] = neg ? ny.l[j] : by_in.l[j];
    }
    if (ainf) {
        a.x = bx;
        a.y = widen<1, 34>(mul(by, f28_one()));
        a.z = bz;
        zz = bzz;
        ainf = false;
        return;
    }
    // step 1: X1*ZZ2 | X2*ZZ1 | Z2*ZZ2 | Z1*ZZ1
    const auto p1 = mul(qsel(ql, widen<2, 34>(a.x), widen<2, 34>(bx), widen<2, 34>(bz), widen<2, 34>(a.z)),
                        qsel(ql, bzz, zz, bzz, zz));                                  // 14*2+15 ok; 34*2 ok
    const auto u1 = qread<0>(p1), u2 = qread<1>(p1), zzz2 = qread<2>(p1), zzz1 = qread<3>(p1);
    const auto h = sub(u2, u1);                                                       // <4,6>
    // step 2: Y1*ZZZ2 | Y2*ZZZ1 | Z1*Z2 | H*H
    const auto h64 = widen<4, 64>(h);
    const auto p2 = mul(qsel(ql, widen<4, 64>(a.y), widen<4, 64>(by), widen<4, 64>(a.z), h64),
                        qsel(ql, widen<4, 6>(zzz2), widen<4, 6>(zzz1), widen<4, 6>(bz), h));   // 14*16+15 = 239 ok; 64*6 ok
    const auto s1 = qread<0>(p2), s2 = qread<1>(p2), z1z2 = qread<2>(p2), hh = qread<3>(p2);
    if (is_zero(hh)) {   // same x: the complete one-lane routine, on every copy
        JACT28 t;
        t.x = bx;
        t.y = by;
        t.z = bz;
        t.zz = bzz;
        t.zzz = zzz2;
        jac28_add(a, ainf, t);
        if (!ainf) zz = sqr(a.z);
        return;
    }
    const auto r = sub(s2, s1);                                                       // <4,6>
    // step 3: H*HH | U1*HH | R*R | Z1Z2*H
    const auto hh46 = widen<4, 6>(hh);
    const auto p3 = mul(qsel(ql, h, widen<4, 6>(u1), r, widen<4, 6>(z1z2)), qsel(ql, hh46, hh46, r, h));   // 239 ok; 36 ok
    const auto hhh = qread<0>(p3), v = qread<1>(p3), rr = qread<2>(p3), z3 = qread<3>(p3);
    const auto x3 = norm(sub(rr, add(hhh, add(v, v))));                               // <6,10> -> <1,10>
    const auto dv = sub(v, x3);                                                       // <4,18>
    const auto s1n = sub(zero, s1);                                                   // <4,4> = -S1
    // step 4: R*(V - X3) | (-S1)*HHH | Z3*Z3 | (Z3*Z3)
    const auto rn = widen<4, 6>(norm(r)), sn = widen<4, 6>(s1n), z3l = widen<4, 6>(z3);
    const auto h18 = widen<4, 18>(hhh), z3r = widen<4, 18>(z3);
    const auto p4 = mul(qsel(ql, rn, sn, z3l, z3l), qsel(ql, dv, h18, z3r, z3r));     // 239 ok; 108 ok
    const auto y3 = norm(add(qread<0>(p4), qread<1>(p4)));                            // <2,4> -> <1,4>
    a.x = widen<1, 34>(x3);
    a.y = widen<1, 34>(y3);
    a.z = widen<2, 4>(z3);
    zz = qread<2>(p4);
}

// An adder wave.  chain 0 sums the k1 terms, receives the k2 chain's sum and returns [k1]P + phi([k2]P) (replicated in
// the quad); chain 1 sums the k2 terms and hands them over (out is not written).  The hand-over is the last pass of
// the SAME loop, so that the addition has one call site: it is ~16 KB of code (two copies and the doubler's loop no
// longer share the instruction cache), and as a called function its operands went through scratch memory.
__device__ __forceinline__ void pipe_adder(XYZZ28 &out, bool &out_inf, PipeShared &sh, const NafMasks &mA, const NafMasks &mB, int chain,
                                        int quad_id, int ql) {
    JAC28 r;
    F28<1, 2> zz;
    bool inf = true;
    uint32_t ev = 0;
    const int top = mA.top > mB.top ? mA.top : mB.top;
    bool p_inf = top < 0, have_flag = top < 0;   // k = 0: infinity whatever the input
    const bool leader = (threadIdx.x & 63) == 0;
    const bool second = quad_id >= 8;   // which twiddle this quad's ladder belongs to (the same masks twice: no difference)
    for (int i = 0; i <= top + 1; i++) {
        const bool closing = i == top + 1;
        const uint32_t(*rec)[16][4];
        bool mine = true;   // this quad's own digit is non-zero (per lane)
        if (!closing) {
            const bool own_a = naf_nz(mA, chain, i), own_b = naf_nz(mB, chain, i);
            const bool own = own_a || own_b;
            if (!(own || naf_nz(mA, 1 - chain, i) || naf_nz(mB, 1 - chain, i))) continue;   // uniform over the wave
            mine = second ? own_b : own_a;
            if (!own) {
                ev++;
                continue;
            }
            if (leader) pipe_store(&sh.consumed[chain], ev);   // nothing below this event is needed any more
            pipe_wait([&]() { return pipe_load_uniform(&sh.produced) > ev; });
            rec = sh.ring[ev % PIPE_SLOTS];
        } else {
            if (leader) pipe_store(&sh.consumed[chain], 0x3fffffffu);
            if (chain == 1) break;
            pipe_wait([&]() { return pipe_load_uniform(&sh.r2done) != 0; });
            rec = sh.r2;
        }
        if (!have_flag) {
            // (a chain without a term of its own gets here at the hand-over; the doubler has published by then)
            pipe_wait([&]() { return pipe_load_uniform(&sh.produced) != 0; });
            p_inf = sh.pinf[quad_id] != 0;
            have_flag = true;
        }
        F28<1, 34> bx, by;
        F28<2, 4> bz;
        F28<1, 2> bzz;
        {
            uint32_t w[56];
#pragma unroll
            for (int k4 = 0; k4 < 14; k4++) {
                const uint4 q = *reinterpret_cast<const uint4 *>(&rec[k4][quad_id][0]);
                w[4 * k4] = q.x; w[4 * k4 + 1] = q.y; w[4 * k4 + 2] = q.z; w[4 * k4 + 3] = q.w;
            }
#pragma unroll
            for (int j = 0; j < 14; j++) {
                bx.l[j] = w[j];
                by.l[j] = w[14 + j];
                bz.l[j] = w[28 + j];
                bzz.l[j] = w[42 + j];
            }
        }
        bool skip = p_inf || !mine, neg = false;   // per quad: a quad without a point (or without a digit here) sits it out
        if (!closing) {
            if (leader) pipe_store(&sh.consumed[chain], ev + 1);   // (the record is in registers)
            ev++;
            neg = second ? naf_neg(mB, chain, i) : naf_neg(mA, chain, i);
        } else {
            skip = skip || sh.r2inf[quad_id] != 0;
            bx = widen<1, 34>(mul(bx, f28_const<1, 1>(FP28_BETA_LAMBDA)));   // phi
        }
        if (!skip) jac28_add_quad_pipe(r, zz, inf, bx, by, bz, bzz, neg, ql);
    }
    if (chain == 1) {
        if (!have_flag) {   // the k2 chain had no term: its flag comes with the first event, which exists (top >= 0 here)
            pipe_wait([&]() { return pipe_load_uniform(&sh.produced) != 0; });
            p_inf = sh.pinf[quad_id] != 0;
        }
        // hand R2 over: X | Y | Z | Z^2, lane ql stores element ql
        const auto e = qsel(ql, widen<2, 34>(r.x), widen<2, 34>(r.y), widen<2, 34>(r.z), widen<2, 34>(zz));
#pragma unroll
        for (int j = 0; j < 14; j++) {
            const int w = ql * 14 + j;
            sh.r2[w >> 2][quad_id][w & 3] = e.l[j];
        }
        if (ql == 0) sh.r2inf[quad_id] = (inf || p_inf) ? 1u : 0u;
        if (leader) pipe_store(&sh.r2done, 1u);
        return;
    }
    const bool res_inf = p_inf || inf;
    if (!res_inf) out = jac28_to_xyzz(r);
    out_inf = res_inf;
}

// The TWO-wave form: one adder wave for both chains, with the two sums parked in LDS between their additions so that
// only one accumulator is in registers at a time (with both in registers the wave spilled and its 86 additions took
// 0.7-0.9 ms against the doubler's 0.53: profiles/r05_fk20_small_ab.txt).  Why it exists at all: workgroups of two
// waves always find a SIMD per wave -- two of them fit a compute unit -- where a radix-8 step of 9..16 transforms
// (336 three-wave workgroups on 256 units) leaves doublers sharing a SIMD with somebody's adder.  The adder is then
// about as long as the doubling chain (86 x ~5.8 us), so the step is bounded by whichever of the two stalls the other.
__device__ __forceinline__ void pipe_acc_store(uint32_t (*dst)[16][4], const JAC28 &a, const F28<1, 2> &zz, int quad_id, int ql) {
    const auto e = qsel(ql, widen<2, 34>(a.x), widen<2, 34>(a.y), widen<2, 34>(a.z), widen<2, 34>(zz));
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const int w = ql * 14 + j;
        dst[w >> 2][quad_id][w & 3] = e.l[j];
    }
}
__device__ __forceinline__ void pipe_rec_load(F28<1, 34> &x, F28<1, 34> &y, F28<2, 4> &z, F28<1, 2> &zz, const uint32_t (*src)[16][4],
                                              int quad_id) {
    uint32_t w[56];
#pragma unroll
    for (int k4 = 0; k4 < 14; k4++) {
        const uint4 q = *reinterpret_cast<const uint4 *>(&src[k4][quad_id][0]);
        w[4 * k4] = q.x; w[4 * k4 + 1] = q.y; w[4 * k4 + 2] = q.z; w[4 * k4 + 3] = q.w;
    }
#pragma unroll
    for (int j = 0; j < 14; j++) {
        x.l[j] = w[j];
        y.l[j] = w[14 + j];
        z.l[j] = w[28 + j];
        zz.l[j] = w[42 + j];
    }
}

__device__ __forceinline__ void pipe_adder_dual(XYZZ28 &out, bool &out_inf, PipeShared &sh, const NafMasks &m, int quad_id, int ql) {
    bool inf0 = true, inf1 = true;   // (per quad only through p_inf: one twiddle per wave, every chain starts at the same bit)
    uint32_t ev = 0;
    bool p_inf = m.top < 0;   // k = 0: infinity whatever the input
    const bool leader = (threadIdx.x & 63) == 0;
    JAC28 a;            // the accumulator in hand; after the closing pass: the result
    F28<1, 2> az;
    bool ai = true;
    for (int i = 0; i <= m.top + 1; i++) {
        const bool closing = i == m.top + 1;
        bool want0 = false, want1 = false;
        F28<1, 34> bx, by;
        F28<2, 4> bz;
        F28<1, 2> bzz;
        if (!closing) {
            want0 = mask_bit(m.nz[0], i);
            want1 = mask_bit(m.nz[1], i);
            if (!(want0 | want1)) continue;   // uniform over the wave
            pipe_wait([&]() { return pipe_load_uniform(&sh.produced) > ev; });
            if (ev == 0) p_inf = sh.pinf[quad_id] != 0;
            pipe_rec_load(bx, by, bz, bzz, sh.ring[ev % PIPE_SLOTS], quad_id);
            ev++;
            if (leader) pipe_store(&sh.consumed[0], ev);   // (the record is in registers)
        } else {
            // R1 += phi(R2), phi(X, Y, Z) = (beta X, Y, Z)
            want0 = true;
            if (!inf1) {
                pipe_rec_load(bx, by, bz, bzz, sh.r2, quad_id);
                bx = widen<1, 34>(mul(bx, f28_const<1, 1>(FP28_BETA_LAMBDA)));
            }
        }
#pragma unroll 1
        for (int c = 0; c < 2; c++) {
            if (!(c ? want1 : want0)) continue;
            const bool skip = p_inf || (closing && inf1);
            uint32_t(*acc)[16][4] = c ? sh.r2 : sh.r1;
            ai = c ? inf1 : inf0;
            if (!ai) pipe_rec_load(a.x, a.y, a.z, az, acc, quad_id);
            if (!skip) jac28_add_quad_pipe(a, az, ai, bx, by, bz, bzz, !closing && naf_neg(m, c, i), ql);
            if (!closing) {
                if (!skip) pipe_acc_store(acc, a, az, quad_id, ql);
                if (!p_inf) {
                    if (c) inf1 = ai;
                    else inf0 = ai;
                }
            }
        }
    }
    const bool res_inf = p_inf || ai;
    if (!res_inf) out = jac28_to_xyzz(a);
    out_inf = res_inf;
}

}  // namespace quad
}  // namespace ckzg
