// g1_quad.hpp -- Jacobian point arithmetic spread over the four lanes of a DPP quad (device only).
//
// Why: the scalar ladders of the verification paths and of the small-batch G1 FFT are *latency* work.  A lone
// wave issues one v_mad_u64_u32 every 9.4 cycles whatever its instruction-level parallelism
// (tools/ubench/mad_latency.hip), so 128 doublings + 46 additions of one lane are ~2 ms -- and at the sizes of
// those calls most SIMDs of the chip have no wave at all.  The remedy is lanes, not instructions: the 7 field
// products of a doubling have only 3 dependency levels and the 16 of an addition only 5, so four cooperating
// lanes per point shorten the sequential chain 2.3-3x and spread one point's work over four times as many
// waves.  All four lanes of a quad hold the SAME point (replicated state); in every step each lane multiplies
// the operand pair chosen for its position, then the products are handed round with DPP quad permutes
// (v_mov_b32_dpp quad_perm: register-to-register, no LDS).  A step costs one Montgomery product whatever the
// lane does, so the spare lane of a step simply repeats a neighbour's product.
//
// Formulas and value bounds are those of jac28_dbl / jac28_add (g1_28.hpp); the exceptional cases of the
// addition (equal or opposite points) fall back to the one-lane complete routine, which every lane of the quad
// then runs on its copy.
#pragma once
#include "g1_28.hpp"

namespace ckzg {
namespace quad {

// the 32-bit value held by lane SRC (0..3) of the calling lane's quad
template <int SRC>
__device__ __forceinline__ uint32_t qread(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, SRC * 0x55, 0xf, 0xf, true);
}
template <int SRC, int L, int V>
__device__ __forceinline__ F28<L, V> qread(const F28<L, V> &a) {
    F28<L, V> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = qread<SRC>(a.l[j]);
    return r;
}

// operand of this lane: v0..v3 for quad positions 0..3
template <int L, int V>
__device__ __forceinline__ F28<L, V> qsel(int ql, const F28<L, V> &v0, const F28<L, V> &v1, const F28<L, V> &v2,
                                          const F28<L, V> &v3) {
    F28<L, V> r;
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const uint32_t lo = (ql & 1) ? v1.l[j] : v0.l[j], hi = (ql & 1) ? v3.l[j] : v2.l[j];
        r.l[j] = (ql & 2) ? hi : lo;
    }
    return r;
}

// a <- 2a.  Three product steps instead of seven products.
__device__ __forceinline__ void jac28_dbl_quad(JAC28 &a, int ql) {
    // step 1: X*X | Y*Y | Y*Z | (X*X)
    {
        const auto x = widen<2, 34>(a.x), y = widen<2, 34>(a.y), z = widen<2, 34>(a.z);
        const auto p1 = mul(qsel(ql, x, y, y, x), qsel(ql, x, y, z, x));   // 14*4+15 ok; 34*34 ok
        const auto A = qread<0>(p1), B = qread<1>(p1), YZ = qread<2>(p1);
        // step 2: E*E | B*B | X*B | (B*B),  E = 3A
        const auto E = add(add(A, A), A);                                   // <3,6>
        const auto e = widen<3, 34>(E), b = widen<3, 34>(B), xx = widen<3, 34>(a.x);
        const auto p2 = mul(qsel(ql, e, b, xx, b), qsel(ql, e, b, b, b));   // 14*9+15 ok; 34*34 ok
        const auto F = qread<0>(p2), C = qread<1>(p2), XB = qread<2>(p2);
        const auto XB2 = add(XB, XB);
        const auto D = add(XB2, XB2);                                       // <4,8> = 4 X Y^2
        const auto X3 = norm(sub(F, add(D, D)));                            // <11,34> -> <1,34>
        const auto dx = norm(sub(D, X3));                                   // <1,72>
        const auto C2 = add(C, C);
        const auto C4 = add(C2, C2);
        const auto C8 = add(C4, C4);                                        // <8,16>
        // step 3: the same product in every lane
        const auto Y3 = norm(sub(mul(E, dx), C8));                          // <1,34>
        a.x = X3;
        a.y = Y3;
        a.z = add(YZ, YZ);                                                  // <2,4>
    }
}

// a <- a + b (b finite, with cached Z^2, Z^3).  Five product steps instead of sixteen products.
__device__ __forceinline__ void jac28_add_quad(JAC28 &a, bool &ainf, const JACT28 &b, int ql) {
    if (ainf) {
        a.x = b.x;
        a.y = widen<1, 34>(mul(b.y, f28_one()));
        a.z = b.z;
        ainf = false;
        return;
    }
    // step 1: Z1*Z1 | X1*ZZ2 | Y1*ZZZ2 | Z1*Z2
    const auto az = widen<2, 34>(a.z);
    const auto p1 = mul(qsel(ql, az, widen<2, 34>(a.x), widen<2, 34>(a.y), az),
                        qsel(ql, a.z, widen<2, 4>(b.zz), widen<2, 4>(b.zzz), b.z));   // 14*4+15 ok; 34*4 ok
    const auto z1z1 = qread<0>(p1), u1 = qread<1>(p1), s1 = qread<2>(p1), z1z2 = qread<3>(p1);
    // step 2: X2*Z1Z1 | Z1*Z1Z1
    const auto bx = widen<2, 34>(b.x);
    const auto p2 = mul(qsel(ql, bx, az, bx, az), z1z1);                              // 14*2+15 ok; 34*2 ok
    const auto u2 = qread<0>(p2), z1c = qread<1>(p2);
    const auto h = sub(u2, u1);                                                       // <4,6>
    // step 3: Y2*Z1^3 | H*H | Z1Z2*H | (H*H)
    const auto h64 = widen<4, 64>(h);
    const auto p3 = mul(qsel(ql, widen<4, 64>(b.y), h64, widen<4, 64>(z1z2), h64),
                        qsel(ql, widen<4, 6>(z1c), h, h, h));                          // 14*16+15 = 239 ok; 64*6 ok
    const auto s2 = qread<0>(p3), hh = qread<1>(p3), z3 = qread<2>(p3);
    if (is_zero(hh)) {  // same x: the point itself or its negative -- the complete one-lane routine, on every copy
        jac28_add(a, ainf, b);
        return;
    }
    const auto r = sub(s2, s1);                                                       // <4,6>
    // step 4: H*HH | U1*HH | R*R | (R*R)
    const auto hh46 = widen<4, 6>(hh);
    const auto p4 = mul(qsel(ql, h, widen<4, 6>(u1), r, r), qsel(ql, hh46, hh46, r, r));   // 239 ok; 36 ok
    const auto hhh = qread<0>(p4), v = qread<1>(p4), rr = qread<2>(p4);
    const auto x3 = norm(sub(rr, add(hhh, add(v, v))));                               // <6,10> -> <1,10>
    const auto dv = sub(v, x3);                                                       // <4,18>
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    const auto s1n = sub(zero, s1);                                                   // <4,4> = -S1
    // step 5: R*(V - X3) | (-S1)*HHH ; Y3 is their sum
    const auto rn = widen<4, 6>(norm(r)), sn = widen<4, 6>(s1n);
    const auto p5 = mul(qsel(ql, rn, sn, rn, sn), qsel(ql, dv, widen<4, 18>(hhh), dv, widen<4, 18>(hhh)));   // 239 ok; 108 ok
    const auto y3 = norm(add(qread<0>(p5), qread<1>(p5)));                            // <2,4> -> <1,4>
    a.x = widen<1, 34>(x3);
    a.y = widen<1, 34>(y3);
    a.z = widen<2, 4>(z3);
}

// [k]P for a 128-bit k (one GLV half), uniform 4-bit windows: the quad form of xyzz28_mul_w4_128.
// Only for points of the prime-order subgroup (every multiple 1..15 is finite).  All four lanes of the quad
// pass the same arguments and receive the same result.  (The first form of the round, kept for A/B builds with
// CKZG_QUAD_W4_UNSIGNED: 15-entry table, five-step additions; the signed-window form below replaces it.)

// a <- a + b in XYZZ coordinates (add-2008-s), both operands general: four product steps instead of fourteen
// products.  Quad form of xyzz28_add; ainf / binf are the (replicated) infinity flags.
__device__ __forceinline__ void xyzz28_add_quad(XYZZ28 &a, bool &ainf, const XYZZ28 &b, bool binf, int ql) {
    if (binf) return;
    if (ainf) {
        a = b;
        ainf = false;
        return;
    }
    // step 1: X1*ZZ2 | X2*ZZ1 | Y1*ZZZ2 | Y2*ZZZ1
    const auto p1 = mul(qsel(ql, a.x, b.x, widen<1, 10>(a.y), widen<1, 10>(b.y)), qsel(ql, b.zz, a.zz, b.zzz, a.zzz));  // 14+15 ok; 20 ok
    const auto u1 = qread<0>(p1), u2 = qread<1>(p1), s1 = qread<2>(p1), s2 = qread<3>(p1);
    const auto p = sub(u2, u1);   // <4,6>
    const auto r = sub(s2, s1);   // <4,6>
    // step 2: P*P | ZZ1*ZZ2 | ZZZ1*ZZZ2 | R*R
    const auto zz1 = widen<4, 6>(a.zz), zz2 = widen<4, 6>(b.zz), zzz1 = widen<4, 6>(a.zzz), zzz2 = widen<4, 6>(b.zzz);
    const auto p2 = mul(qsel(ql, p, zz1, zzz1, r), qsel(ql, p, zz2, zzz2, r));   // 14*16+15 = 239 ok; 36 ok
    const auto pp = qread<0>(p2), zzab = qread<1>(p2), zzzab = qread<2>(p2), rr = qread<3>(p2);
    if (is_zero(pp)) {  // same x: the complete one-lane routine, on every copy
        xyzz28_add(a, ainf, b, binf);
        return;
    }
    // step 3: P*PP | U1*PP | ZZ1ZZ2*PP
    const auto p3 = mul(qsel(ql, p, widen<4, 6>(u1), widen<4, 6>(zzab), p), pp);   // 14*4+15 ok; 12 ok
    const auto ppp = qread<0>(p3), q = qread<1>(p3), zz3 = qread<2>(p3);
    const auto x3 = norm(sub(rr, add(ppp, add(q, q))));   // <6,10> -> <1,10>
    const auto d = sub(q, x3);                            // <4,18>
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
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

// Fold the THREADS per-thread accumulators of a workgroup into thread 0 with the quad addition: every pair of a
// level is added by one quad (four product steps instead of a lane's fourteen products), so a level of the
// 256-thread tree costs ~1/3 of the one-lane fold and the tail of an accumulate workgroup -- when its SIMDs are
// nearly empty -- shrinks accordingly.  LDS is limb-major ([56 limbs + flag][THREADS] u32).
template <int THREADS>
__device__ __forceinline__ void block_reduce_xyzz28_quad(XYZZ28 &acc, bool &inf, uint32_t (*sh)[THREADS]) {
    const int tid = threadIdx.x, ql = tid & 3, quad_id = tid >> 2;
    constexpr int QUADS = THREADS / 4;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
#pragma unroll
        for (int k = 0; k < 56; k++) sh[k][tid] = src[k];
        sh[56][tid] = inf ? 1u : 0u;
    }
    __syncthreads();
    for (int s = THREADS / 2; s >= 1; s >>= 1) {
        for (int pr = quad_id; pr < s; pr += QUADS) {
            XYZZ28 x, y;
            uint32_t *dx = reinterpret_cast<uint32_t *>(&x), *dy = reinterpret_cast<uint32_t *>(&y);
#pragma unroll
            for (int k = 0; k < 56; k++) {
                dx[k] = sh[k][pr];
                dy[k] = sh[k][pr + s];
            }
            bool xi = sh[56][pr] != 0;
            const bool yi = sh[56][pr + s] != 0;
            xyzz28_add_quad(x, xi, y, yi, ql);
            if (ql == 0) {
#pragma unroll
                for (int k = 0; k < 56; k++) sh[k][pr] = dx[k];
                sh[56][pr] = xi ? 1u : 0u;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(&acc);
#pragma unroll
        for (int k = 0; k < 56; k++) dst[k] = sh[k][0];
        inf = sh[56][0] != 0;
    }
}

// [|x|]P for the BLS parameter (63 doublings, 5 additions): quad form of jac28_mul_bls_x
__device__ __forceinline__ void jac28_mul_bls_x_quad(JAC28 &out, bool &out_inf, const JAC28 &p, bool p_inf, int ql) {
    JAC28 acc = p;
    bool inf = p_inf;
    if (!p_inf) {
        const JACT28 pt = jac28_table_entry(p);
        for (int b = 62; b >= 0; b--) {
            if (!inf) jac28_dbl_quad(acc, ql);
            if (b == 62 || b == 60 || b == 57 || b == 48 || b == 16) jac28_add_quad(acc, inf, pt, ql);
        }
    }
    out = acc;
    out_inf = inf;
}

// quad form of g1_28_in_subgroup: [x^2]P == (beta^2 X, -Y) for a finite curve point (any point of E(Fp):
// the additions fall back to the complete routine when they meet an exceptional case)
__device__ __noinline__ bool g1_28_in_subgroup_quad(const F28<1, 2> &x, const F28<1, 2> &y, int ql) {
    JAC28 p, q1, q;
    p.x = widen<1, 34>(x);
    p.y = widen<1, 34>(y);
    p.z = widen<2, 4>(f28_one());
    bool i1, i2;
    jac28_mul_bls_x_quad(q1, i1, p, false, ql);
    jac28_mul_bls_x_quad(q, i2, q1, i1, ql);
    if (i2) return false;
    auto zz = sqr(q.z);
    auto bx = mul(x, f28_const<1, 1>(FP28_BETA_LAMBDA2));
    if (!f28_equal(q.x, mul(bx, zz))) return false;
    return is_zero(mul(add(q.y, mul(y, mul(q.z, zz))), f28_one()));
}

// ---- the co-Z table and the mixed addition of g1_28.hpp (section "effectively affine" table) on four lanes ----
// Two more steps go: the accumulator carries Z^2 (computed in a spare lane of the previous doubling or addition),
// so a mixed addition is FOUR product steps (the Jacobian-table form needs five, six for a phi entry), and the
// table costs 15 steps instead of 30.

// a <- 2a, zz <- (new Z)^2: jac28_dbl_quad with the square of 2YZ in the spare lane of step 2
__device__ __forceinline__ void jac28_dbl_quad_zz(JAC28 &a, F28<1, 2> &zz, int ql) {
    const auto x = widen<2, 34>(a.x), y = widen<2, 34>(a.y), z = widen<2, 34>(a.z);
    const auto p1 = mul(qsel(ql, x, y, y, x), qsel(ql, x, y, z, x));     // X*X | Y*Y | Y*Z | (X*X)
    const auto A = qread<0>(p1), B = qread<1>(p1), YZ = qread<2>(p1);
    const auto E = add(add(A, A), A);                                     // <3,6>
    const auto Z3 = add(YZ, YZ);                                          // <2,4>
    const auto e = widen<3, 34>(E), b = widen<3, 34>(B), xx = widen<3, 34>(a.x), z3 = widen<3, 34>(Z3);
    const auto p2 = mul(qsel(ql, e, b, xx, z3), qsel(ql, e, b, b, z3));   // E*E | B*B | X*B | Z3*Z3; 14*9+15 ok
    const auto F = qread<0>(p2), C = qread<1>(p2), XB = qread<2>(p2);
    zz = qread<3>(p2);
    const auto XB2 = add(XB, XB);
    const auto D = add(XB2, XB2);                                         // <4,8>
    const auto X3 = norm(sub(F, add(D, D)));                              // <1,34>
    const auto dx = norm(sub(D, X3));                                     // <1,72>
    const auto C2 = add(C, C);
    const auto C4 = add(C2, C2);
    const auto C8 = add(C4, C4);                                          // <8,16>
    const auto Y3 = norm(sub(mul(E, dx), C8));                            // <1,34>
    a.x = X3;
    a.y = Y3;
    a.z = Z3;
}

// a <- a + (x2, +-y2), the point affine on the curve a lives on; zz = Z(a)^2 in and out.  Four product steps.
__device__ __forceinline__ void jac28_madd_quad_zz(JAC28 &a, F28<1, 2> &zz, bool &ainf, const F28<1, 20> &x2,
                                                   const F28<1, 20> &y2in, bool neg, int ql) {
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
    F28<1, 41> y2;
    {
        const auto ny = norm(sub_k<21>(zero, y2in));                      // <1,21>
#pragma unroll
        for (int j = 0; j < 14; j++) y2.l[j] = neg ? ny.l[j] : y2in.l[j];
    }
    if (ainf) {
        a.x = widen<1, 34>(x2);
        a.y = widen<1, 34>(mul(y2, f28_one()));
        a.z = widen<2, 4>(f28_one());
        zz = widen<1, 2>(f28_one());
        ainf = false;
        return;
    }
    // step A: X2*ZZ1 | Z1*ZZ1
    const auto x2w = widen<2, 41>(x2), azw = widen<2, 41>(a.z);
    const auto pA = mul(qsel(ql, x2w, azw, x2w, azw), widen<2, 41>(zz));  // 14*4+15 ok; 41*41 ok
    const auto u2 = qread<0>(pA), z1c = qread<1>(pA);
    const auto h = sub_k<35>(u2, a.x);                                    // <4,37>
    // step B: Y2*Z1^3 | H*H | Z1*H | (H*H)
    const auto hw = widen<4, 41>(h);
    const auto pB = mul(qsel(ql, widen<4, 41>(y2), hw, widen<4, 41>(a.z), hw),
                        qsel(ql, widen<4, 37>(z1c), h, h, h));            // 14*16+15 = 239 ok; 41*37 ok
    const auto s2 = qread<0>(pB), hh = qread<1>(pB), z3 = qread<2>(pB);
    const auto r = sub_k<35>(s2, a.y);                                    // <4,37>
    if (is_zero(hh)) {  // same x: the table point itself (double it) or its negative (infinity)
        if (is_zero(mul(r, f28_one()))) {
            a.x = widen<1, 34>(x2);
            a.y = widen<1, 34>(mul(y2, f28_one()));
            a.z = widen<2, 4>(f28_one());
            jac28_dbl_quad_zz(a, zz, ql);
        } else {
            ainf = true;
        }
        return;
    }
    // step C: H*HH | X1*HH | R*R | Z3*Z3
    const auto hh37 = widen<4, 37>(hh), z3w = widen<4, 37>(z3);
    const auto pC = mul(qsel(ql, h, widen<4, 37>(a.x), r, z3w), qsel(ql, hh37, hh37, r, z3w));   // 239 ok; 37*37 ok
    const auto hhh = qread<0>(pC), v = qread<1>(pC), rr = qread<2>(pC);
    const auto zz3 = qread<3>(pC);
    const auto x3 = norm(sub(rr, add(hhh, add(v, v))));                   // <6,10> -> <1,10>
    const auto dv = sub(v, x3);                                           // <4,18>
    const auto s1n = sub_k<35>(zero, a.y);                                // <3,35> = -Y1
    // step D: R*(V - X3) | (-Y1)*HHH ; Y3 is their sum
    const auto rn = widen<4, 37>(norm(r)), sn = widen<4, 37>(s1n);
    const auto h18 = widen<4, 18>(hhh);
    const auto pD = mul(qsel(ql, rn, sn, rn, sn), qsel(ql, dv, h18, dv, h18));   // 239 ok; 37*18 ok
    const auto y3 = norm(add(qread<0>(pD), qread<1>(pD)));                // <2,4> -> <1,4>
    a.x = widen<1, 34>(x3);
    a.y = widen<1, 34>(y3);
    a.z = widen<2, 4>(z3);
    zz = zz3;
}

// coz28_addu on four lanes, three steps; up to two further points over the old Z follow in spare lanes (q1 fully,
// q2's x only: its y is returned by the caller's next step through w)
__device__ __forceinline__ void coz28_addu_quad(CoZ28 &sum, CoZ28 &p1, CoZ28 &p2, F28<1, 2> &z, F28<4, 6> &w,
                                                CoZ28 *q1, CoZ28 *q2, int ql) {
    const auto dX = sub_k<21>(p2.x, p1.x);                                // <4,41>
    const auto dY = sub_k<21>(p2.y, p1.y);
    // step 1: dX*dX | dY*dY | Z*dX | (dX*dX)
    const auto pa = mul(qsel(ql, dX, dY, widen<4, 41>(z), dX), qsel(ql, dX, dY, dX, dX));   // 239 ok; 41*41 ok
    const auto c = qread<0>(pa), D = qread<1>(pa);
    z = qread<2>(pa);
    // step 2: X1*c | X2*c | Q1.x*c | Q2.x*c
    const auto pb = mul(qsel(ql, p1.x, p2.x, q1 ? q1->x : p1.x, q2 ? q2->x : p1.x), c);
    const auto W1 = qread<0>(pb), W2 = qread<1>(pb), Q1x = qread<2>(pb), Q2x = qread<3>(pb);
    w = sub(W2, W1);                                                      // <4,6> = dX^3
    const auto X3 = norm(sub(D, add(W1, W2)));                            // <5,10> -> <1,10>
    const auto t = sub(W1, X3);                                           // <4,18>
    // step 3: Y1*w | Y2*w | dY*(W1 - X3) | Q1.y*w
    const auto w18 = widen<4, 18>(w);
    const auto pc = mul(qsel(ql, widen<4, 41>(p1.y), widen<4, 41>(p2.y), dY, widen<4, 41>(q1 ? q1->y : p1.y)),
                        qsel(ql, w18, w18, t, w18));                      // 239 ok; 41*18 ok
    const auto A1 = qread<0>(pc), A2 = qread<1>(pc), M = qread<2>(pc), Q1y = qread<3>(pc);
    const auto Y3 = norm(sub(M, A1));                                     // <4,6> -> <1,6>
    sum.x = widen<1, 20>(X3);
    sum.y = widen<1, 20>(Y3);
    p1.x = widen<1, 20>(W1);
    p1.y = widen<1, 20>(A1);
    p2.x = widen<1, 20>(W2);
    p2.y = widen<1, 20>(A2);
    if (q1) {
        q1->x = widen<1, 20>(Q1x);
        q1->y = widen<1, 20>(Q1y);
    }
    if (q2) q2->x = widen<1, 20>(Q2x);
}

// eat28_build on four lanes: 15 product steps
__device__ __forceinline__ void eat28_build_quad(EAT28 (&tbl)[4], F28<1, 2> &zc, const XYZZ28 &p, int ql) {
    // step 0: X1 = x*zz | Y1 = y*zzz
    const auto px = p.x, py = widen<1, 10>(p.y);
    const auto p0 = mul(qsel(ql, px, py, px, py), qsel(ql, p.zz, p.zzz, p.zz, p.zzz));
    const auto X1 = qread<0>(p0), Y1 = qread<1>(p0);
    // the doubling that also leaves P over 2P's Z (eat28_build): steps 1-3
    const auto p1 = mul(qsel(ql, X1, Y1, Y1, X1), qsel(ql, X1, Y1, p.zz, X1));   // X1^2 | Y1^2 | Y1*Z1
    const auto A = qread<0>(p1), B = qread<1>(p1), YZ = qread<2>(p1);
    const auto E = add(add(A, A), A);                                     // <3,6>
    const auto e = widen<3, 6>(E), b = widen<3, 6>(B), xx = widen<3, 6>(X1), zr = widen<3, 6>(add(YZ, YZ));
    const auto one3 = widen<3, 6>(f28_one());
    const auto p2 = mul(qsel(ql, e, b, xx, zr), qsel(ql, e, b, b, one3));  // E^2 | B^2 | X1*B | Z2 back under <1,2>
    const auto F = qread<0>(p2), C = qread<1>(p2), XB = qread<2>(p2);
    F28<1, 2> z = qread<3>(p2);
    const auto XB2 = add(XB, XB);
    const auto D = add(XB2, XB2);                                         // <4,8>
    const auto X2 = norm(sub_k<17>(F, add(D, D)));                        // <1,19>
    const auto dx = norm(sub_k<20>(D, X2));                               // <1,28>
    const auto C2 = add(C, C);
    const auto C4 = add(C2, C2);
    const auto C8 = add(C4, C4);                                          // <8,16>
    const auto Y2 = norm(sub_k<17>(mul(E, dx), C8));                      // <1,19>
    CoZ28 t1, t2, t3, t5, t7;
    t1.x = widen<1, 20>(norm(D));
    t1.y = widen<1, 20>(norm(C8));
    t2.x = widen<1, 20>(X2);
    t2.y = widen<1, 20>(Y2);
    F28<4, 6> w;
    coz28_addu_quad(t3, t2, t1, z, w, nullptr, nullptr, ql);              // 3P = 2P + P
    coz28_addu_quad(t5, t2, t3, z, w, &t1, nullptr, ql);                  // 5P = 2P + 3P; P follows
    coz28_addu_quad(t7, t2, t5, z, w, &t1, &t3, ql);                      // 7P = 2P + 5P; P and 3P.x follow
    // last step but one: 3P.y * w (the same product in every lane); last: beta * x of the four entries
    t3.y = widen<1, 20>(mul(t3.y, w));
    const auto bx = mul(qsel(ql, t1.x, t3.x, t5.x, t7.x), f28_const<1, 1>(FP28_BETA_LAMBDA));
    tbl[0].x = t1.x; tbl[0].y = t1.y; tbl[0].bx = widen<1, 20>(qread<0>(bx));
    tbl[1].x = t3.x; tbl[1].y = t3.y; tbl[1].bx = widen<1, 20>(qread<1>(bx));
    tbl[2].x = t5.x; tbl[2].y = t5.y; tbl[2].bx = widen<1, 20>(qread<2>(bx));
    tbl[3].x = t7.x; tbl[3].y = t7.y; tbl[3].bx = widen<1, 20>(qread<3>(bx));
    zc = z;
}

// a <- a + b (b finite, Jacobian with cached Z^2, Z^3), zz = Z(a)^2 in and out: with Z1^2 at hand the products
// X2*Z1^2 and Z1^3 move up into the first step and the addition is FOUR product steps instead of five.
__device__ __forceinline__ void jac28_add_quad_zz(JAC28 &a, F28<1, 2> &zz, bool &ainf, const JACT28 &b, int ql) {
    if (ainf) {
        a.x = b.x;
        a.y = widen<1, 34>(mul(b.y, f28_one()));
        a.z = b.z;
        zz = b.zz;
        ainf = false;
        return;
    }
    // step 1: X1*ZZ2 | Y1*ZZZ2 | X2*ZZ1 | Z1*ZZ1
    const auto az = widen<2, 34>(a.z);
    const auto zz1 = widen<2, 4>(zz);
    const auto p1 = mul(qsel(ql, widen<2, 34>(a.x), widen<2, 34>(a.y), widen<2, 34>(b.x), az),
                        qsel(ql, widen<2, 4>(b.zz), widen<2, 4>(b.zzz), zz1, zz1));   // 14*4+15 ok; 34*4 ok
    const auto u1 = qread<0>(p1), s1 = qread<1>(p1), u2 = qread<2>(p1), z1c = qread<3>(p1);
    const auto h = sub(u2, u1);                                                       // <4,6>
    // step 2: Y2*Z1^3 | H*H | Z1*Z2 | (H*H)
    const auto h64 = widen<4, 64>(h);
    const auto p2 = mul(qsel(ql, widen<4, 64>(b.y), h64, widen<4, 64>(a.z), h64),
                        qsel(ql, widen<4, 6>(z1c), h, widen<4, 6>(b.z), h));          // 239 ok; 64*6 ok
    const auto s2 = qread<0>(p2), hh = qread<1>(p2), z1z2 = qread<2>(p2);
    if (is_zero(hh)) {  // same x: the complete one-lane routine, on every copy
        jac28_add(a, ainf, b);
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
    F28<1, 0> zero;
#pragma unroll
    for (int j = 0; j < 14; j++) zero.l[j] = 0;
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

// [k]P for a 128-bit k (one GLV half): uniform SIGNED 4-bit windows (digits -8..8, 33 of them), so the table is
// P..8P (7 additions instead of 14) and every addition is the four-step form above.  ~560 dependent product steps
// instead of ~650.  Only for points of the prime-order subgroup (every multiple 1..8 is finite).  All four lanes
// of the quad pass the same arguments and receive the same result.
__device__ __noinline__ void xyzz28_mul_w4_128_quad(XYZZ28 &out, bool &out_inf, const XYZZ28 &p, bool p_inf,
                                                    const uint32_t *k, int ql) {
    JACT28 tbl[8];
    JAC28 acc;
    F28<1, 2> zz;
    bool inf = true;
    if (!p_inf) {
        // carries of the signed recoding: digit i = nibble i + c_i - 16 c_(i+1), c_(i+1) = [nibble i + c_i > 8]
        uint64_t cm = 0;
        {
            uint32_t c = 0;
            for (int i = 0; i < 32; i++) {
                const uint32_t nib = ((k[i >> 3] >> ((i & 7) * 4)) & 15u) + c;
                c = nib > 8u ? 1u : 0u;
                cm |= (uint64_t)c << (i + 1);
            }
        }
        JAC28 cur = jac28_from_xyzz(p);
        tbl[0] = jac28_table_entry(cur);
        F28<1, 2> czz = tbl[0].zz;
        for (int i = 1; i < 8; i++) {
            bool ci = false;
            jac28_add_quad_zz(cur, czz, ci, tbl[0], ql);
            tbl[i].x = cur.x;
            tbl[i].y = widen<1, 64>(cur.y);
            tbl[i].z = cur.z;
            tbl[i].zz = czz;
        }
        // Z^3 of entries 2P..8P: two steps (entry 1 + 4 per step; lane 3 of the second repeats a product)
        {
            const auto pz = mul(qsel(ql, tbl[1].z, tbl[2].z, tbl[3].z, tbl[4].z), qsel(ql, tbl[1].zz, tbl[2].zz, tbl[3].zz, tbl[4].zz));
            tbl[1].zzz = qread<0>(pz); tbl[2].zzz = qread<1>(pz); tbl[3].zzz = qread<2>(pz); tbl[4].zzz = qread<3>(pz);
            const auto pw = mul(qsel(ql, tbl[5].z, tbl[6].z, tbl[7].z, tbl[7].z), qsel(ql, tbl[5].zz, tbl[6].zz, tbl[7].zz, tbl[7].zz));
            tbl[5].zzz = qread<0>(pw); tbl[6].zzz = qread<1>(pw); tbl[7].zzz = qread<2>(pw);
        }
        for (int w = 32; w >= 0; w--) {
            if (!inf) {
                jac28_dbl_quad_zz(acc, zz, ql);
                jac28_dbl_quad_zz(acc, zz, ql);
                jac28_dbl_quad_zz(acc, zz, ql);
                jac28_dbl_quad_zz(acc, zz, ql);
            }
            const int nib = w < 32 ? (int)((k[w >> 3] >> ((w & 7) * 4)) & 15u) : 0;
            const int d = nib + (int)((cm >> w) & 1u) - (w < 32 ? 16 * (int)((cm >> (w + 1)) & 1u) : 0);
            if (d != 0) {
                // ONE call site for both signs: the quads of a wave have different digits, and two call sites
                // would run the addition twice per window (measured: slower than the unsigned form)
                JACT28 e = tbl[(d > 0 ? d : -d) - 1];
                const JACT28 n = jact28_neg(e);
#pragma unroll
                for (int j = 0; j < 14; j++) e.y.l[j] = d < 0 ? n.y.l[j] : e.y.l[j];
                jac28_add_quad_zz(acc, zz, inf, e, ql);
            }
        }
    }
    if (!inf) out = jac28_to_xyzz(acc);
    out_inf = inf;
}

// [k]P = [k1]P + [k2]phi(P) with both halves in width-4 NAF: quad form of xyzz28_mul_glv_naf (the G1 FFT's
// twiddle multiplication; the digit strings are shared by the whole wave).  Co-Z table, mixed additions.
__device__ __noinline__ void xyzz28_mul_glv_naf_quad(XYZZ28 &out, bool &out_inf, const XYZZ28 &p, bool p_inf,
                                                     const int8_t *naf1, const int8_t *naf2, int ql) {
    bool inf = true;
    if (!p_inf) {
        EAT28 tbl[4];
        F28<1, 2> zc, zz;
        eat28_build_quad(tbl, zc, p, ql);
        JAC28 acc;
        for (int i = GLV_NAF_LEN - 1; i >= 0; i--) {
            if (!inf) jac28_dbl_quad_zz(acc, zz, ql);
            const int d1 = naf1[i], d2 = naf2[i];
            if (d1) {
                const EAT28 &e = tbl[(d1 > 0 ? d1 : -d1) >> 1];
                jac28_madd_quad_zz(acc, zz, inf, e.x, e.y, d1 < 0, ql);
            }
            if (d2) {
                const EAT28 &e = tbl[(d2 > 0 ? d2 : -d2) >> 1];
                jac28_madd_quad_zz(acc, zz, inf, e.bx, e.y, d2 < 0, ql);
            }
        }
        if (!inf) {
            acc.z = widen<2, 4>(mul(acc.z, zc));   // home from the isomorphic curve
            out = jac28_to_xyzz(acc);
        }
    }
    out_inf = inf;
}

}  // namespace quad
}  // namespace ckzg
