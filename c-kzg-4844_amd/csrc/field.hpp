// field.hpp -- Montgomery arithmetic for the two BLS12-381 prime fields on 32-bit limbs.
//
// One source for device (gfx950: v_mad_u64_u32 carry chains, everything unrolled into VGPRs) and
// host (setup-time work: point decompression, pairing tower).  Limb layout: little-endian u32,
// Montgomery radix 2^(32*N) -- on a little-endian machine this is byte-for-byte the layout of
// blst_fr (4xu64, R=2^256) and blst_fp (6xu64, R=2^384), i.e. the fr_t / fp_t the reference
// keeps inside KZGSettings (src/common/fr.h:27, bindings/go/blst_headers/blst.h:58-67).
//
// Replaces, for the hot path, the blst field calls behind src/common/fr.c:32-161 and everything
// inside blst_p1_add_or_double / blst_p1s_mult_pippenger (src/common/ec.c:29, lincomb.c:114).
#pragma once
#include <stdint.h>
#include "bls_consts32.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#define HDNI __host__ __device__
#else
#define HD inline __attribute__((always_inline))
#define HDNI
#endif

namespace ckzg {

struct FpParams {
    static constexpr int N = 12;
    static constexpr uint32_t NINV = (uint32_t)FP_NINV32;
    static constexpr int BITS = 381;
    HD static constexpr uint32_t mod(int i) { return FP_P[i]; }
    HD static constexpr uint32_t r1(int i) { return FP_R1[i]; }
    HD static constexpr uint32_t r2(int i) { return FP_R2[i]; }
};

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t NINV = (uint32_t)FR_NINV32;
    static constexpr int BITS = 255;
    HD static constexpr uint32_t mod(int i) { return FR_R[i]; }
    HD static constexpr uint32_t r1(int i) { return FR_R1[i]; }
    HD static constexpr uint32_t r2(int i) { return FR_R2[i]; }
};

// Element of Z/m in Montgomery form, fully reduced: 0 <= value < m.
template <class P>
struct Mont {
    static constexpr int N = P::N;
    uint32_t l[N];

    HD static Mont zero() {
        Mont r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    HD static Mont one() {
        Mont r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::r1(i);
        return r;
    }
    HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i];
        return acc == 0;
    }
    HD bool operator==(const Mont &o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    HD bool operator!=(const Mont &o) const { return !(*this == o); }
};

// r = a - b over N limbs, returns the borrow (0/1)
template <int N>
HD uint32_t limbs_sub(uint32_t *r, const uint32_t *a, const uint32_t *b) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a[i] - b[i] - br;
        r[i] = (uint32_t)d;
        br = (uint32_t)(d >> 32) & 1u;
    }
    return br;
}

template <int N>
HD uint32_t limbs_add(uint32_t *r, const uint32_t *a, const uint32_t *b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t s = (uint64_t)a[i] + b[i] + c;
        r[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    return c;
}

// a >= b as N-limb integers
template <int N>
HD bool limbs_geq(const uint32_t *a, const uint32_t *b) {
    uint32_t t[N];
    return limbs_sub<N>(t, a, b) == 0;
}

template <class P>
HD void mod_limbs(uint32_t *m) {
#pragma unroll
    for (int i = 0; i < P::N; i++) m[i] = P::mod(i);
}

// Both moduli leave the top limb with spare bits (381 < 384, 255 < 256): a + b never carries out.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
// host forms of add / sub on 64-bit limbs (same bytes, half the carry chain)
template <class P>
inline Mont<P> host_addsub(const Mont<P> &a, const Mont<P> &b, bool subtract) {
    constexpr int H = P::N / 2;
    typedef unsigned __int128 u128;
    uint64_t x[H], y[H], m[H], t[H], u[H];
    __builtin_memcpy(x, a.l, sizeof x);
    __builtin_memcpy(y, b.l, sizeof y);
#pragma unroll
    for (int i = 0; i < H; i++) m[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
    uint64_t c = 0, c2 = 0;
    if (!subtract) {
#pragma unroll
        for (int i = 0; i < H; i++) {
            u128 s = (u128)x[i] + y[i] + c;
            t[i] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
#pragma unroll
        for (int i = 0; i < H; i++) {
            u128 d = (u128)t[i] - m[i] - c2;
            u[i] = (uint64_t)d;
            c2 = (uint64_t)(d >> 64) & 1;
        }
        // a + b never carries out of the top limb (both moduli leave spare bits): keep t if t < m
#pragma unroll
        for (int i = 0; i < H; i++) t[i] = c2 ? t[i] : u[i];
    } else {
#pragma unroll
        for (int i = 0; i < H; i++) {
            u128 d = (u128)x[i] - y[i] - c;
            t[i] = (uint64_t)d;
            c = (uint64_t)(d >> 64) & 1;
        }
#pragma unroll
        for (int i = 0; i < H; i++) {
            u128 s = (u128)t[i] + m[i] + c2;
            u[i] = (uint64_t)s;
            c2 = (uint64_t)(s >> 64);
        }
#pragma unroll
        for (int i = 0; i < H; i++) t[i] = c ? u[i] : t[i];
    }
    Mont<P> r;
    __builtin_memcpy(r.l, t, sizeof t);
    return r;
}
#endif

template <class P>
HD Mont<P> add(const Mont<P> &a, const Mont<P> &b) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
    return host_addsub<P>(a, b, false);
#else
    constexpr int N = P::N;
    uint32_t m[N], t[N], s[N];
    mod_limbs<P>(m);
    limbs_add<N>(t, a.l, b.l);
    uint32_t br = limbs_sub<N>(s, t, m);
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = br ? t[i] : s[i];
    return r;
#endif
}

template <class P>
HD Mont<P> sub(const Mont<P> &a, const Mont<P> &b) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
    return host_addsub<P>(a, b, true);
#else
    constexpr int N = P::N;
    uint32_t m[N], t[N], s[N];
    mod_limbs<P>(m);
    uint32_t br = limbs_sub<N>(t, a.l, b.l);
    limbs_add<N>(s, t, m);
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = br ? s[i] : t[i];
    return r;
#endif
}

template <class P>
HD Mont<P> neg(const Mont<P> &a) {
    return sub(Mont<P>::zero(), a);
}

template <class P>
HD Mont<P> dbl(const Mont<P> &a) {
    return add(a, a);
}

// Montgomery product a*b/2^(32N) mod m, operand scanning with the reduction step interleaved
// (CIOS).  Every inner step is one 32x32+64 multiply-add: v_mad_u64_u32 on gfx950.
template <class P>
HD Mont<P> mul(const Mont<P> &a, const Mont<P> &b) {
    constexpr int N = P::N;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
    // Host code path (setup, pairing, small-batch verification): the same CIOS on 64-bit limbs -- on a
    // little-endian host the 32-bit limb array IS the 64-bit limb array -- with the modulus words and
    // -m^-1 mod 2^64 folded at compile time.
    constexpr int H = N / 2;
    typedef unsigned __int128 u128;
    uint64_t x[H], y[H], t[H + 2];
    __builtin_memcpy(x, a.l, sizeof x);
    __builtin_memcpy(y, b.l, sizeof y);
    constexpr uint64_t m0 = (uint64_t)P::mod(0) | ((uint64_t)P::mod(1) << 32);
    // m^-1 mod 2^64 from the 32-bit constant by one Newton step
    constexpr uint64_t inv32 = (uint64_t)0 - (uint64_t)P::NINV;
    constexpr uint64_t ninv = (uint64_t)0 - inv32 * (2 - m0 * inv32);
    uint64_t m[H];
#pragma unroll
    for (int i = 0; i < H; i++) m[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
#pragma unroll
    for (int i = 0; i < H + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < H; i++) {
        u128 c = 0;
#pragma unroll
        for (int j = 0; j < H; j++) {
            c += (u128)x[j] * y[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[H];
        t[H] = (uint64_t)c;
        t[H + 1] = (uint64_t)(c >> 64);
        const uint64_t q = t[0] * ninv;
        c = ((u128)q * m[0] + t[0]) >> 64;
#pragma unroll
        for (int j = 1; j < H; j++) {
            c += (u128)q * m[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[H];
        t[H - 1] = (uint64_t)c;
        t[H] = t[H + 1] + (uint64_t)(c >> 64);
    }
    uint64_t s64[H];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < H; i++) {
        u128 d = (u128)t[i] - m[i] - br;
        s64[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < H; i++) s64[i] = br ? t[i] : s64[i];
    __builtin_memcpy(r.l, s64, sizeof s64);
    return r;
#else
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            c = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N] = (uint32_t)c;
        t[N + 1] = (uint32_t)(c >> 32);
        const uint32_t q = t[0] * P::NINV;
        c = ((uint64_t)q * P::mod(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c = (uint64_t)q * P::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (uint32_t)c;
        t[N] = t[N + 1] + (uint32_t)(c >> 32);
    }
    // t < 2m here and t[N] == 0 for both moduli (4m < 2^(32N)); one conditional subtraction
    uint32_t m[N], s[N];
    mod_limbs<P>(m);
    uint32_t br = limbs_sub<N>(s, t, m);
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = br ? t[i] : s[i];
    return r;
#endif
}

template <class P>
HD Mont<P> sqr(const Mont<P> &a) {
    return mul(a, a);
}

// out of / into Montgomery form (canonical little-endian limbs)
template <class P>
HD void to_raw(uint32_t *raw, const Mont<P> &a) {
    Mont<P> one;
#pragma unroll
    for (int i = 0; i < P::N; i++) one.l[i] = (i == 0);
    Mont<P> r = mul(a, one);
#pragma unroll
    for (int i = 0; i < P::N; i++) raw[i] = r.l[i];
}

// raw may be any N-limb integer (reduced mod m as a side effect)
template <class P>
HD Mont<P> from_raw(const uint32_t *raw) {
    Mont<P> a, r2;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        a.l[i] = raw[i];
        r2.l[i] = P::r2(i);
    }
    return mul(a, r2);
}

// a^e, e given as nbits-bit little-endian limb array (not constant time; exponents are public)
template <class P>
HDNI Mont<P> pow_limbs(const Mont<P> &a, const uint32_t *e, int nbits) {
    Mont<P> acc = Mont<P>::one();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = mul(acc, a);
    }
    return acc;
}

using Fp = Mont<FpParams>;
using Fr = Mont<FrParams>;

HDNI inline Fp fp_inv(const Fp &a) {
    uint32_t e[12];
    for (int i = 0; i < 12; i++) e[i] = FP_INV_EXP[i];
    return pow_limbs(a, e, 381);
}

HDNI inline Fr fr_inv(const Fr &a) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FR_INV_EXP[i];
    return pow_limbs(a, e, 255);
}

HD Fr fr_from_u64(uint64_t v) {
    uint32_t raw[8] = {(uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0};
    return from_raw<FrParams>(raw);
}

}  // namespace ckzg
