// dev_inline.hpp -- small device-only helpers shared by the .hip translation units.
#pragma once
#include "device.hpp"
#include "g1_28.hpp"

namespace ckzg {
namespace dev {

__device__ __forceinline__ void load_be256(uint32_t s[8], const uint8_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    s[7] = __builtin_bswap32(a.x);
    s[6] = __builtin_bswap32(a.y);
    s[5] = __builtin_bswap32(a.z);
    s[4] = __builtin_bswap32(a.w);
    s[3] = __builtin_bswap32(b.x);
    s[2] = __builtin_bswap32(b.y);
    s[1] = __builtin_bswap32(b.z);
    s[0] = __builtin_bswap32(b.w);
}

// signed base-2^wbits digits of a 256-bit integer, digit w written at dst[w * stride].
// Digits lie in [-2^(wbits-1), 2^(wbits-1) - 1], so they fit int16_t up to wbits = 16.  With
// wbits in {4, 8, 16} the nwin windows cover exactly 256 bits: the top digit of a canonical scalar
// (< r < 0.46 * 2^255) stays below 2^(wbits-1) and never carries out.
__device__ __forceinline__ void recode_signed(int16_t *dst, size_t stride, uint32_t s[8], int wbits,
                                              int nwin) {
    const uint32_t mask = (1u << wbits) - 1u, half = 1u << (wbits - 1);
    uint32_t carry = 0;
    for (int w = 0; w < nwin; w++) {
        int d = (int)((s[0] & mask) + carry);
#pragma unroll
        for (int k = 0; k < 7; k++) s[k] = (s[k] >> wbits) | (s[k + 1] << (32 - wbits));
        s[7] >>= wbits;
        carry = 0;
        if ((uint32_t)d >= half) {
            d -= (int)(mask + 1u);
            carry = 1;
        }
        dst[(size_t)w * stride] = (int16_t)d;
    }
}

// Fold the per-thread accumulators (28-bit domain) of a workgroup into thread 0.  LDS is
// limb-major ([56 limbs + infinity flag][THREADS/2] u32) so a wave's lanes hit consecutive banks.
template <int THREADS>
__device__ __forceinline__ void block_reduce_xyzz28(XYZZ28 &acc, bool &inf, uint32_t (*sh)[THREADS / 2]) {
    const int tid = threadIdx.x;
    for (int s = THREADS / 2; s >= 1; s >>= 1) {
        if (tid >= s && tid < 2 * s) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
#pragma unroll
            for (int k = 0; k < 56; k++) sh[k][tid - s] = src[k];
            sh[56][tid - s] = inf ? 1u : 0u;
        }
        __syncthreads();
        if (tid < s) {
            XYZZ28 o;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
            for (int k = 0; k < 56; k++) dst[k] = sh[k][tid];
            xyzz28_add(acc, inf, o, sh[56][tid] != 0);
        }
        __syncthreads();
    }
}

}  // namespace dev
}  // namespace ckzg
