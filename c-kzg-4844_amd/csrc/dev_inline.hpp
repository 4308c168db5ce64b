// dev_inline.hpp -- small device-only helpers shared by the .hip translation units.
#pragma once
#include "device.hpp"

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

}  // namespace dev
}  // namespace ckzg
