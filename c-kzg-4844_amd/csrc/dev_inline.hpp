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

// Digits of one canonical scalar for the fixed-base MSM kernels (device.hpp: FixedBaseTable): balanced
// GLV split, then twin signed windows per half -- windows 0..twin-1 from k2 (the phi half),
// twin..2*twin-1 from k1; digit w lands at dst[w * stride].
__device__ __forceinline__ void glv_digits(int16_t *dst, size_t stride, const uint32_t *s, int wbits, int twin) {
    uint32_t m1[4], m2[4];
    bool n1, n2;
    glv_split_signed(s, m1, n1, m2, n2);
    recode_signed_128(dst, stride, m2, n2, wbits, twin);
    recode_signed_128(dst + (size_t)twin * stride, stride, m1, n1, wbits, twin);
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
