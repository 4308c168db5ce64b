// ntt.hip -- radix-2 number-theoretic transforms over Fr for gfx950, plus the byte <-> field
// conversions either side of them.
//
// Replaces fr_fft_fast / fr_fft / fr_ifft / coset_fft / coset_ifft (src/eip7594/fft.c:70-146,
// 257-301), shift_poly and poly_lagrange_to_monomial (src/eip7594/poly.c:38-80),
// bit_reversal_permutation on Fr arrays (src/common/utils.c:103-140), blob_to_polynomial
// (src/eip4844/blob.c:31-38) and the bytes_from_bls_field loops (src/eip7594/eip7594.c:113-120).
//
// Design: the reference's recursive DIT works natural-order in -> natural-order out and the
// callers bit-reverse before/after.  Here two butterfly networks are provided and no permutation
// pass exists at all:
//   DIF  natural-order in  -> bit-reversed out   (coefficients -> cells order)
//   DIT  bit-reversed in   -> natural-order out  (blob / cell order -> coefficients)
// A workgroup of 1024 threads owns a tile of 4096 field elements in LDS, stored limb-major
// ([8][4096] u32 = 128 KiB of the CU's 160 KiB) so that the 64 lanes of a wave touch 64 different
// banks; all log2(n) stages of every size-n sub-transform in the tile run out of LDS with one
// coalesced 32-byte-per-lane load and store.  Size 8192 adds one global-memory stage.
#include "device.hpp"

namespace ckzg {
namespace dev {

constexpr int TILE = 4096;
constexpr int TILE_THREADS = 1024;

__device__ __forceinline__ Fr lds_get(uint32_t (*sh)[TILE], int idx) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.l[k] = sh[k][idx];
    return r;
}

__device__ __forceinline__ void lds_put(uint32_t (*sh)[TILE], int idx, const Fr &v) {
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k][idx] = v.l[k];
}

__device__ __forceinline__ Fr load_fr(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

__device__ __forceinline__ void store_fr(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// roots[i] = w^i for the 8192-th root w (8193 entries); a size-m sub-domain uses stride 8192/m
__device__ __forceinline__ Fr twiddle(const Fr *roots, int j, int m, bool inverse) {
    int idx = j * (N_EXT / m);
    return load_fr(roots + (inverse ? N_EXT - idx : idx));
}

// All stages of the size-2^logn transforms contained in each 4096-element tile.
//   DIF: out[brp(k)] = sum_i in[i] w^(ik)        DIT: out[k] = sum_i in[brp(i)] w^(ik)
// `scale` (Montgomery form) multiplies every output; pass one() for none.
template <bool DIF>
__global__ __launch_bounds__(TILE_THREADS) void k_ntt_tile(Fr *data, const Fr *roots, int logn,
                                                           int inverse, Fr scale, int do_scale) {
    __shared__ uint32_t sh[8][TILE];
    Fr *tile = data + (size_t)blockIdx.x * TILE;
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < TILE / TILE_THREADS; r++) {
        int idx = tid + r * TILE_THREADS;
        lds_put(sh, idx, load_fr(tile + idx));
    }
    __syncthreads();
    for (int st = 0; st < logn; st++) {
        int s = DIF ? logn - st : st + 1;  // sub-transform size 2^s at this stage
        int half = 1 << (s - 1);
#pragma unroll
        for (int r = 0; r < (TILE / 2) / TILE_THREADS; r++) {
            int b = tid + r * TILE_THREADS;
            int j = b & (half - 1);
            int i0 = ((b >> (s - 1)) << s) + j;
            int i1 = i0 + half;
            Fr u = lds_get(sh, i0), v = lds_get(sh, i1);
            if (DIF) {
                Fr d = sub(u, v);
                if (j != 0) d = mul(d, twiddle(roots, j, 2 * half, inverse != 0));
                lds_put(sh, i0, add(u, v));
                lds_put(sh, i1, d);
            } else {
                if (j != 0) v = mul(v, twiddle(roots, j, 2 * half, inverse != 0));
                lds_put(sh, i0, add(u, v));
                lds_put(sh, i1, sub(u, v));
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < TILE / TILE_THREADS; r++) {
        int idx = tid + r * TILE_THREADS;
        Fr v = lds_get(sh, idx);
        if (do_scale) v = mul(v, scale);
        store_fr(tile + idx, v);
    }
}

// The one out-of-LDS stage of a size-8192 transform (distance 4096), batched.
//   DIF first stage:  a[i], a[i+4096] <- a[i]+a[i+4096], (a[i]-a[i+4096]) w^i
//   DIT last stage:   a[i], a[i+4096] <- a[i]+w^i a[i+4096], a[i]-w^i a[i+4096]   (then * scale)
template <bool DIF>
__global__ void k_ntt8192_outer(Fr *data, const Fr *roots, size_t total_butterflies, int inverse,
                                Fr scale, int do_scale) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total_butterflies) return;
    size_t vec = g >> 12;
    int i = (int)(g & 4095);
    Fr *a = data + vec * N_EXT;
    Fr u = load_fr(a + i), v = load_fr(a + i + 4096);
    Fr w = twiddle(roots, i, N_EXT, inverse != 0);
    Fr x, y;
    if (DIF) {
        x = add(u, v);
        y = mul(sub(u, v), w);
    } else {
        v = mul(v, w);
        x = add(u, v);
        y = sub(u, v);
        if (do_scale) {
            x = mul(x, scale);
            y = mul(y, scale);
        }
    }
    store_fr(a + i, x);
    store_fr(a + i + 4096, y);
}

static Fr inv_pow2(int logn) {
    // (2^logn)^-1 in Montgomery form: halve `one` logn times  (x/2 = (x + (x odd ? r : 0)) >> 1)
    Fr v = Fr::one();
    uint32_t m[8];
    mod_limbs<FrParams>(m);
    for (int k = 0; k < logn; k++) {
        uint32_t t[9];
        uint32_t c = 0;
        if (v.l[0] & 1u) {
            c = limbs_add<8>(t, v.l, m);
        } else {
            for (int i = 0; i < 8; i++) t[i] = v.l[i];
        }
        t[8] = c;
        for (int i = 0; i < 8; i++) v.l[i] = (t[i] >> 1) | (t[i + 1] << 31);
    }
    return v;
}

// n-point transforms on `count` vectors stored back to back (count*n must be a multiple of 4096
// unless n == 8192).  dif: natural -> bit-reversed; !dif: bit-reversed -> natural.
// inverse: use w^-1 instead of w.  scale: multiply every output by 1/n (fr_ifft, fft.c:127-146,
// is inverse+scale; fk20.c:199-209 is a forward transform followed by * 1/128).
int fr_ntt_batch(DeviceCtx *ctx, Fr *d_data, size_t count, int logn, bool dif, bool inverse,
                 bool scale_by_inv_n) {
    if (count == 0) return 0;
    Fr scale = Fr::one();
    int do_scale = 0;
    if (scale_by_inv_n) {
        scale = inv_pow2(logn);
        do_scale = 1;
    }
    if (logn <= 12) {
        size_t elems = count << logn;
        if (elems % TILE) return 2;
        unsigned tiles = (unsigned)(elems / TILE);
        if (dif) {
            hipLaunchKernelGGL(k_ntt_tile<true>, dim3(tiles), dim3(TILE_THREADS), 0, ctx->stream, d_data,
                               ctx->d_roots, logn, inverse ? 1 : 0, scale, do_scale);
        } else {
            hipLaunchKernelGGL(k_ntt_tile<false>, dim3(tiles), dim3(TILE_THREADS), 0, ctx->stream, d_data,
                               ctx->d_roots, logn, inverse ? 1 : 0, scale, do_scale);
        }
    } else if (logn == 13) {
        size_t bf = count * 4096;
        unsigned blocks = (unsigned)((bf + 255) / 256);
        unsigned tiles = (unsigned)(count * 2);
        if (dif) {
            hipLaunchKernelGGL(k_ntt8192_outer<true>, dim3(blocks), dim3(256), 0, ctx->stream, d_data,
                               ctx->d_roots, bf, inverse ? 1 : 0, scale, 0);
            hipLaunchKernelGGL(k_ntt_tile<true>, dim3(tiles), dim3(TILE_THREADS), 0, ctx->stream, d_data,
                               ctx->d_roots, 12, inverse ? 1 : 0, scale, do_scale);
        } else {
            hipLaunchKernelGGL(k_ntt_tile<false>, dim3(tiles), dim3(TILE_THREADS), 0, ctx->stream, d_data,
                               ctx->d_roots, 12, inverse ? 1 : 0, scale, 0);
            hipLaunchKernelGGL(k_ntt8192_outer<false>, dim3(blocks), dim3(256), 0, ctx->stream, d_data,
                               ctx->d_roots, bf, inverse ? 1 : 0, scale, do_scale);
        }
    } else {
        return 2;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// bytes <-> Fr
// ------------------------------------------------------------------------------------------

// big-endian 32-byte strings -> Montgomery Fr; a value >= r sets bad[g / elems_per_unit]
__global__ void k_bytes_to_fr(Fr *out, uint32_t *bad, const uint8_t *in, size_t total,
                              uint32_t elems_per_unit) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total) return;
    const uint4 *q = reinterpret_cast<const uint4 *>(in + g * 32);
    uint4 a = q[0], b = q[1];
    uint32_t s[8], r[8];
    s[7] = __builtin_bswap32(a.x); s[6] = __builtin_bswap32(a.y);
    s[5] = __builtin_bswap32(a.z); s[4] = __builtin_bswap32(a.w);
    s[3] = __builtin_bswap32(b.x); s[2] = __builtin_bswap32(b.y);
    s[1] = __builtin_bswap32(b.z); s[0] = __builtin_bswap32(b.w);
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = FR_R[k];
    if (limbs_geq<8>(s, r)) {
        if (bad) atomicOr(&bad[g / elems_per_unit], 1u);
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = 0;
    }
    store_fr(out + g, from_raw<FrParams>(s));
}

// Montgomery Fr -> canonical big-endian bytes
__global__ void k_fr_to_bytes(uint8_t *out, const Fr *in, size_t total) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total) return;
    uint32_t raw[8];
    to_raw<FrParams>(raw, load_fr(in + g));
    uint4 *q = reinterpret_cast<uint4 *>(out + g * 32);
    q[0] = make_uint4(__builtin_bswap32(raw[7]), __builtin_bswap32(raw[6]), __builtin_bswap32(raw[5]),
                      __builtin_bswap32(raw[4]));
    q[1] = make_uint4(__builtin_bswap32(raw[3]), __builtin_bswap32(raw[2]), __builtin_bswap32(raw[1]),
                      __builtin_bswap32(raw[0]));
}

// dst[v][0..n_src) = src[v][0..n_src), dst[v][n_src..n_dst) = 0
__global__ void k_zero_extend(Fr *dst, const Fr *src, size_t count, uint32_t n_src, uint32_t n_dst) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= count * n_dst) return;
    size_t v = g / n_dst;
    uint32_t i = (uint32_t)(g % n_dst);
    Fr x = Fr::zero();
    if (i < n_src) x = load_fr(src + v * n_src + i);
    store_fr(dst + g, x);
}

int bytes_to_fr_batch(DeviceCtx *ctx, Fr *d_out, uint32_t *d_bad, const uint8_t *d_in, size_t total,
                      uint32_t elems_per_unit) {
    if (!total) return 0;
    hipLaunchKernelGGL(k_bytes_to_fr, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_out, d_bad, d_in, total, elems_per_unit);
    HIP_TRY(hipGetLastError());
    return 0;
}

int fr_to_bytes_batch(DeviceCtx *ctx, uint8_t *d_out, const Fr *d_in, size_t total) {
    if (!total) return 0;
    hipLaunchKernelGGL(k_fr_to_bytes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_out, d_in, total);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zero_extend_batch(DeviceCtx *ctx, Fr *d_dst, const Fr *d_src, size_t count, uint32_t n_src,
                      uint32_t n_dst) {
    size_t total = count * n_dst;
    if (!total) return 0;
    hipLaunchKernelGGL(k_zero_extend, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_dst, d_src, count, n_src, n_dst);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace dev
}  // namespace ckzg
