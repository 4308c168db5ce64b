// fk20.hip -- cells and FK20 cell proofs on the GPU.
//
// Replaces compute_cells_and_kzg_proofs (src/eip7594/eip7594.c:61-157), compute_fk20_cell_proofs
// and circulant_coeffs_stride (src/eip7594/fk20.c:55-286), g1_fft / g1_ifft_unscaled
// (src/eip7594/fft.c:164-240) and, at setup time, init_fk20_multi_settings / toeplitz_part_1
// (src/setup/setup.c:197-330).
//
// Per blob:  bytes -> Fr -> DIT inverse NTT(4096) [blob order is already bit-reversed]
//            -> cells:  zero-extend, DIF NTT(8192) [output order = cell order] -> bytes
//            -> proofs: 64 circulant vectors -> DIF NTT(128) -> signed digits
//                       -> 128 fixed-base MSMs of 64 points (table over x_ext_fft columns)
//                       -> G1 DIF-FFT(128, w^-1) -> drop upper half -> G1 DIT-FFT(128)
//                       -> bit-reverse, batch-normalise, compress.
// The two G1 transforms use the DIF/DIT pair so that, as for Fr, no permutation pass exists.
#include "device.hpp"
#include "dev_inline.hpp"
#include "g1_28.hpp"
#include "g1_quad.hpp"
#include "g1_pipe.hpp"

namespace ckzg {
namespace dev {

__device__ __forceinline__ uint32_t brp7(uint32_t v) { return __brev(v) >> 25; }

__device__ __forceinline__ Fr ld_fr(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

__device__ __forceinline__ void st_fr(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// ------------------------------------------------------------------------------------------
// scalar side
// ------------------------------------------------------------------------------------------

// circ[v][i][j], i = offset 0..63, j = 0..127  (fk20.c:55-78 with r = 64, l = 64, d = 4095):
//   j == 0           -> poly[d - i]
//   j = 128 - k, k in 1..62 -> poly[d - i - 64k]
//   otherwise 0
__global__ void k_fk20_circulant(Fr *circ, const Fr *poly, size_t total) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total) return;
    size_t v = g >> 13;
    uint32_t i = (uint32_t)(g >> 7) & 63u, j = (uint32_t)g & 127u;
    Fr x = Fr::zero();
    const Fr *p = poly + v * N_BLOB;
    if (j == 0) {
        x = ld_fr(p + (N_BLOB - 1 - i));
    } else {
        uint32_t k = 128 - j;
        if (k >= 1 && k <= 62) x = ld_fr(p + (N_BLOB - 1 - i - 64 * k));
    }
    st_fr(circ + g, x);
}

// cfft[v][i][p] holds (after the DIF NTT, already scaled by 1/128) the value for MSM column
// j = brp7(p).  Recode it into digits[(v*128 + j)][w][i].
__global__ void k_fk20_digits(int16_t *digits, const Fr *cfft, size_t total, int wbits, int twin) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total) return;
    // lanes run over i fastest: the 2*twin digit rows a wave writes are then 128 contiguous bytes each (with p
    // fastest every 2-byte digit went to its own cache line: 3.3 ms per 2048 blobs); the 32-byte reads of cfft
    // become strided by 4 KB instead, one full sector each
    size_t v = g >> 13;
    uint32_t p = (uint32_t)(g >> 6) & 127u, i = (uint32_t)g & 63u;
    uint32_t j = brp7(p);
    uint32_t s[8];
    to_raw<FrParams>(s, ld_fr(cfft + (v << 13) + ((size_t)i << 7) + p));
    glv_digits(digits + ((v * 128 + j) * (size_t)(2 * twin) * 64) + i, 64, s, wbits, twin);
}

// ------------------------------------------------------------------------------------------
// G1 FFT of size 128 over a batch of vectors: one launch per butterfly stage
//
// A butterfly is one 255-bit scalar multiplication (~0.8 M multiply-adds) and two additions; the
// points it touches are 2 x 192 B.  So the stages go through global memory and the kernel keeps
// nothing but the ladder alive: no LDS, ~200 VGPRs, two waves per SIMD.  Lanes are ordered
// transform-fastest (lane = butterfly * nfft + transform), so a wave works on the SAME butterfly of
// 64 different transforms: one twiddle per wave (window digits uniform, no divergence) and the
// butterflies whose twiddle is 1 form whole waves that skip the ladder instead of idling in it.
// ------------------------------------------------------------------------------------------

// One record per twiddle w^(64 i), i = 0..128: the GLV halves {k1[4], k2[4]} (g1_28.hpp: glv_split)
// followed by their width-4 NAF digit strings (wnaf4_128), built once per context by k_glv_roots.
// ... and by their plain NAF strings (quad::naf2_128, 132 bytes each) for the two-wave pipeline of g1_pipe.hpp.
constexpr int TW_NAF2_OFF = 8 + 2 * (GLV_NAF_LEN / 4);         // word offset of the plain NAF strings
constexpr int TW_NAF2_STRIDE = 132;                            // bytes per string (quad::NAF2_LEN = 130, padded)
constexpr int TW_REC_WORDS = TW_NAF2_OFF + 2 * (TW_NAF2_STRIDE / 4);
static_assert(quad::NAF2_LEN <= TW_NAF2_STRIDE, "plain NAF string does not fit its slot");

__device__ __noinline__ void g1_mul_root(XYZZ28 &p, bool &inf, const uint32_t *rec) {
    XYZZ28 o;
    bool oi;
    const int8_t *naf = reinterpret_cast<const int8_t *>(rec + 8);
    xyzz28_mul_glv_naf(o, oi, p, inf, naf, naf + GLV_NAF_LEN);
    p = o;
    inf = oi;
}

// roots_glv[i] = GLV halves of w^(64 i), i = 0..128 (every twiddle of a size-128 transform).
// Stage s (butterfly span 2^s).  DIF: x = u + v, y = (u - v) w; DIT: v' = v w, x = u + v', y = u - v'
// (fft.c:164-185 computes the same butterflies recursively).  A DIF pass runs s = 7..1 (natural in,
// bit-reversed out), a DIT pass s = 1..7 (bit-reversed in, natural out).  Each stage is two launches,
// so that the ladder kernel holds nothing but its own state: k_g1_fft_twiddle multiplies slot i1 of
// every butterfly whose twiddle is not 1, k_g1_fft_addsub replaces (u, v) by (u + v, u - v); DIF runs
// addsub then twiddle, DIT twiddle then addsub.
// lane g -> (butterfly bf, transform f), transform-fastest over nfft rounded up to a multiple of 64 so
// that every wave belongs to exactly one butterfly (lanes with f >= nfft are idle)
__device__ __forceinline__ void butterfly_index(size_t g, uint32_t nfft, int s, uint32_t &f, int &j, int &i0, int &i1) {
    const uint32_t pad = (nfft + 63u) & ~63u;
    const uint32_t bf = (uint32_t)(g / pad);
    f = (uint32_t)(g - (size_t)bf * pad);
    const int half = 1 << (s - 1);
    j = (int)bf & (half - 1);
    i0 = (((int)bf >> (s - 1)) << s) + j;
    i1 = i0 + half;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_fft_twiddle(G1XYZZ *data, const uint32_t *roots_glv, uint32_t nfft, int s,
                                                       int inverse) {
    const size_t g = blockIdx.x * (size_t)64 + threadIdx.x;
    uint32_t f;
    int j, i0, i1;
    butterfly_index(g, nfft, s, f, j, i0, i1);
    if (f >= nfft || j == 0) return;
    int ridx = j * (N_EXT / (2 << (s - 1)));
    if (inverse) ridx = N_EXT - ridx;
    G1XYZZ *slot = data + (size_t)f * 128 + i1;
    bool vi;
    XYZZ28 v = xyzz28_from_xyzz(*slot, vi);
    g1_mul_root(v, vi, roots_glv + (size_t)(ridx / (N_EXT / 128)) * TW_REC_WORDS);
    *slot = xyzz28_to_xyzz(v, vi);
}

// The latency form for small batches: four lanes per butterfly (g1_quad.hpp), transforms padded to 16 per wave so
// that a wave (16 quads) still works on ONE twiddle.  The ladder is ~2.3x shorter; a batch of up to a few
// hundred blobs leaves most SIMDs idle anyway.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_fft_twiddle_quad(
    G1XYZZ *data, const uint32_t *roots_glv, uint32_t nfft, int s, int inverse) {
    const size_t q = (blockIdx.x * (size_t)64 + threadIdx.x) >> 2;
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t pad = (nfft + 15u) & ~15u;
    const uint32_t bf = (uint32_t)(q / pad);
    uint32_t f = (uint32_t)(q - (size_t)bf * pad);
    const int half = 1 << (s - 1);
    const int j = (int)bf & (half - 1);
    const int i1 = ((((int)bf >> (s - 1)) << s) + j) + half;
    if (j == 0 || bf >= 64u) return;   // uniform per wave (a wave serves one butterfly index)
    // Quads of the padding repeat the last transform instead of leaving the wave partly masked: measured on the
    // radix-4 ladder kernel, a wave with 4 of its 16 quads active runs the same ladder ~20 % SLOWER than a full one
    // (one-blob FK20 10.0 -> 8.2 ms, profiles/r02_quad_ab.txt).
    const bool live = f < nfft;
    if (!live) f = nfft - 1;
    int ridx = j * (N_EXT / (2 << (s - 1)));
    if (inverse) ridx = N_EXT - ridx;
    G1XYZZ *slot = data + (size_t)f * 128 + i1;
    bool vi;
    XYZZ28 v = xyzz28_from_xyzz(*slot, vi), o;
    bool oi;
    const uint32_t *rec = roots_glv + (size_t)(ridx / (N_EXT / 128)) * TW_REC_WORDS;
    const int8_t *naf = reinterpret_cast<const int8_t *>(rec + 8);
    quad::xyzz28_mul_glv_naf_quad(o, oi, v, vi, naf, naf + GLV_NAF_LEN, ql);
    if (ql == 0 && live) *slot = xyzz28_to_xyzz(o, oi);
}

__global__ __launch_bounds__(64) void k_g1_fft_addsub(G1XYZZ *data, uint32_t nfft, int s) {
    const size_t g = blockIdx.x * (size_t)64 + threadIdx.x;
    uint32_t f;
    int j, i0, i1;
    butterfly_index(g, nfft, s, f, j, i0, i1);
    if (f >= nfft) return;
    G1XYZZ *vec = data + (size_t)f * 128;
    bool ui, vi;
    XYZZ28 u = xyzz28_from_xyzz(vec[i0], ui), v = xyzz28_from_xyzz(vec[i1], vi);
    XYZZ28 x = u;
    bool xi = ui;
    xyzz28_add(x, xi, v, vi);
    vec[i0] = xyzz28_to_xyzz(x, xi);
    xyzz28_add(u, ui, xyzz28_neg(v), vi);
    vec[i1] = xyzz28_to_xyzz(u, ui);
}

// The last DIF stage, the truncation h[64..127] = 0 (fk20.c:264-266: in bit-reversed order these are
// the odd positions) and the first DIT stage fused: both slots of a pair receive u + v.
__global__ __launch_bounds__(64) void k_g1_fft_fold(G1XYZZ *data, size_t npairs) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= npairs) return;
    bool ui, vi;
    XYZZ28 u = xyzz28_from_xyzz(data[2 * g], ui), v = xyzz28_from_xyzz(data[2 * g + 1], vi);
    xyzz28_add(u, ui, v, vi);
    G1XYZZ r = xyzz28_to_xyzz(u, ui);
    data[2 * g] = r;
    data[2 * g + 1] = r;
}

// ------------------------------------------------------------------------------------------
// Radix-4 steps for small batches: two butterfly stages per launch pair, ONE ladder deep.
//
// In a radix-2 pass the twiddle multiplications of stage s-1 wait for those of stage s: 12 dependent ladders
// for the two transforms of FK20.  Written out on the four points of a 4-point group, every product of the
// second stage is a product of an INPUT combination with a product of two twiddles -- again a 128th root of
// unity -- so all of them can start at once: five independent ladders per group instead of four dependent
// ones (25 % more ladder work, half the depth; small batches leave the chip idle anyway).
//   DIF pair (s, s-1), H = 2^(s-1), Q = H/2, E = 128/2^s, points a0..a3 at t, t+Q, t+H, t+H+Q (t < Q):
//     L0 = (a0-a2) w^(tE)   L1 = (a1-a3) w^(tE+32)   L2 = (a0-a2) w^(3tE)   L3 = (a1-a3) w^(3tE+32)
//     L4 = ((a0+a2)-(a1+a3)) w^(2tE)
//     out: [t] = (a0+a2)+(a1+a3)   [t+Q] = L4   [t+H] = L0+L1   [t+H+Q] = L2-L3
//   DIT pair (s, s+1), H = 2^(s-1), g = t*(128/2^s)/2, points x0..x3 at t, t+H, t+2H, t+3H (t < H):
//     L0 = x1 w^(2g)   L1 = x2 w^g   L2 = x3 w^(3g)   L3 = x2 w^(g+32)   L4 = x3 w^(3g+32)
//     y0 = x0+L0, y1 = x0-L0, u = L1+L2, v = L3-L4
//     out: [t] = y0+u   [t+2H] = y0-u   [t+H] = y1+v   [t+3H] = y1-v
// (fft.c:164-185 computes the same butterflies recursively.)  The ladders are the four-lane form
// (g1_quad.hpp); lanes are ordered transform-fastest and padded to 16 transforms, so a wave (16 quads) works
// on ONE (group, ladder) pair: one twiddle, uniform NAF digits.
// ------------------------------------------------------------------------------------------

constexpr int R4_LADDERS = 32 * 5;  // per transform and radix-4 step

__device__ __forceinline__ void r4_group(int grp, int s, int dif, int &t, int &p0, int &p1, int &p2, int &p3) {
    if (dif) {
        const int Q = 1 << (s - 2), blk = grp / Q;
        t = grp - blk * Q;
        p0 = (blk << s) + t;
        p1 = p0 + Q;
        p2 = p0 + 2 * Q;
        p3 = p2 + Q;
    } else {
        const int H = 1 << (s - 1), blk = grp / H;
        t = grp - blk * H;
        p0 = (blk << (s + 1)) + t;
        p1 = p0 + H;
        p2 = p0 + 2 * H;
        p3 = p0 + 3 * H;
    }
}

// the additions around the radix-4 ladders: latency work, four lanes per addition (A/B: CKZG_R4_ONE_LANE_ADDS)
__device__ __forceinline__ void r4_add(XYZZ28 &a, bool &ainf, const XYZZ28 &b, bool binf, int ql) {
    quad::xyzz28_add_quad(a, ainf, b, binf, ql);
}

__device__ __forceinline__ int r4_exponent(int k, int t, int s, int dif) {
    const int E = 128 >> s;
    if (dif) {
        const int tE = t * E;
        switch (k) {
            case 0: return tE;
            case 1: return tE + 32;
            case 2: return 3 * tE;
            case 3: return 3 * tE + 32;
            default: return 2 * tE;
        }
    }
    const int g = t * E / 2;
    switch (k) {
        case 0: return 2 * g;
        case 1: return g;
        case 2: return 3 * g;
        case 3: return g + 32;
        default: return 3 * g + 32;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_fft_r4_ladder(
    G1XYZZ *lad, const G1XYZZ *data, const uint32_t *roots_glv, uint32_t nfft, int s, int dif, int inverse) {
    const size_t q = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 2;
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t pad = (nfft + 15u) & ~15u;
    const uint32_t ell = (uint32_t)(q / pad);
    uint32_t f = (uint32_t)(q - (size_t)ell * pad);
    if (ell >= (uint32_t)R4_LADDERS) return;
    // quads of the padding repeat the last transform instead of leaving the wave partly masked
    const bool live = f < nfft;
    if (!live) f = nfft - 1;
    const int grp = (int)ell / 5, k = (int)ell % 5;
    int t, p0, p1, p2, p3;
    r4_group(grp, s, dif, t, p0, p1, p2, p3);
    const G1XYZZ *vec = data + (size_t)f * 128;
    XYZZ28 v;
    bool vi;
    if (dif) {
        bool i0, i1, i2, i3;
        if (k == 0 || k == 2) {         // a0 - a2
            v = xyzz28_from_xyzz(vec[p0], vi);
            XYZZ28 b = xyzz28_from_xyzz(vec[p2], i2);
            r4_add(v, vi, xyzz28_neg(b), i2, ql);
        } else if (k == 1 || k == 3) {  // a1 - a3
            v = xyzz28_from_xyzz(vec[p1], vi);
            XYZZ28 b = xyzz28_from_xyzz(vec[p3], i3);
            r4_add(v, vi, xyzz28_neg(b), i3, ql);
        } else {                        // (a0 + a2) - (a1 + a3)
            v = xyzz28_from_xyzz(vec[p0], vi);
            XYZZ28 b = xyzz28_from_xyzz(vec[p2], i2);
            r4_add(v, vi, b, i2, ql);
            XYZZ28 c = xyzz28_from_xyzz(vec[p1], i1), d = xyzz28_from_xyzz(vec[p3], i3);
            r4_add(c, i1, d, i3, ql);
            r4_add(v, vi, xyzz28_neg(c), i1, ql);
        }
        (void)i0;
    } else {
        const int src = k == 0 ? p1 : ((k == 1 || k == 3) ? p2 : p3);
        v = xyzz28_from_xyzz(vec[src], vi);
    }
    int e = r4_exponent(k, t, s, dif) & 127;
    XYZZ28 o = v;
    bool oi = vi;
    if (e != 0) {
        const int rec_i = inverse ? 128 - e : e;
        const uint32_t *rec = roots_glv + (size_t)rec_i * TW_REC_WORDS;
        const int8_t *naf = reinterpret_cast<const int8_t *>(rec + 8);
        quad::xyzz28_mul_glv_naf_quad(o, oi, v, vi, naf, naf + GLV_NAF_LEN, ql);
    }
    if (ql == 0 && live) lad[(size_t)f * R4_LADDERS + ell] = xyzz28_to_xyzz(o, oi);
}

// one DPP quad per OUTPUT point (its two or three additions are latency, g1_quad.hpp): quad g -> (transform f,
// group, which of the four outputs)
__global__ __launch_bounds__(64) void k_g1_fft_r4_post(G1XYZZ *out, const G1XYZZ *data, const G1XYZZ *lad, uint32_t nfft,
                                                      int s, int dif) {
    const size_t g = (blockIdx.x * (size_t)64 + threadIdx.x) >> 2;
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t f = (uint32_t)(g >> 7);
    if (f >= nfft) return;
    const int grp = (int)((g >> 2) & 31), w = (int)(g & 3);
    int t, p0, p1, p2, p3;
    r4_group(grp, s, dif, t, p0, p1, p2, p3);
    const G1XYZZ *vec = data + (size_t)f * 128, *L = lad + (size_t)f * R4_LADDERS + grp * 5;
    XYZZ28 r;
    bool ri;
    int dst;
    if (dif) {
        if (w == 0) {          // (a0 + a2) + (a1 + a3)
            bool i1, i2, i3;
            r = xyzz28_from_xyzz(vec[p0], ri);
            XYZZ28 b = xyzz28_from_xyzz(vec[p2], i2);
            r4_add(r, ri, b, i2, ql);
            XYZZ28 c = xyzz28_from_xyzz(vec[p1], i1), d = xyzz28_from_xyzz(vec[p3], i3);
            r4_add(c, i1, d, i3, ql);
            r4_add(r, ri, c, i1, ql);
            dst = p0;
        } else if (w == 1) {   // L4
            r = xyzz28_from_xyzz(L[4], ri);
            dst = p1;
        } else if (w == 2) {   // L0 + L1
            bool bi;
            r = xyzz28_from_xyzz(L[0], ri);
            XYZZ28 b = xyzz28_from_xyzz(L[1], bi);
            r4_add(r, ri, b, bi, ql);
            dst = p2;
        } else {               // L2 - L3
            bool bi;
            r = xyzz28_from_xyzz(L[2], ri);
            XYZZ28 b = xyzz28_from_xyzz(L[3], bi);
            r4_add(r, ri, xyzz28_neg(b), bi, ql);
            dst = p3;
        }
    } else {
        // w = 0: y0 + u -> [t]; 1: y1 + v -> [t+H]; 2: y0 - u -> [t+2H]; 3: y1 - v -> [t+3H]
        bool ai, bi, ci;
        r = xyzz28_from_xyzz(vec[p0], ri);
        XYZZ28 a = xyzz28_from_xyzz(L[0], ai);
        r4_add(r, ri, (w & 1) ? xyzz28_neg(a) : a, ai, ql);              // y0 / y1
        XYZZ28 b = xyzz28_from_xyzz(L[(w & 1) ? 3 : 1], bi), c = xyzz28_from_xyzz(L[(w & 1) ? 4 : 2], ci);
        r4_add(b, bi, (w & 1) ? xyzz28_neg(c) : c, ci, ql);              // u = L1 + L2 / v = L3 - L4
        r4_add(r, ri, (w & 2) ? xyzz28_neg(b) : b, bi, ql);
        dst = w == 0 ? p0 : (w == 1 ? p1 : (w == 2 ? p2 : p3));
    }
    if (ql == 0) out[(size_t)f * 128 + dst] = xyzz28_to_xyzz(r, ri);
}

// ------------------------------------------------------------------------------------------
// Radix-8 steps for the smallest batches: THREE butterfly stages per launch pair, still one ladder deep.
//
// The six twiddle stages on either side of the truncation are two radix-8 steps instead of three radix-4 ones: four
// dependent ladders for the two transforms of FK20 instead of six, for 21 ladders per 8-point group instead of
// 2 x 5 per 2 x 4 points (+68 % ladder work -- free while the chip is mostly idle, i.e. up to 16 blobs with the
// three-wave ladder).  With N the block size after (DIT) / before (DIF) the step, Q = N/8, E = 128/N, and the eight
// points of a group at x_m = base + t + Q m (t < Q), sub-block c at base + c Q + t, r = bitrev3(c):
//   DIF (stages s, s-1, s-2; N = 2^s):     out[base + c Q + t] = w^(E r t) * sum_m x_m w^(16 m r)
//   DIT (stages s, s+1, s+2; N = 2^(s+2)): out[x_m] = sum_r w^(16 r m) * (w^(E r t) * Y_r),  Y_r = in[base + bitrev3(r) Q + t]
// (fft.c:164-185 computes the same butterflies recursively.)  Because w^64 = -1, the products that are really needed
// are, for both directions, L(r, j) = V(r, j) * w^(E r t + 16 r j):
//   r = 4: j = 0;   r = 2, 6: j = 0, 1;   r odd: j = 0..3            -> 1 + 4 + 16 = 21 ladders
//   DIF: V(4,0) = sum_m (-1)^m x_m;  V(r,j) = sum_a (-1)^a x_(j+2a) for r = 2, 6;  V(r,j) = x_j - x_(j+4) for r odd
//        out(r=0) = sum_m x_m;  out(4) = L(4,0);  out(2|6) = L(r,0) + L(r,1);  out(r odd) = sum_j L(r,j)
//   DIT: V(r,j) = Y_r;  out[x_m] = Y_0 + (-1)^m L(4,0) + (-1)^(m>>1) (L(2,m&1) + L(6,m&1)) + (-1)^(m>>2) sum_(r odd) L(r,m&3)
// The input combinations are formed by the ladder's doubler wave itself (at most seven quad additions in front of
// ~390 doubling steps); the sums around the ladders by k_g1_fft_r8_post, one quad per output point.
// ------------------------------------------------------------------------------------------

constexpr int R8_PER_GROUP = 21;
constexpr int R8_LADDERS = 16 * R8_PER_GROUP;  // per transform and radix-8 step

__device__ __forceinline__ int bitrev3(int v) { return ((v & 1) << 2) | (v & 2) | ((v >> 2) & 1); }

// ladder ell (0..20) of a group -> (r, j)
__device__ __forceinline__ void r8_ladder_rj(int ell, int &r, int &j) {
    if (ell == 0) {
        r = 4;
        j = 0;
    } else if (ell < 3) {
        r = 2;
        j = ell - 1;
    } else if (ell < 5) {
        r = 6;
        j = ell - 3;
    } else {
        r = 2 * ((ell - 5) >> 2) + 1;
        j = (ell - 5) & 3;
    }
}
// index of L(r, j) in a group's 21 ladder results
__device__ __forceinline__ int r8_ladder_index(int r, int j) {
    if (r == 4) return 0;
    if (r == 2) return 1 + j;
    if (r == 6) return 3 + j;
    return 5 + 4 * (r >> 1) + j;
}
// group geometry: s is the step's first stage (the highest for DIF, the lowest for DIT)
__device__ __forceinline__ void r8_group(int grp, int s, int dif, int &t, int &base, int &Q, int &E) {
    const int lgN = dif ? s : s + 2;
    Q = 1 << (lgN - 3);
    E = 128 >> lgN;
    const int blk = grp / Q;
    t = grp - blk * Q;
    base = blk << lgN;
}

// G1XYZZ -> raw records (and back at the end of the last step): one lane per point
__global__ __launch_bounds__(64) void k_g1_to_raw(uint32_t *raw, const G1XYZZ *in, size_t n) {
    const size_t g = blockIdx.x * (size_t)64 + threadIdx.x;
    if (g >= n) return;
    bool inf;
    const XYZZ28 v = xyzz28_from_xyzz(in[g], inf);
    quad::raw_store(raw + g * quad::RAW_WORDS, v, inf);
}

// two = 1 (batches of <= 8 transforms): a workgroup serves TWO ladder indices, eight transforms each -- quads 0..7 ladder
// 2 * blockIdx.x, quads 8..15 ladder 2 * blockIdx.x + 1 (g1_pipe.hpp: two twiddles per wave) -- so that a step is 168
// workgroups and each has a compute unit to itself.  two = 2 (9..16 transforms, launched with 128 threads): the two-wave
// form -- one adder wave for both chains, its sums parked in LDS -- whose 336 workgroups fit two to a compute unit.
__device__ __forceinline__ void r8_ladder_pipe_body(quad::PipeShared &sh, uint32_t *lad, const uint32_t *data, const uint32_t *roots_glv,
                                                    uint32_t nfft, int s, int dif, int inverse, int two) {
    constexpr int RW = quad::RAW_WORDS;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int quad_id = lane >> 2, ql = lane & 3;
    if (threadIdx.x == 0) {
        sh.produced = 0;
        sh.consumed[0] = 0;
        sh.consumed[1] = 0;
        sh.r2done = 0;
    }
    __syncthreads();
    // the (up to) two ladder indices of this workgroup and this quad's own (ell, f)
    uint32_t ell_a, ell_b, ell, f;
    if (two == 1) {
        ell_a = 2 * blockIdx.x;
        ell_b = ell_a + 1;
        ell = quad_id < 8 ? ell_a : ell_b;
        f = (uint32_t)(quad_id & 7);
    } else {
        const size_t q = blockIdx.x * (size_t)16 + quad_id;
        const uint32_t pad = (nfft + 15u) & ~15u;
        ell_a = ell_b = ell = (uint32_t)(q / pad);   // (pad is a multiple of 16: one ladder index per workgroup)
        f = (uint32_t)(q - (size_t)ell * pad);
    }
    if (ell_a >= (uint32_t)R8_LADDERS) return;   // uniform over the workgroup
    const bool live = f < nfft;
    if (!live) f = nfft - 1;   // quads of the padding repeat the last transform
    // twiddle exponents of the two ladder indices (uniform) and of this quad
    int e_ab[2];
    quad::NafMasks m_ab[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t el = h ? ell_b : ell_a;
        int r, j, t, base, Q, E;
        r8_ladder_rj((int)el % R8_PER_GROUP, r, j);
        r8_group((int)el / R8_PER_GROUP, s, dif, t, base, Q, E);
        e_ab[h] = (E * r * t + 16 * r * j) & 127;
        if (e_ab[h] != 0) {
            const uint32_t *rec = roots_glv + (size_t)(inverse ? 128 - e_ab[h] : e_ab[h]) * TW_REC_WORDS;
            const int8_t *naf1 = reinterpret_cast<const int8_t *>(rec + TW_NAF2_OFF);
            m_ab[h] = quad::naf_masks(naf1, naf1 + TW_NAF2_STRIDE);
        } else {
            m_ab[h] = quad::naf_masks_none();
        }
    }
    const bool any_ladder = e_ab[0] != 0 || e_ab[1] != 0;
    const bool my_ladder = (ell == ell_a ? e_ab[0] : e_ab[1]) != 0;   // (per quad) twiddle 1: the combination is the result
    int r, j, t, base, Q, E;
    r8_ladder_rj((int)ell % R8_PER_GROUP, r, j);
    r8_group((int)ell / R8_PER_GROUP, s, dif, t, base, Q, E);
    uint32_t *dst = lad + ((size_t)f * R8_LADDERS + ell) * RW;
    if (wave == 0) {
        const uint32_t *vec = data + ((size_t)f * 128 + base + t) * RW;
        XYZZ28 v;
        bool vi;
        if (dif) {
            // V(r, j): first term x_j, then alternating signs over the stride the residue class of r fixes
            const int step = r == 4 ? 1 : ((r & 1) ? 4 : 2), terms = 8 / step;
            v = quad::raw_load(vec + (size_t)Q * j * RW, vi);
            for (int a = 1; a < 8; a++) {   // (uniform trip count: the two halves of a wave may differ in `terms`)
                if (a >= terms) continue;
                bool bi;
                const XYZZ28 b = quad::raw_load(vec + (size_t)Q * (j + step * a) * RW, bi);
                quad::xyzz28_addsub_quad(v, vi, b, bi, (a & 1) != 0, ql);
            }
        } else {
            v = quad::raw_load(vec + (size_t)Q * bitrev3(r) * RW, vi);
        }
        if (!my_ladder && ql == 0 && live) quad::raw_store(dst, v, vi);
        if (!any_ladder) return;   // (the adder waves have left already)
        // quads without a ladder of their own ride along as points at infinity: the adders skip them
        quad::pipe_doubler(sh, v, vi || !my_ladder, m_ab[0], m_ab[1], quad_id, ql, two == 2);
    } else {
        if (!any_ladder) return;
        XYZZ28 o;
        bool oi = true;
        if (two == 2)
            quad::pipe_adder_dual(o, oi, sh, m_ab[0], quad_id, ql);
        else
            quad::pipe_adder(o, oi, sh, m_ab[0], m_ab[1], wave - 1, quad_id, ql);
        if (wave == 1 && my_ladder && ql == 0 && live) quad::raw_store(dst, o, oi);
    }
}

__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_fft_r8_ladder_pipe(
    uint32_t *lad, const uint32_t *data, const uint32_t *roots_glv, uint32_t nfft, int s, int dif, int inverse, int two) {
    __shared__ quad::PipeShared sh;
    r8_ladder_pipe_body(sh, lad, data, roots_glv, nfft, s, dif, inverse, two);
}
// the two-wave form with the whole register file per wave (one wave per SIMD: two workgroups per compute unit can
// never share a SIMD, and the adder wave -- one accumulator, the operand and the addition's temporaries -- does not spill)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_g1_fft_r8_ladder_pipe2(
    uint32_t *lad, const uint32_t *data, const uint32_t *roots_glv, uint32_t nfft, int s, int dif, int inverse) {
    __shared__ quad::PipeShared sh;
    r8_ladder_pipe_body(sh, lad, data, roots_glv, nfft, s, dif, inverse, 2);
}

// The same radix-8 ladders on ONE wave per 16 ladders (g1_quad.hpp's left-to-right ladder): for batches whose
// three-wave workgroups would no longer find a SIMD per wave (17 blobs upwards) the shorter chain of the pipeline is
// lost to sharing, but four ladder depths instead of six are not.
__device__ __forceinline__ void r8_ladder_body(uint32_t *lad, const uint32_t *data, const uint32_t *roots_glv, uint32_t nfft, int s, int dif,
                                               int inverse) {
    constexpr int RW = quad::RAW_WORDS;
    // (workgroups of four waves: the waves of ONE workgroup never share a SIMD, while single-wave workgroups are
    // doubled up on SIMDs that are already taken -- tools/ubench/wave_placement.hip)
    const size_t q = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 2;
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t pad = (nfft + 15u) & ~15u;
    const uint32_t ell = (uint32_t)(q / pad);
    uint32_t f = (uint32_t)(q - (size_t)ell * pad);
    if (ell >= (uint32_t)R8_LADDERS) return;
    const bool live = f < nfft;
    if (!live) f = nfft - 1;   // quads of the padding repeat the last transform
    const int grp = (int)ell / R8_PER_GROUP;
    int r, j, t, base, Q, E;
    r8_ladder_rj((int)ell % R8_PER_GROUP, r, j);
    r8_group(grp, s, dif, t, base, Q, E);
    const int e = (E * r * t + 16 * r * j) & 127;
    const uint32_t *vec = data + ((size_t)f * 128 + base + t) * RW;
    XYZZ28 v;
    bool vi;
    if (dif) {
        const int step = r == 4 ? 1 : ((r & 1) ? 4 : 2), terms = 8 / step;
        v = quad::raw_load(vec + (size_t)Q * j * RW, vi);
        for (int a = 1; a < terms; a++) {
            bool bi;
            const XYZZ28 b = quad::raw_load(vec + (size_t)Q * (j + step * a) * RW, bi);
            quad::xyzz28_addsub_quad(v, vi, b, bi, (a & 1) != 0, ql);
        }
    } else {
        v = quad::raw_load(vec + (size_t)Q * bitrev3(r) * RW, vi);
    }
    XYZZ28 o = v;
    bool oi = vi;
    if (e != 0) {
        const uint32_t *rec = roots_glv + (size_t)(inverse ? 128 - e : e) * TW_REC_WORDS;
        const int8_t *naf = reinterpret_cast<const int8_t *>(rec + 8);
        quad::xyzz28_mul_glv_naf_quad(o, oi, v, vi, naf, naf + GLV_NAF_LEN, ql);
    }
    if (ql == 0 && live) quad::raw_store(lad + ((size_t)f * R8_LADDERS + ell) * RW, o, oi);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_g1_fft_r8_ladder(
    uint32_t *lad, const uint32_t *data, const uint32_t *roots_glv, uint32_t nfft, int s, int dif, int inverse) {
    r8_ladder_body(lad, data, roots_glv, nfft, s, dif, inverse);
}
// the same with the whole register file per wave, for batches whose waves fit the chip once (<= 48 transforms)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_g1_fft_r8_ladder_full(
    uint32_t *lad, const uint32_t *data, const uint32_t *roots_glv, uint32_t nfft, int s, int dif, int inverse) {
    r8_ladder_body(lad, data, roots_glv, nfft, s, dif, inverse);
}

// one DPP quad per OUTPUT point: quad g -> (transform f, group, which of the eight outputs).  final_out: the last step
// of the transform writes G1XYZZ for the normalisation that follows.
__global__ __launch_bounds__(64) void k_g1_fft_r8_post(uint32_t *out, G1XYZZ *final_out, const uint32_t *data, const uint32_t *lad,
                                                      uint32_t nfft, int s, int dif) {
    constexpr int RW = quad::RAW_WORDS;
    const size_t g = (blockIdx.x * (size_t)64 + threadIdx.x) >> 2;
    const int ql = (int)(threadIdx.x & 3);
    const uint32_t f = (uint32_t)(g >> 7);
    if (f >= nfft) return;
    const int grp = (int)((g >> 3) & 15), w = (int)(g & 7);
    int t, base, Q, E;
    r8_group(grp, s, dif, t, base, Q, E);
    const uint32_t *vec = data + ((size_t)f * 128 + base + t) * RW, *L = lad + ((size_t)f * R8_LADDERS + grp * R8_PER_GROUP) * RW;
    XYZZ28 acc;
    bool ai;
    int dst;
    if (dif) {
        const int r = bitrev3(w);   // w = sub-block c
        dst = base + w * Q + t;
        if (r == 0) {
            acc = quad::raw_load(vec, ai);
            for (int m = 1; m < 8; m++) {
                bool bi;
                const XYZZ28 b = quad::raw_load(vec + (size_t)Q * m * RW, bi);
                quad::xyzz28_addsub_quad(acc, ai, b, bi, false, ql);
            }
        } else {
            const int terms = r == 4 ? 1 : ((r & 1) ? 4 : 2);
            acc = quad::raw_load(L + (size_t)r8_ladder_index(r, 0) * RW, ai);
            for (int j = 1; j < terms; j++) {
                bool bi;
                const XYZZ28 b = quad::raw_load(L + (size_t)r8_ladder_index(r, j) * RW, bi);
                quad::xyzz28_addsub_quad(acc, ai, b, bi, false, ql);
            }
        }
    } else {
        const int m = w;
        dst = base + t + Q * m;
        acc = quad::raw_load(vec, ai);   // Y_0 (bitrev3(0) = 0)
        // the seven other terms: (ladder index, sign)
        for (int k = 0; k < 7; k++) {
            int idx;
            bool neg;
            if (k == 0) {
                idx = 0;                                   // (-1)^m L(4,0)
                neg = (m & 1) != 0;
            } else if (k < 3) {
                idx = r8_ladder_index(k == 1 ? 2 : 6, m & 1);   // (-1)^(m>>1) L(r, m&1), r = 2, 6
                neg = ((m >> 1) & 1) != 0;
            } else {
                idx = r8_ladder_index(2 * (k - 3) + 1, m & 3);  // (-1)^(m>>2) L(r, m&3), r odd
                neg = ((m >> 2) & 1) != 0;
            }
            bool bi;
            const XYZZ28 b = quad::raw_load(L + (size_t)idx * RW, bi);
            quad::xyzz28_addsub_quad(acc, ai, b, bi, neg, ql);
        }
    }
    if (ql == 0) {
        if (final_out)
            final_out[(size_t)f * 128 + dst] = xyzz28_to_xyzz(acc, ai);
        else
            quad::raw_store(out + ((size_t)f * 128 + dst) * RW, acc, ai);
    }
}

// the last DIF stage, the truncation and the first DIT stage on raw records (k_g1_fft_fold): both slots of a pair
// receive u + v; one quad per pair
__global__ __launch_bounds__(64) void k_g1_fft_fold_raw(uint32_t *data, size_t npairs) {
    constexpr int RW = quad::RAW_WORDS;
    const size_t g = (blockIdx.x * (size_t)64 + threadIdx.x) >> 2;
    const int ql = (int)(threadIdx.x & 3);
    if (g >= npairs) return;
    bool ui, vi;
    XYZZ28 u = quad::raw_load(data + 2 * g * RW, ui);
    const XYZZ28 v = quad::raw_load(data + (2 * g + 1) * RW, vi);
    quad::xyzz28_addsub_quad(u, ui, v, vi, false, ql);
    if (ql == 0) {
        quad::raw_store(data + 2 * g * RW, u, ui);
        quad::raw_store(data + (2 * g + 1) * RW, u, ui);
    }
}

// Hand-over points (same-box A/Bs: profiles/r05_fk20_small_ab.txt).  Three-wave ladders while their 21 workgroups per
// transform and step mostly find a SIMD per wave (<= 16 transforms: 336 workgroups); radix-8 steps with the one-wave
// ladder while a step's waves fit the chip about once (<= 48 transforms: 1008 waves); radix-4 / radix-2 beyond.
static size_t r8_pipe_max_transforms() {
    static const size_t v = (size_t)ab_knob("CKZG_HIP_R8_PIPE_MAX", 16);
    return v;
}
static size_t r8_max_transforms() {
    static const size_t v = (size_t)ab_knob("CKZG_HIP_R8_FFT_MAX", 48);
    return v;
}

// Both transforms of FK20 for a small batch as radix-8 steps on raw records: inverse DIF (7,6,5), (4,3,2), the fold,
// forward DIT (2,3,4), (5,6,7).  d_u: nfft x 128 points in and out (G1XYZZ); d_a, d_b: nfft x 128 raw records;
// d_lad: nfft x R8_LADDERS raw records.
static int g1_fft_r8_fk20(DeviceCtx *ctx, G1XYZZ *d_u, uint32_t *d_a, uint32_t *d_b, uint32_t *d_lad, const uint32_t *d_glv,
                          size_t nfft) {
    const size_t pad = (nfft + 15) / 16 * 16;
    const dim3 lgrid((unsigned)(pad * R8_LADDERS / 16)), pgrid((unsigned)(nfft * 128 * 4 / 64)), block(64);
    hipLaunchKernelGGL(k_g1_to_raw, dim3((unsigned)((nfft * 128 + 63) / 64)), block, 0, ctx->stream, d_a, d_u, nfft * 128);
    uint32_t *cur = d_a, *nxt = d_b;
    const bool pipe = nfft <= r8_pipe_max_transforms();
    static const size_t two_max = (size_t)ab_knob("CKZG_HIP_R8_TWO_MAX", 8);
    static const bool dual = ab_knob("CKZG_HIP_R8_DUAL", 1) != 0;
    static const bool one_full = ab_knob("CKZG_HIP_R8_ONE_FULL", 1) != 0;
    static const unsigned wg1 = (unsigned)ab_knob("CKZG_HIP_LADDER_WG", 256);   // threads per workgroup of the one-wave ladder kernels
    auto step = [&](int s, int dif, int inverse, G1XYZZ *final_out) {
        if (pipe && nfft <= two_max) {   // two ladder indices per workgroup: 168 workgroups, one per compute unit
            hipLaunchKernelGGL(k_g1_fft_r8_ladder_pipe, dim3((unsigned)(R8_LADDERS / 2)), dim3(192), 0, ctx->stream, d_lad, cur, d_glv,
                               (uint32_t)nfft, s, dif, inverse, 1);
        } else if (pipe && dual)   // two-wave workgroups: two to a compute unit, a SIMD per wave
            hipLaunchKernelGGL(k_g1_fft_r8_ladder_pipe2, lgrid, dim3(128), 0, ctx->stream, d_lad, cur, d_glv, (uint32_t)nfft, s, dif, inverse);
        else if (pipe)
            hipLaunchKernelGGL(k_g1_fft_r8_ladder_pipe, lgrid, dim3(192), 0, ctx->stream, d_lad, cur, d_glv, (uint32_t)nfft, s, dif, inverse, 0);
        else if (one_full)
            hipLaunchKernelGGL(k_g1_fft_r8_ladder_full, dim3((unsigned)(pad * R8_LADDERS * 4 / wg1)), dim3(wg1), 0, ctx->stream, d_lad, cur,
                               d_glv, (uint32_t)nfft, s, dif, inverse);
        else
            hipLaunchKernelGGL(k_g1_fft_r8_ladder, dim3((unsigned)(pad * R8_LADDERS * 4 / wg1)), dim3(wg1), 0, ctx->stream, d_lad, cur, d_glv,
                               (uint32_t)nfft, s, dif, inverse);
        hipLaunchKernelGGL(k_g1_fft_r8_post, pgrid, block, 0, ctx->stream, nxt, final_out, cur, d_lad, (uint32_t)nfft, s, dif);
        uint32_t *x = cur;
        cur = nxt;
        nxt = x;
    };
    step(7, 1, 1, nullptr);
    step(4, 1, 1, nullptr);
    hipLaunchKernelGGL(k_g1_fft_fold_raw, dim3((unsigned)((nfft * 64 * 4 + 63) / 64)), block, 0, ctx->stream, cur, nfft * 64);
    step(2, 0, 0, nullptr);
    step(5, 0, 0, d_u);
    HIP_TRY(hipGetLastError());
    return 0;
}

// a radix-4 pass over stage pairs: DIF (s_hi, s_hi-1), ..., down to s_lo; DIT (s_lo, s_lo+1), ... up to s_hi.
// d_tmp: nfft x 128 points (the post kernel cannot write in place: every output reads all four inputs),
// d_lad: nfft x R4_LADDERS points.  The result ends in d_data.
static int g1_fft_r4_pairs(DeviceCtx *ctx, G1XYZZ *d_data, G1XYZZ *d_tmp, G1XYZZ *d_lad, const uint32_t *d_glv, size_t nfft,
                           bool dif, int s_from, int s_to, int inverse) {
    const size_t pad = (nfft + 15) / 16 * 16;
    // (the ladder kernel in workgroups of four waves: see r8_ladder_body)
    static const unsigned wg1 = (unsigned)ab_knob("CKZG_HIP_LADDER_WG", 256);
    const dim3 lgrid((unsigned)((pad * R4_LADDERS * 4 + wg1 - 1) / wg1)), lblock(wg1), pgrid((unsigned)(nfft * 128 * 4 / 64)), block(64);
    G1XYZZ *cur = d_data, *nxt = d_tmp;
    for (int s = s_from; dif ? s >= s_to + 1 : s + 1 <= s_to; s += dif ? -2 : 2) {
        // (round 5 first put the three-wave ladder of g1_pipe.hpp here -- radix-4 step 0.87 -> 0.53 ms, profiles/r05_fk20_small_ab.txt --
        // before the radix-8 forms took over every batch size at which a pipeline's waves find a SIMD each; the batches that
        // still come here, 49..128 transforms, are past that point)
        hipLaunchKernelGGL(k_g1_fft_r4_ladder, lgrid, lblock, 0, ctx->stream, d_lad, cur, d_glv, (uint32_t)nfft, s, dif ? 1 : 0,
                           inverse);
        hipLaunchKernelGGL(k_g1_fft_r4_post, pgrid, block, 0, ctx->stream, nxt, cur, d_lad, (uint32_t)nfft, s, dif ? 1 : 0);
        G1XYZZ *x = cur;
        cur = nxt;
        nxt = x;
    }
    if (cur != d_data)
        HIP_TRY(hipMemcpyAsync(d_data, cur, nfft * 128 * sizeof(G1XYZZ), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

static size_t r4_max_transforms() {
    // measured hand-over (tools/bench_fk20_sizes.py, profiles/r02_quad_ab.txt): radix-4 pairs win up to ~100
    // transforms, tie with the radix-2 four-lane form up to ~200, lose beyond
    static const size_t v = (size_t)ab_knob("CKZG_HIP_R4_FFT_MAX", 128);
    return v;
}

static int g1_fft_stages(DeviceCtx *ctx, G1XYZZ *d_data, const uint32_t *d_glv, size_t nfft, bool dif,
                         int s_from, int s_to, int inverse) {
    // dif: s runs downwards from s_from to s_to; dit: upwards
    const dim3 grid((unsigned)((nfft + 63) / 64 * 64)), block(64);  // 64 butterflies x padded transforms / 64 lanes
    // four lanes per butterfly while that is at most ~two waves per SIMD (64 butterflies x nfft x 4 lanes): up
    // to there the one-lane form is latency-bound (a lone wave issues a mad every 9.4 cycles) and the quad
    // form's 12 lane-products per doubling instead of 7 cost nothing
    static const size_t quad_max = (size_t)ab_knob("CKZG_HIP_QUAD_FFT_MAX", 512);
    const bool use_quad = nfft <= quad_max;
    const dim3 qgrid((unsigned)((nfft + 15) / 16 * 16 * 4));        // 64 butterflies x padded transforms x 4 / 64 lanes
    for (int s = s_from; dif ? s >= s_to : s <= s_to; s += dif ? -1 : 1) {
        if (dif) hipLaunchKernelGGL(k_g1_fft_addsub, grid, block, 0, ctx->stream, d_data, (uint32_t)nfft, s);
        if (s > 1 && use_quad)
            hipLaunchKernelGGL(k_g1_fft_twiddle_quad, qgrid, block, 0, ctx->stream, d_data, d_glv, (uint32_t)nfft, s, inverse);
        else if (s > 1)
            hipLaunchKernelGGL(k_g1_fft_twiddle, grid, block, 0, ctx->stream, d_data, d_glv, (uint32_t)nfft, s, inverse);
        if (!dif) hipLaunchKernelGGL(k_g1_fft_addsub, grid, block, 0, ctx->stream, d_data, (uint32_t)nfft, s);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// out48[v][brp7(k)] (or [k]) = compress(in[v][k])
__global__ void k_compress(uint8_t *out48, const G1Affine *in, size_t n, int out_brp) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint8_t buf[48];
    g1_compress_affine(buf, in[g]);
    // eip7594.c:133: the proofs leave in bit-reversed order within each blob
    const size_t o = out_brp ? ((g & ~(size_t)127) | brp7((uint32_t)g & 127u)) : g;
    for (int k = 0; k < 48; k++) out48[o * 48 + k] = buf[k];
}

// ------------------------------------------------------------------------------------------
// setup: x_ext_fft columns (setup.c:238-330)
// ------------------------------------------------------------------------------------------

// xin[off][k], off = 0..63: k < 63 -> monomial[4096 - 64 - 1 - off - 64k], else infinity
__global__ void k_xext_gather(G1XYZZ *xin, const G1Affine *monomial) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 64 * 128) return;
    uint32_t off = g >> 7, k = g & 127u;
    G1XYZZ v = G1XYZZ::inf();
    if (k < 63) v = xyzz_from_affine(monomial[N_BLOB - N_CELL - 1 - off - 64 * k]);
    xin[g] = v;
}

// xin[off][p] (DIF output: natural index j = brp7(p))  ->  cols[j*64 + off]
__global__ void k_xext_transpose(G1XYZZ *cols, const G1XYZZ *xin) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 64 * 128) return;
    uint32_t off = g >> 7, p = g & 127u;
    cols[brp7(p) * 64 + off] = xin[g];
}

// twiddle records (GLV halves + NAF digits) of the 129 twiddles w^(64 i) of the size-128 G1 transforms
__global__ void k_glv_roots(uint32_t *out, const Fr *roots) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > 128) return;
    uint32_t raw[8], k1[4], k2[4];
    to_raw<FrParams>(raw, ld_fr(roots + (size_t)g * (N_EXT / 128)));
    glv_split(raw, k1, k2);
    uint32_t *rec = out + (size_t)g * TW_REC_WORDS;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        rec[k] = k1[k];
        rec[4 + k] = k2[k];
    }
    int8_t *naf = reinterpret_cast<int8_t *>(rec + 8);
    wnaf4_128(naf, k1);
    wnaf4_128(naf + GLV_NAF_LEN, k2);
    int8_t *naf2 = reinterpret_cast<int8_t *>(rec + TW_NAF2_OFF);
    for (int i = 0; i < 2 * TW_NAF2_STRIDE; i++) naf2[i] = 0;
    quad::naf2_128(naf2, k1);
    quad::naf2_128(naf2 + TW_NAF2_STRIDE, k2);
}

static int ensure_roots_raw(DeviceCtx *ctx, uint32_t **out) {
    // allocated lazily and kept for the life of the context
    if (!ctx->d_roots_raw) {
        HIP_TRY(hipMalloc(&ctx->d_roots_raw, (size_t)129 * TW_REC_WORDS * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_glv_roots, dim3(3), dim3(64), 0, ctx->stream, ctx->d_roots_raw, ctx->d_roots);
        HIP_TRY(hipGetLastError());
    }
    *out = ctx->d_roots_raw;
    return 0;
}

int fk20_setup_device(DeviceCtx *ctx, const G1Affine *d_monomial, G1Affine *h_xext) {
    uint32_t *d_rr = nullptr;
    int rc = ensure_roots_raw(ctx, &d_rr);
    if (rc) return rc;
    DevTmp xin, cols, prefix;   // construction scratch: gone on every exit path
    const size_t npts = 64 * 128;
    HIP_TRY(hipMalloc(&xin.p, npts * sizeof(G1XYZZ)));
    HIP_TRY(hipMalloc(&cols.p, npts * sizeof(G1XYZZ)));
    HIP_TRY(hipMalloc(&prefix.p, npts * sizeof(Fp)));
    G1XYZZ *d_xin = static_cast<G1XYZZ *>(xin.p), *d_cols = static_cast<G1XYZZ *>(cols.p);
    Fp *d_prefix = static_cast<Fp *>(prefix.p);
    if (!ctx->d_xext) HIP_TRY(hipMalloc(&ctx->d_xext, npts * sizeof(G1Affine)));
    hipLaunchKernelGGL(k_xext_gather, dim3(npts / 256), dim3(256), 0, ctx->stream, d_xin, d_monomial);
    rc = g1_fft_stages(ctx, d_xin, d_rr, 64, /*dif=*/true, 7, 1, /*inverse=*/0);
    if (!rc) {
        hipLaunchKernelGGL(k_xext_transpose, dim3(npts / 256), dim3(256), 0, ctx->stream, d_cols, d_xin);
        if (hipGetLastError() != hipSuccess) rc = 2;
    }
    if (!rc) rc = batch_to_affine_device(ctx, ctx->d_xext, d_cols, d_prefix, npts);
    // the scratch is freed when this function returns: nothing may still be running on it, error or not
    if (dev::sync_stream(ctx->stream) != hipSuccess && !rc) rc = 2;
    if (rc) return rc;
    if (h_xext) HIP_TRY(hipMemcpy(h_xext, ctx->d_xext, npts * sizeof(G1Affine), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------
// proofs and cells
// ------------------------------------------------------------------------------------------

static size_t al(size_t v) { return (v + 255) / 256 * 256; }

// Scratch layout for FK20 over n polynomials (offsets into ctx->scratch at `base`)
struct Fk20Scratch {
    Fr *circ;          // [n][64][128]
    int16_t *digits;   // [n*128][nwin][64]
    G1XYZZ *u;         // [n][128]
    G1Affine *aff;     // [n][128]
    Fp *prefix;        // [n][128]
    G1XYZZ *r4_tmp;    // [n][128]          (small batches: radix-4 G1 FFT steps)
    G1XYZZ *r4_lad;    // [n][R4_LADDERS]
    uint32_t *r8_a, *r8_b, *r8_lad;   // raw records (g1_pipe.hpp): [n][128] twice, [n][R8_LADDERS] -- the smallest batches
    size_t bytes;
};

static Fk20Scratch fk20_layout(uint8_t *base, size_t n, int nwin) {
    Fk20Scratch s;
    size_t off = 0;
    s.circ = reinterpret_cast<Fr *>(base + off);
    off += al(n * 8192 * sizeof(Fr));
    s.digits = reinterpret_cast<int16_t *>(base + off);
    off += al(n * 128 * (size_t)nwin * 64 * sizeof(int16_t));
    s.u = reinterpret_cast<G1XYZZ *>(base + off);
    off += al(n * 128 * sizeof(G1XYZZ));
    s.aff = reinterpret_cast<G1Affine *>(base + off);
    off += al(n * 128 * sizeof(G1Affine));
    s.prefix = reinterpret_cast<Fp *>(base + off);
    off += al(n * 128 * sizeof(Fp));
    s.r4_tmp = s.r4_lad = nullptr;
    s.r8_a = s.r8_b = s.r8_lad = nullptr;
    if (n <= r8_max_transforms()) {
        const size_t rec = quad::RAW_WORDS * sizeof(uint32_t);
        s.r8_a = reinterpret_cast<uint32_t *>(base + off);
        off += al(n * 128 * rec);
        s.r8_b = reinterpret_cast<uint32_t *>(base + off);
        off += al(n * 128 * rec);
        s.r8_lad = reinterpret_cast<uint32_t *>(base + off);
        off += al(n * R8_LADDERS * rec);
    } else if (n <= r4_max_transforms()) {
        s.r4_tmp = reinterpret_cast<G1XYZZ *>(base + off);
        off += al(n * 128 * sizeof(G1XYZZ));
        s.r4_lad = reinterpret_cast<G1XYZZ *>(base + off);
        off += al(n * R4_LADDERS * sizeof(G1XYZZ));
    }
    s.bytes = off;
    return s;
}

// proofs from monomial coefficients, using scratch that the caller reserved at `base`
static int fk20_run(DeviceCtx *ctx, uint8_t *d_proofs, const Fr *d_poly, size_t n, uint8_t *base) {
    const FixedBaseTable &t = ctx->fk20;
    if (!t.d_table) return 2;
    uint32_t *d_rr = nullptr;
    int rc = ensure_roots_raw(ctx, &d_rr);
    if (rc) return rc;
    Fk20Scratch s = fk20_layout(base, n, t.nwin);
    size_t total = n * 8192;
    hipLaunchKernelGGL(k_fk20_circulant, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       s.circ, d_poly, total);
    HIP_TRY(hipGetLastError());
    // 64 forward NTTs of size 128 per blob, each output * 1/128 (fk20.c:199-209)
    rc = fr_ntt_batch(ctx, s.circ, n * 64, 7, /*dif=*/true, /*inverse=*/false, /*scale=*/true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fk20_digits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       s.digits, s.circ, total, t.wbits, t.twin);
    HIP_TRY(hipGetLastError());
    rc = msm_small_vectors_device(ctx, t, s.u, s.digits, n * 128, 64, 128);
    if (rc) return rc;
    // h = IFFT(u) truncated to its first 64 entries, proofs = FFT(h) (fk20.c:257-269): inverse DIF
    // stages 7..2, the fused stage pair around the truncation, forward DIT stages 2..7
    HIP_TRY(hipEventRecord(ctx->ev[7], ctx->stream));
    if (s.r8_lad) {
        // the smallest batches: stage triples on raw records (4 ladder depths)
        rc = g1_fft_r8_fk20(ctx, s.u, s.r8_a, s.r8_b, s.r8_lad, d_rr, n);
        if (rc) return rc;
    } else {
        const bool r4 = s.r4_lad != nullptr;   // small batch: stage pairs with independent ladders (6 ladder depths, not 12)
        rc = r4 ? g1_fft_r4_pairs(ctx, s.u, s.r4_tmp, s.r4_lad, d_rr, n, /*dif=*/true, 7, 2, /*inverse=*/1)
                : g1_fft_stages(ctx, s.u, d_rr, n, /*dif=*/true, 7, 2, /*inverse=*/1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_g1_fft_fold, dim3((unsigned)((n * 64 + 63) / 64)), dim3(64), 0, ctx->stream, s.u, n * 64);
        rc = r4 ? g1_fft_r4_pairs(ctx, s.u, s.r4_tmp, s.r4_lad, d_rr, n, /*dif=*/false, 2, 7, /*inverse=*/0)
                : g1_fft_stages(ctx, s.u, d_rr, n, /*dif=*/false, 2, 7, /*inverse=*/0);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(ctx->ev[8], ctx->stream));
    rc = batch_to_affine_device(ctx, s.aff, s.u, s.prefix, n * 128);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compress, dim3((unsigned)((n * 128 + 63) / 64)), dim3(64), 0, ctx->stream,
                       d_proofs, s.aff, n * 128, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// Low-latency cell proofs: no G1 FFT.
//
// The proof of cell k' is the commitment to the quotient of p(X) by X^64 - c, c = h^64 for the
// cell's coset shift h = w^brp7(k'), i.e. c = w_128^brp7(k'):
//        q(X) = sum_u a_u X^u,   a_u = sum_{k>=0} c^k p[u + 64(k+1)].
// For a fixed u the 128 values of a_u (one per c = w_128^m) are the size-128 DFT of the sequence
// s_u[k] = p[u + 64(k+1)], so all scalars come from 4096 small Fr NTTs and every proof is one
// 4096-point fixed-base MSM over the monomial setup points -- 128 independent, perfectly regular
// MSMs and zero sequential G1 work.  This does ~10x the point additions of FK20 (fk20.c:139-286)
// but has no 7-stage x 255-bit scalar-multiplication dependency chain, so one blob takes a few
// milliseconds instead of the ~28 ms FK20 needs for any small batch; FK20 stays the high-throughput path
// for large batches.
// With the DIF transform the value for output cell k' lands at position k' (brp7(brp7(k')) = k').
// ------------------------------------------------------------------------------------------

__global__ void k_direct_fill(Fr *a, const Fr *poly, size_t total) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total) return;
    size_t v = g >> 19;
    uint32_t u = (uint32_t)(g >> 7) & 4095u, k = (uint32_t)g & 127u;
    uint32_t idx = u + 64u * (k + 1u);
    Fr x = Fr::zero();
    if (k <= 62 && idx < (uint32_t)N_BLOB) x = ld_fr(poly + v * N_BLOB + idx);
    st_fr(a + g, x);
}

// a[v][u][k'] -> digits[(v*128 + k')][w][u]
__global__ void k_direct_digits(int16_t *digits, const Fr *a, size_t total, int wbits, int twin) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total) return;
    uint32_t u = (uint32_t)g & 4095u, kp = (uint32_t)(g >> 12) & 127u;
    size_t v = g >> 19;
    uint32_t s[8];
    to_raw<FrParams>(s, ld_fr(a + (((v << 12) + u) << 7) + kp));
    glv_digits(digits + ((v * 128 + kp) * (size_t)(2 * twin) * N_BLOB) + u, N_BLOB, s, wbits, twin);
}

struct DirectScratch {
    Fr *a;            // [n][4096][128]
    int16_t *digits;  // [n*128][nwin][4096]
    G1XYZZ *partials;
    size_t bytes;
};

static DirectScratch direct_layout(uint8_t *base, size_t n, const FixedBaseTable &t) {
    DirectScratch s;
    size_t off = 0;
    s.a = reinterpret_cast<Fr *>(base + off);
    off += al(n * 4096 * 128 * sizeof(Fr));
    s.digits = reinterpret_cast<int16_t *>(base + off);
    off += al(n * 128 * (size_t)t.nwin * 4096 * sizeof(int16_t));
    s.partials = reinterpret_cast<G1XYZZ *>(base + off);
    off += al(msm_partials_needed(t, n * 128) * sizeof(G1XYZZ));
    s.bytes = off;
    return s;
}

static int direct_run(DeviceCtx *ctx, uint8_t *d_proofs, const Fr *d_poly, size_t n, uint8_t *base) {
    const FixedBaseTable &t = ctx->mono;
    DirectScratch s = direct_layout(base, n, t);
    size_t total = n << 19;
    hipLaunchKernelGGL(k_direct_fill, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, s.a,
                       d_poly, total);
    HIP_TRY(hipGetLastError());
    int rc = fr_ntt_batch(ctx, s.a, n * 4096, 7, /*dif=*/true, /*inverse=*/false, /*scale=*/false);
    if (rc) return rc;
    hipLaunchKernelGGL(k_direct_digits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       s.digits, s.a, total, t.wbits, t.twin);
    HIP_TRY(hipGetLastError());
    return msm_from_digits_device(ctx, t, d_proofs, s.digits, s.partials, n * 128);
}

static bool use_direct(const DeviceCtx *ctx, size_t n) {
    return ctx->mono.d_table != nullptr && n <= (size_t)ctx->direct_max;
}

int fk20_proofs_device(DeviceCtx *ctx, uint8_t *d_proofs, const Fr *d_poly_monomial, size_t n) {
    if (n == 0) return 0;
    int rc;
    if (use_direct(ctx, n)) {
        DirectScratch probe = direct_layout(nullptr, n, ctx->mono);
        rc = scratch_reserve(ctx, probe.bytes);
        if (rc) return rc;
        rc = direct_run(ctx, d_proofs, d_poly_monomial, n, static_cast<uint8_t *>(ctx->scratch.ptr));
    } else {
        Fk20Scratch probe = fk20_layout(nullptr, n, ctx->fk20.nwin);
        rc = scratch_reserve(ctx, probe.bytes);
        if (rc) return rc;
        rc = fk20_run(ctx, d_proofs, d_poly_monomial, n, static_cast<uint8_t *>(ctx->scratch.ptr));
    }
    if (rc) return rc;
    HIP_TRY(dev::sync_stream(ctx->stream));
    return 0;
}

__global__ void k_bad_to_status(uint8_t *status, const uint32_t *bad, size_t n) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g < n) status[g] = bad[g] ? 1 : 0;
}

// ---- the three enqueue-only stages of compute_cells_and_kzg_proofs (no waits): callers that pipeline
// ---- copies against them (ckzg_api.hip) order them with events on ctx->stream ----

// blobs -> monomial coefficients d_poly[k][4096] (kept for the proof stage) and, if d_cells != nullptr,
// the 128 cells per blob; d_ext is k x 8192 Fr of scratch (only touched when cells are wanted);
// d_bad[k] must have been zeroed on the stream.
int cells_stage_enqueue(DeviceCtx *ctx, uint8_t *d_cells, Fr *d_poly, Fr *d_ext, uint32_t *d_bad,
                        const uint8_t *d_blobs, size_t k) {
    if (k == 0) return 0;
    int rc = bytes_to_fr_batch(ctx, d_poly, d_bad, d_blobs, k * N_BLOB, N_BLOB);
    if (rc) return rc;
    // blob = evaluations in bit-reversed order: DIT inverse transform gives the coefficients
    // (poly_lagrange_to_monomial, poly.c:58-80, without its permutation pass)
    rc = fr_ntt_batch(ctx, d_poly, k, 12, /*dif=*/false, /*inverse=*/true, /*scale=*/true);
    if (rc) return rc;
    if (d_cells) {
        rc = zero_extend_batch(ctx, d_ext, d_poly, k, N_BLOB, N_EXT);
        if (rc) return rc;
        rc = fr_ntt_batch(ctx, d_ext, k, 13, /*dif=*/true, /*inverse=*/false, /*scale=*/false);
        if (rc) return rc;
        rc = fr_to_bytes_batch(ctx, d_cells, d_ext, k * N_EXT);
        if (rc) return rc;
    }
    return 0;
}

bool proofs_use_direct(const DeviceCtx *ctx, size_t n) { return use_direct(ctx, n); }

size_t proofs_scratch_bytes(const DeviceCtx *ctx, size_t k, bool direct) {
    return direct ? direct_layout(nullptr, k, ctx->mono).bytes : fk20_layout(nullptr, k, ctx->fk20.nwin).bytes;
}

// d_poly[k][4096] -> 128 proofs per blob; `scratch` holds proofs_scratch_bytes(ctx, k, direct)
int proofs_stage_enqueue(DeviceCtx *ctx, uint8_t *d_proofs, const Fr *d_poly, size_t k, uint8_t *scratch, bool direct) {
    if (k == 0) return 0;
    return direct ? direct_run(ctx, d_proofs, d_poly, k, scratch) : fk20_run(ctx, d_proofs, d_poly, k, scratch);
}

int bad_to_status_enqueue(DeviceCtx *ctx, uint8_t *d_status, const uint32_t *d_bad, size_t k) {
    if (k == 0) return 0;
    hipLaunchKernelGGL(k_bad_to_status, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, ctx->stream, d_status, d_bad, k);
    HIP_TRY(hipGetLastError());
    return 0;
}

int cells_and_proofs_device(DeviceCtx *ctx, uint8_t *d_cells, uint8_t *d_proofs, uint8_t *d_status,
                            const uint8_t *d_blobs, size_t n, uint8_t *h_cells, bool *cells_copied) {
    if (cells_copied) *cells_copied = false;
    if (n == 0) return 0;
    // process in chunks so scratch stays bounded (about 1.2 MB per blob); the G1 FFT launches one
    // wave per blob, so a chunk should be several times the chip's 1024 SIMDs to keep them busy
    const size_t CH = 4096;
    size_t m = n < CH ? n : CH;
    size_t poly_b = al(m * N_BLOB * sizeof(Fr)), ext_b = al(m * N_EXT * sizeof(Fr)), bad_b = al(m * 4);
    const bool direct = d_proofs != nullptr && use_direct(ctx, n);
    size_t proof_scratch = proofs_scratch_bytes(ctx, m, direct);
    int rc = scratch_reserve(ctx, poly_b + ext_b + bad_b + proof_scratch);
    if (rc) return rc;
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch.ptr);
    Fr *d_poly = reinterpret_cast<Fr *>(base);
    Fr *d_ext = reinterpret_cast<Fr *>(base + poly_b);
    uint32_t *d_bad = reinterpret_cast<uint32_t *>(base + poly_b + ext_b);
    uint8_t *fk_base = base + poly_b + ext_b + bad_b;
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (n <= 64 && d_cells && d_proofs) {
        // Latency path (the one-blob call): cells and proofs both hang off the coefficients and share nothing
        // else, so the three small kernels of the cells run on the slot's second stream underneath the first
        // kernels of the proof path instead of in front of them.
        for (int i = 0; i < 2; i++) {
            if (!ctx->stage_ev[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
        }
        HIP_TRY(hipMemsetAsync(d_bad, 0, n * 4, ctx->stream));
        rc = cells_stage_enqueue(ctx, nullptr, d_poly, d_ext, d_bad, d_blobs, n);  // coefficients only
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ctx->stage_ev[0], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->stage_ev[0], 0));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->copy_stream;  // the slot is leased exclusively: the launchers below follow ctx->stream
        rc = zero_extend_batch(ctx, d_ext, d_poly, n, N_BLOB, N_EXT);
        if (!rc) rc = fr_ntt_batch(ctx, d_ext, n, 13, /*dif=*/true, /*inverse=*/false, /*scale=*/false);
        if (!rc) rc = fr_to_bytes_batch(ctx, d_cells, d_ext, n * N_EXT);
        ctx->stream = main_stream;
        if (rc) return rc;
        if (h_cells) {   // the bulk of the output crosses PCIe underneath the proof kernels
            HIP_TRY(hipMemcpyAsync(h_cells, d_cells, n * (size_t)N_EXT * 32, hipMemcpyDeviceToHost, ctx->copy_stream));
            if (cells_copied) *cells_copied = true;
        }
        HIP_TRY(hipEventRecord(ctx->stage_ev[1], ctx->copy_stream));
        rc = proofs_stage_enqueue(ctx, d_proofs, d_poly, n, fk_base, direct);
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->stage_ev[1], 0));
        if (d_status) {
            rc = bad_to_status_enqueue(ctx, d_status, d_bad, n);
            if (rc) return rc;
        }
        n = 0;  // done: skip the chunk loop below
    }
    for (size_t off = 0; off < n; off += CH) {
        size_t k = n - off < CH ? n - off : CH;
        HIP_TRY(hipMemsetAsync(d_bad, 0, k * 4, ctx->stream));
        rc = cells_stage_enqueue(ctx, d_cells ? d_cells + off * (size_t)N_EXT * 32 : nullptr, d_poly, d_ext, d_bad,
                                 d_blobs + off * (size_t)N_BLOB * 32, k);
        if (rc) return rc;
        if (d_proofs) {
            rc = proofs_stage_enqueue(ctx, d_proofs + off * 128 * 48, d_poly, k, fk_base, direct);
            if (rc) return rc;
        }
        if (d_status) {
            rc = bad_to_status_enqueue(ctx, d_status + off, d_bad, k);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(dev::sync_stream(ctx->stream));
    float ms;
    if (hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[4]) == hipSuccess) ctx->last_ms[3] = ms;
    if (d_proofs && !direct && hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6]) == hipSuccess) ctx->last_ms[1] = ms;
    if (d_proofs && direct && hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ms[1] = ms;
    ctx->last_ms[4] = -1;
    if (d_proofs && !direct && hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[8]) == hipSuccess) ctx->last_ms[4] = ms;
    (void)hipGetLastError();
    return 0;
}

// Kernel times of the most recent FK20 run on this slot, for callers that enqueue it themselves (the recover batch):
// last_ms[1] = k_msm_small, last_ms[4] = the two G1 FFTs (ckzg_hip_last_kernel_ms).  Stream must be idle.
void fk20_collect_times(DeviceCtx *ctx) {
    float ms;
    ctx->last_ms[1] = ctx->last_ms[4] = -1;
    if (hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6]) == hipSuccess) ctx->last_ms[1] = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[8]) == hipSuccess) ctx->last_ms[4] = ms;
    (void)hipGetLastError();
}

}  // namespace dev
}  // namespace ckzg
