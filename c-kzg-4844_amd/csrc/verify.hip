// verify.hip -- the data-parallel per-blob / per-point work of the verification and recovery
// paths, batched on the GPU:
//   * barycentric evaluation of a blob polynomial at a challenge point
//     (evaluate_polynomial_in_evaluation_form + fr_batch_inv, src/eip4844/eip4844.c:80-106,192-240)
//   * G1 point validation: decompress (Fp square root), curve and subgroup checks
//     (validate_kzg_g1, src/common/bytes.c:81-95)
//   * variable-base linear combinations sum_i k_i P_i (g1_lincomb_naive / g1_lincomb_fast over
//     proofs and commitments, src/eip4844/eip4844.c:731-746, src/eip7594/eip7594.c:530,807,926)
//   * element-wise Fr helpers used by recover_cells (src/eip7594/recovery.c:281,322-328)
#include "device.hpp"
#include "dev_inline.hpp"
#include "g1_28.hpp"
#include "g1_quad.hpp"
#include "fr_inv.hpp"
#include "fr29.hpp"

namespace ckzg {
namespace dev {

__device__ __forceinline__ Fr vld_fr(const Fr *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

__device__ __forceinline__ void vst_fr(Fr *p, const Fr &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// safegcd (fr_inv.hpp): ~10x fewer instructions than the 255-squaring Fermat ladder
__device__ __noinline__ Fr fr_inv_dev(const Fr &a) { return fr_inv_safegcd(a); }

// ------------------------------------------------------------------------------------------
// barycentric evaluation: one 256-thread workgroup per polynomial
//   y = (z^4096 - 1)/4096 * sum_i p_i w_i / (z - w_i),   or p_m if z == w_m
// The products run on nine 29-bit limbs (fr29.hpp: 162 multiply-adds a product where the 32-bit CIOS form pays two
// carry instructions per multiply-add), a term p_i * (w_i / (z - w_i)) comes out in the library's radix because the
// radix travels with the operand, and the eight waves of a workgroup share ONE inversion: a wave executes an
// inversion's instructions whether one lane needs it or 64, so what Montgomery's trick has to spread is the number
// of WAVES that run one -- lane l of the first wave inverts the product of the eight waves' lane-l products, every
// thread finds its own inverse in shared memory (ev29::invert_across).
// ------------------------------------------------------------------------------------------

constexpr int EV_THREADS = 64 * ev29::WAVES;
constexpr int EV_PER = ev29::PER;
static_assert(EV_THREADS * EV_PER == N_BLOB, "one workgroup per polynomial");
constexpr size_t EVAL_ONE_WAVE_FROM = 256;   // polynomials per launch from which k_eval_tree runs one wave each
constexpr size_t EVAL_TWO_PER_SIMD_FROM = 1024;   // ... above which a compute unit takes eight of them at a time

__device__ __noinline__ Fr29 fr29_inv_dev(const Fr29 &a) { return fr29_inv(a); }

// the evaluation domain in fr29.hpp's form (canonical, radix 2^261), made once per context: out[i] = brp_roots[i],
// i < 4096 (k_eval_barycentric), then tab[m] = 1 / brp_roots[2 m], m < 2048 (k_eval_tree)
__global__ void k_roots29(uint32_t *out, const Fr *brp_roots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N_BLOB + N_BLOB / 2) return;
    Fr29 r;
    if (i < N_BLOB) r = fr29_from_fr(vld_fr(brp_roots + i));
    else r = fr29_canonical<0>(fr29_inv_dev(fr29_from_fr(vld_fr(brp_roots + 2 * (i - N_BLOB)))));
#pragma unroll
    for (int k = 0; k < 9; k++) out[(size_t)i * 9 + k] = r.l[k];
}

int roots29_build(DeviceCtx *ctx, uint32_t *d_out) {
    hipLaunchKernelGGL(k_roots29, dim3((ROOTS29_ENTRIES + 255) / 256), dim3(256), 0, ctx->stream, d_out, ctx->d_brp_roots);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// evaluation without inversions (fr29.hpp: ev29::tree_node says why), straight from the blobs' bytes: the leaves are
// the big-endian 32-byte field elements as they lie in the blob (blob_to_polynomial, blob.c:31-38, folded in --
// verification needs the converted polynomial for nothing else, so the conversion kernel, its 131 KB per blob of
// writes and their re-read are gone); an element >= r sets bad_out[blob] (bytes.c:52-70).
// A thread folds 2^LOG_PER consecutive leaves depth first, then six levels of lane exchanges inside the wave -- both
// lanes of a pair compute the parent, no lane idles at a branch -- and, where a polynomial has more than one wave, its
// first thread folds the waves' values.  LOG_PER = 6: ONE wave per polynomial, eight polynomials per workgroup turn --
// the throughput form (2 x 4095 products + 11 squarings per polynomial); LOG_PER = 4: four waves per polynomial, the
// latency form for small batches.
// What the first form of this kernel lost two thirds of its time to was waiting, not arithmetic: a product was a
// CALL, a call drains every outstanding load (the callee opens with s_waitcnt 0), so each leaf's load was waited for
// in full, one after the other (2 KB apart per lane: never a coalesced request).  Now
//   * a thread's leaves arrive eight at a time (256 B) in registers, the NEXT eight requested before the current
//     eight are folded, and the fold of eight leaves (seven nodes, fourteen products) contains no call;
//   * the per-node constants tab[m] = 1 / brp_roots[2 m] (2048 x 36 B) live in LDS, filled once per workgroup and
//     laid out so that the lanes of a wave read consecutive words at the leaf level: their reads are counted by
//     lgkmcnt and do not wait for the blob's bytes in flight (vmcnt);
//   * the upper levels of a thread's subtree (one node per 16 / 32 / 64 leaves) are folded with called products right
//     after the wait for the tile, when nothing is in flight.
// ------------------------------------------------------------------------------------------
constexpr int EV_TAB_N = N_BLOB / 2;
__device__ __forceinline__ int ev_tab_pos(int m) { return ((m & 31) << 6) | (m >> 5); }
__device__ __forceinline__ Fr29 ev_tab_get(const uint32_t (*tab_s)[EV_TAB_N], int m) {
    const int pos = ev_tab_pos(m);
    Fr29 r;
#pragma unroll
    for (int w = 0; w < 9; w++) r.l[w] = tab_s[w][pos];
    return r;
}
// leaf k of a tile (two uint4 per leaf, as loaded)
__device__ __forceinline__ Fr29 ev_leaf(const uint4 *tile, int k, uint32_t &bad) {
    const uint4 a = tile[2 * k], b = tile[2 * k + 1];
    const uint32_t s[8] = {__builtin_bswap32(b.w), __builtin_bswap32(b.z), __builtin_bswap32(b.y), __builtin_bswap32(b.x),
                           __builtin_bswap32(a.w), __builtin_bswap32(a.z), __builtin_bswap32(a.y), __builtin_bswap32(a.x)};
    return ev29::tree_leaf_from_words(s, bad);
}
// the level-L node over leaves [K0, K0 + 2^L) of the tile; m0 = index of the tile's first level-1 node among the
// polynomial's (its first leaf / 2)
template <int L, int K0>
__device__ __forceinline__ Fr29 ev_tile_node(const uint4 *tile, const uint32_t (*tab_s)[EV_TAB_N], const Fr29 *zp, int m0,
                                             uint32_t &bad) {
    if constexpr (L == 0) {
        return ev_leaf(tile, K0, bad);
    } else {
        const Fr29 e = ev_tile_node<L - 1, K0>(tile, tab_s, zp, m0, bad);
        const Fr29 o = ev_tile_node<L - 1, K0 + (1 << (L - 1))>(tile, tab_s, zp, m0, bad);
        // the children's X: z^(2^(L-1)) * tab[(first leaf of the node) >> L]
        const Fr29 x = fr29_mul_inline(zp[L - 1], ev_tab_get(tab_s, (m0 >> (L - 1)) + (K0 >> L)));
        return ev29::tree_combine<L - 1, true>(e, o, x);
    }
}

constexpr int EV_TILE_LOG = 2, EV_TILE = 1 << EV_TILE_LOG;   // leaves per tile: 128 B per thread in flight, 128 B in use

// One workgroup per compute unit at a time (the table is 72 of its 160 KB of LDS; two workgroups of 75 KB were never
// co-resident when tried): the throughput form brings its two waves per SIMD in ONE workgroup of eight waves.
template <int LOG_PER, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void k_eval_tree(Fr *y_out, uint32_t *bad_out, const uint8_t *blobs, const Fr *zs,
                                                          const uint32_t *tab_words, unsigned n) {
    constexpr int PER = 1 << LOG_PER, T = N_BLOB >> LOG_PER, W = T / 64, POLYS = THREADS / T, TILES = PER / EV_TILE;
    static_assert(LOG_PER == 6 || LOG_PER == 4, "one wave or four per polynomial");
    __shared__ uint32_t tab_s[9][EV_TAB_N];
    __shared__ uint32_t sh[9][4];
    __shared__ uint32_t zp_s[THREADS / 64][LOG_PER][9];   // per wave: z^(2^l), l >= EV_TILE_LOG (the upper levels' factors)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < EV_TAB_N; e += THREADS) {
        const int pos = ev_tab_pos(e);
#pragma unroll
        for (int w = 0; w < 9; w++) tab_s[w][pos] = tab_words[(size_t)e * 9 + w];
    }
    __syncthreads();
    const Fr29 *tab = reinterpret_cast<const Fr29 *>(tab_words);
    for (unsigned first = blockIdx.x * POLYS; first < n; first += gridDim.x * POLYS) {
        const unsigned poly = first + (POLYS > 1 ? (unsigned)wave : 0u);
        if (POLYS > 1 && poly >= n) break;   // (the whole wave; no barrier below in this form)
        const int t = POLYS > 1 ? lane : tid;   // the thread's index within its polynomial
        const uint4 *pb = reinterpret_cast<const uint4 *>(blobs + (size_t)poly * (N_BLOB * 32)) + (size_t)t * (PER * 2);
        uint4 cur[2 * EV_TILE], nxt[2 * EV_TILE];
#pragma unroll
        for (int k = 0; k < 2 * EV_TILE; k++) cur[k] = pb[k];
        Fr29 zp[EV_TILE_LOG], zc;
        {
            // every lane computes the powers (uniform per wave); the upper ones are parked in LDS
            Fr29 p = fr29_from_fr(vld_fr(zs + poly));
#pragma unroll
            for (int l = 0; l < LOG_PER; l++) {
                if (l < EV_TILE_LOG) {
                    zp[l] = p;
                } else if (lane == 0) {
#pragma unroll
                    for (int w = 0; w < 9; w++) zp_s[wave][l][w] = p.l[w];
                }
                p = fr29_mul(p, p);
            }
            zc = p;   // z^(2^LOG_PER)
        }
        auto zp_upper = [&](int l) {
            Fr29 r;
#pragma unroll
            for (int w = 0; w < 9; w++) r.l[w] = zp_s[wave][l][w];
            return r;
        };
        uint32_t bad = 0;
        Fr29 s2, s3, s4, s5, v;
        s2 = s3 = s4 = s5 = v = fr29_const(FR29_ONE);
#pragma unroll 1
        for (int j = 0; j < TILES; j++) {
            if (j + 1 < TILES) {
#pragma unroll
                for (int k = 0; k < 2 * EV_TILE; k++) nxt[k] = pb[2 * EV_TILE * (j + 1) + k];
            }
            // four leaves -> their level-2 node, no call inside
            v = ev_tile_node<EV_TILE_LOG, 0>(cur, tab_s, zp, t * (PER / 2) + (EV_TILE / 2) * j, bad);
#pragma unroll
            for (int k = 0; k < 2 * EV_TILE; k++) cur[k] = nxt[k];
            // the nodes above it that this tile completes (levels 3 .. LOG_PER), with called products: the tile just
            // requested has the whole next fold to arrive
            const int m2 = t * (PER / 8) + (j >> 1);   // index of the level-3 node this tile belongs to
            if (j & 1) {
                v = ev29::tree_combine<2>(s2, v, fr29_mul(zp_upper(2), ev_tab_get(tab_s, m2)));
                if (j & 2) {
                    v = ev29::tree_combine<3>(s3, v, fr29_mul(zp_upper(3), ev_tab_get(tab_s, m2 >> 1)));
                    if constexpr (LOG_PER > 4) {
                        if (j & 4) {
                            v = ev29::tree_combine<4>(s4, v, fr29_mul(zp_upper(4), ev_tab_get(tab_s, m2 >> 2)));
                            if (j & 8) v = ev29::tree_combine<5>(s5, v, fr29_mul(zp_upper(5), ev_tab_get(tab_s, m2 >> 3)));
                            else s5 = v;
                        } else {
                            s4 = v;
                        }
                    }
                } else {
                    s3 = v;
                }
            } else {
                s2 = v;
            }
        }
        v = ev29::tree_canonical<LOG_PER>(v);
        if (bad) atomicOr(bad_out + poly, 1u);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            Fr29 other;
#pragma unroll
            for (int i = 0; i < 9; i++) other.l[i] = (uint32_t)__shfl_xor((int)v.l[i], 1 << k);
            const bool odd = (lane >> k) & 1;
            Fr29 e, o;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                e.l[i] = odd ? other.l[i] : v.l[i];
                o.l[i] = odd ? v.l[i] : other.l[i];
            }
            v = fr29_canonical<1>(ev29::tree_combine<0>(e, o, fr29_mul(zc, tab[t >> (k + 1)])));
            if (k < 5 || W > 1) zc = fr29_mul(zc, zc);
        }
        if (W > 1) {
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 9; i++) sh[i][wave] = v.l[i];
            }
            __syncthreads();
            if (tid == 0) {
                Fr29 a[W];
#pragma unroll
                for (int w = 0; w < W; w++) {
#pragma unroll
                    for (int i = 0; i < 9; i++) a[w].l[i] = sh[i][w];
                }
#pragma unroll
                for (int nn = W; nn > 1; nn >>= 1) {
#pragma unroll
                    for (int m = 0; m < nn / 2; m++)
                        a[m] = fr29_canonical<1>(ev29::tree_combine<0>(a[2 * m], a[2 * m + 1], fr29_mul(zc, tab[m])));
                    zc = fr29_mul(zc, zc);
                }
                v = a[0];
            }
            __syncthreads();   // sh is free for the next polynomial of this workgroup
        }
        if (t == 0) vst_fr(y_out + poly, ev29::tree_finish_from_integers(v));
    }
}

__device__ __forceinline__ void ev_put(uint32_t (*sh)[EV_THREADS], int col, const Fr29 &v) {
#pragma unroll
    for (int k = 0; k < 9; k++) sh[k][col] = v.l[k];
}
__device__ __forceinline__ Fr29 ev_get(uint32_t (*sh)[EV_THREADS], int col) {
    Fr29 v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = sh[k][col];
    return v;
}

// QUOT: additionally write the quotient polynomial of the KZG opening at z in evaluation form,
//   q_i = (p_i - y)/(w_i - z) = (y - p_i) * 1/(z - w_i)          (eip4844.c:441-456)
// as canonical little-endian scalars ready for the MSM recoding.  The inverses are parked in the
// output buffer until y is known.  hit_out[blob] = index of the domain point equal to z, or -1;
// for such a blob (eip4844.c:458-481) q is not produced here and the caller takes the scalar path.
template <bool QUOT>
__global__ __launch_bounds__(EV_THREADS) void k_eval_barycentric(Fr *y_out, uint32_t *q_raw, int *hit_out,
                                                                 const Fr *poly, const Fr *zs,
                                                                 const uint32_t *roots29_words) {
    __shared__ uint32_t sh[9][EV_THREADS];    // the threads' products, then their inverses; later the sum tree and y
    __shared__ uint32_t sh2[9][EV_THREADS];   // the first wave's prefix products over the waves
    __shared__ uint32_t sh_f[9];              // (z^4096 - 1)/4096
    __shared__ int hit;
    const int tid = threadIdx.x;
    const Fr *p = poly + (size_t)blockIdx.x * N_BLOB;
    uint32_t *park = q_raw + (size_t)blockIdx.x * N_BLOB * 8;
    const Fr29 *roots29 = reinterpret_cast<const Fr29 *>(roots29_words);
    const Fr29 z = fr29_from_fr(vld_fr(zs + blockIdx.x));
    if (tid == 0) hit = -1;
    __syncthreads();
    // Montgomery's trick over the thread's EV_PER denominators.  Only the prefix products are kept: a denominator
    // is one subtraction from a root the way back loads anyway.
    Fr29 pre[EV_PER], acc;
    const int h = ev29::forward(pre, acc, z, roots29, tid, EV_THREADS);
    if (h >= 0) hit = h;  // at most one domain point equals z
    ev_put(sh, tid, acc);
    __syncthreads();
    if (hit >= 0) {
        if (tid == 0) {
            vst_fr(y_out + blockIdx.x, vld_fr(p + hit));
            if (hit_out) hit_out[blockIdx.x] = hit;
        }
        return;
    }
    if (tid == 0 && hit_out) hit_out[blockIdx.x] = -1;
    if (tid < 64) {
        // the first wave: one inversion per lane for the eight waves' products
        ev29::invert_across([&](int w) { return ev_get(sh, tid + 64 * w); },
                            [&](int w, const Fr29 &v) { ev_put(sh2, tid + 64 * w, v); },
                            [&](int w) { return ev_get(sh2, tid + 64 * w); },
                            [&](int w, const Fr29 &v) { ev_put(sh, tid + 64 * w, v); },
                            [](const Fr29 &v) { return fr29_inv_dev(v); });
    } else if (tid == 64) {
        // meanwhile one lane of the second wave: the factor in front of the sum
        const Fr29 f = ev29::vanishing_over_n(z);
#pragma unroll
        for (int k = 0; k < 9; k++) sh_f[k] = f.l[k];
    }
    __syncthreads();
    Fr sum = ev29::to_fr_radix256(
        ev29::backward(pre, ev_get(sh, tid), z, roots29, p, tid, EV_THREADS, QUOT ? park : nullptr));
    __syncthreads();   // every thread has read its inverse: the rows are free for the sum
    // workgroup sum
    for (int s = EV_THREADS / 2; s >= 1; s >>= 1) {
        if (tid >= s && tid < 2 * s) {
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k][tid - s] = sum.l[k];
        }
        __syncthreads();
        if (tid < s) {
            Fr o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.l[k] = sh[k][tid];
            sum = add(sum, o);
        }
        __syncthreads();
    }
    if (tid == 0) {
        Fr29 f;
#pragma unroll
        for (int k = 0; k < 9; k++) f.l[k] = sh_f[k];
        const Fr y = ev29::scale(sum, f);
        vst_fr(y_out + blockIdx.x, y);
        if (QUOT) {
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k][0] = y.l[k];
        }
    }
    if (QUOT) {
        __syncthreads();
        Fr y;
#pragma unroll
        for (int k = 0; k < 8; k++) y.l[k] = sh[k][0];
        const Fr29 two5 = fr29_const(FR29_2POW5);
#pragma unroll
        for (int k = 0; k < EV_PER; k++) {
            int i = tid + k * EV_THREADS;
            // (y - p_i) 2^256 times 2^261/(z - w_i) is q_i 2^256; times 2^5 (an integer) is q_i itself
            const Fr d = sub(y, vld_fr(p + i));
            uint4 *slot = reinterpret_cast<uint4 *>(park + (size_t)i * 8);
            const uint4 lo = slot[0], hi = slot[1];
            const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const Fr29 q = fr29_canonical<0>(fr29_mul(fr29_mul(fr29_pack(d.l), fr29_pack(dw)), two5));
            uint32_t raw[8];
            fr29_unpack(raw, q);
            slot[0] = make_uint4(raw[0], raw[1], raw[2], raw[3]);
            slot[1] = make_uint4(raw[4], raw[5], raw[6], raw[7]);
        }
    }
}

int eval_blob_bytes_batch_device(DeviceCtx *ctx, Fr *d_y, uint32_t *d_bad, const uint8_t *d_blob_bytes, const Fr *d_z, size_t n) {
    if (!n) return 0;
    const uint32_t *tab = ctx->d_brp_roots29 + (size_t)N_BLOB * 9;
    if (n > EVAL_TWO_PER_SIMD_FROM) {
        // eight polynomials per workgroup turn, one workgroup per compute unit; a workgroup that takes several turns
        // fills its table once
        const size_t turns = (n + 7) / 8;
        const unsigned grid = (unsigned)(turns < 256 ? turns : 256);
        hipLaunchKernelGGL((k_eval_tree<6, 512>), dim3(grid), dim3(512), 0, ctx->stream, d_y, d_bad, d_blob_bytes, d_z, tab, (unsigned)n);
    } else if (n >= EVAL_ONE_WAVE_FROM) {
        // up to 1024 polynomials: a wave each with a SIMD to itself (four per compute unit)
        hipLaunchKernelGGL((k_eval_tree<6, 256>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_y, d_bad, d_blob_bytes, d_z,
                           tab, (unsigned)n);
    } else {
        hipLaunchKernelGGL((k_eval_tree<4, 256>), dim3((unsigned)n), dim3(256), 0, ctx->stream, d_y, d_bad, d_blob_bytes, d_z, tab,
                           (unsigned)n);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The scalars of the three sums of a blob batch (eip4844.c:697-758: sum r^i proof_i, sum r^i z_i proof_i,
// sum r^i C_i) as digit-ready vectors over the 2n points of a call-time table (commitments [0, n), proofs [n, 2n);
// the caller zeroed sc): one lane per blob raises the batch challenge to its own index -- <= 2 log2(n) products --
// so that nothing but r itself (32 bytes, a kernel argument) crosses PCIe between the transcript and the sums.
// r^(2^k), k < 24, made on the host (23 squarings) and passed by value: a lane's power of r is then the product of
// the entries its index selects -- <= 13 products for n = 8192, ~6 on average, and no squarings of its own.
struct RPow2 {
    Fr p[24];
};
__device__ __forceinline__ Fr rpow_at(const RPow2 &t, uint32_t i) {
    Fr pw = Fr::one();
    bool first = true;
#pragma unroll 1
    for (int k = 0; k < 24 && (i >> k); k++) {
        if ((i >> k) & 1u) {
            pw = first ? t.p[k] : mul(pw, t.p[k]);
            first = false;
        }
    }
    return pw;
}
static RPow2 rpow2_of(const Fr &r) {
    RPow2 t;
    t.p[0] = r;
    for (int k = 1; k < 24; k++) t.p[k] = mul(t.p[k - 1], t.p[k - 1]);
    return t;
}

__global__ void k_rlc_scalars(uint32_t *sc, const Fr *z, RPow2 rp2, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr pw = rpow_at(rp2, i);
    uint32_t a[8], c[8];
    to_raw<FrParams>(a, pw);
    to_raw<FrParams>(c, mul(pw, z[i]));
    uint32_t *v0 = sc + ((size_t)0 * 2 * n + n + i) * 8;   // r^i        on proof_i
    uint32_t *v1 = sc + ((size_t)1 * 2 * n + n + i) * 8;   // r^i z_i    on proof_i
    uint32_t *v2 = sc + ((size_t)2 * 2 * n + i) * 8;       // r^i        on C_i
#pragma unroll
    for (int k = 0; k < 8; k++) {
        v0[k] = a[k];
        v1[k] = c[k];
        v2[k] = a[k];
    }
}

int rlc_scalars_enqueue(hipStream_t stream, uint32_t *d_sc, const Fr *d_z, const Fr &r, size_t n) {
    if (!n) return 0;
    HIP_TRY(hipMemsetAsync(d_sc, 0, 6 * n * 8 * sizeof(uint32_t), stream));
    if (n >= ((size_t)1 << 24)) return 2;
    hipLaunchKernelGGL(k_rlc_scalars, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_sc, d_z, rpow2_of(r), (uint32_t)n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// y_i = p_i(z_i) and the quotient scalars q (canonical limbs, [n][4096][8]); d_hit[i] >= 0 flags a
// blob whose z lies in the evaluation domain
int eval_quotient_batch_device(DeviceCtx *ctx, Fr *d_y, uint32_t *d_q_raw, int *d_hit, const Fr *d_poly,
                               const Fr *d_z, size_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_eval_barycentric<true>, dim3((unsigned)n), dim3(EV_THREADS), 0, ctx->stream, d_y,
                       d_q_raw, d_hit, d_poly, d_z, ctx->d_brp_roots29);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// G1 validation: one thread per 48-byte compressed point
// status: 0 ok, 1 invalid (bad encoding / not on curve / not in the r-torsion subgroup)
// ------------------------------------------------------------------------------------------

// [k]P for an affine P through the 28-bit-limb windowed ladder (g1_28.hpp); result in the
// fully reduced 2^384-domain XYZZ form
__device__ __noinline__ G1XYZZ affine_mul_w4(const G1Affine &a, const uint32_t *k) {
    XYZZ28 p, o;
    bool oi;
    p.x = widen<1, 10>(f28_from_fp(a.x));
    p.y = widen<1, 6>(f28_from_fp(a.y));
    p.zz = widen<1, 2>(f28_one());
    p.zzz = p.zz;
    xyzz28_mul_w4(o, oi, p, a.is_inf(), k);
    return xyzz28_to_xyzz(o, oi);
}

// Same decisions as g1_uncompress (g1.hpp) + a subgroup check, but on the 28-bit-limb arithmetic:
// the square root is a sliding-window power (381 squarings + ~80 products) and the subgroup test
// the endomorphism identity [x^2]P = (beta^2 X, -Y) (g1_28.hpp: 126 doublings + 10 additions)
// instead of a 255-bit ladder by r -- about 3x fewer instructions per point.
// MODE 0: decompress and subgroup-check; MODE 1: decompress only (curve membership), the caller runs
// k_subgroup_g1 on the result -- on another stream, next to the work that consumes the points.
template <int MODE>
__global__ void k_validate_g1(G1Affine *out, uint8_t *status, const uint8_t *in48, size_t n) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    // lanes past the end repeat the last point instead of exiting: a partly masked wave runs the 381-squaring
    // square root measurably slower than a full one (fk20.hip, k_g1_fft_twiddle_quad)
    const bool live = g < n;
    if (!live) g = n - 1;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(in48 + g * 48);
    uint32_t raw[12];
#pragma unroll
    for (int i = 0; i < 12; i++) raw[i] = __builtin_bswap32(src[11 - i]);  // big-endian bytes -> LE limbs
    const uint32_t b0 = raw[11] >> 24;
    G1Affine a = G1Affine::inf();
    uint8_t st = 0;
    if (!(b0 & 0x80)) {
        st = 1;  // uncompressed form is not accepted
    } else if (b0 & 0x40) {
        uint32_t rest = raw[11] & 0x3fffffffu;
#pragma unroll
        for (int i = 0; i < 11; i++) rest |= raw[i];
        if (rest) st = 1;  // infinity must be exactly 0xc0 00 .. 00
    } else {
        raw[11] &= 0x1fffffffu;
        uint32_t m[12];
        mod_limbs<FpParams>(m);
        if (limbs_geq<12>(raw, m)) {
            st = 1;
        } else {
            Fp x = from_raw<FpParams>(raw);
            F28<1, 2> x28 = f28_from_fp(x), y28;
            if (!g1_28_solve_y(y28, x28)) {
                st = 1;  // not on the curve
            } else {
                Fp y = f28_to_fp(y28);
                if (fp_is_lex_largest(y) != ((b0 & 0x20) != 0)) {
                    y = neg(y);
                    y28 = f28_from_fp(y);
                }
                if (MODE == 0 && !g1_28_in_subgroup(x28, y28)) {
                    st = 1;
                } else {
                    a = {x, y};
                }
            }
        }
    }
    if (live) {
        out[g] = a;
        status[g] = st;
    }
}

// status[i] = 1 for a finite point outside the prime-order subgroup (points are left untouched)
__global__ void k_subgroup_g1(uint8_t *status, const G1Affine *pts, size_t n) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= n) return;
    G1Affine a = pts[g];
    uint8_t st = 0;
    if (!a.is_inf() && !g1_28_in_subgroup(f28_from_fp(a.x), f28_from_fp(a.y))) st = 1;
    status[g] = st;
}

// the same with four lanes per point (g1_quad.hpp): the 126 doublings are the critical path of a small batch
__global__ __launch_bounds__(64) void k_subgroup_g1_quad(uint8_t *status, const G1Affine *pts, size_t n) {
    size_t g = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 2;
    const bool live = g < n;
    if (!live) g = n - 1;   // no partly masked waves (see k_validate_g1)
    G1Affine a = pts[g];
    uint8_t st = 0;
    if (!a.is_inf() && !quad::g1_28_in_subgroup_quad(f28_from_fp(a.x), f28_from_fp(a.y), (int)(threadIdx.x & 3))) st = 1;
    if ((threadIdx.x & 3) == 0 && live) status[g] = st;
}

int validate_g1_batch_device(DeviceCtx *ctx, G1Affine *d_out, uint8_t *d_status, const uint8_t *d_in48,
                             size_t n, hipStream_t stream) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_validate_g1<0>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream ? stream : ctx->stream,
                       d_out, d_status, d_in48, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int decompress_g1_batch_device(DeviceCtx *ctx, G1Affine *d_out, uint8_t *d_status, const uint8_t *d_in48, size_t n,
                               hipStream_t stream) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_validate_g1<1>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream ? stream : ctx->stream,
                       d_out, d_status, d_in48, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int subgroup_g1_batch_device(DeviceCtx *ctx, uint8_t *d_status, const G1Affine *d_pts, size_t n, hipStream_t stream) {
    if (!n) return 0;
    // four lanes per point while that is at most ~2 waves per SIMD
    static const size_t quad_max = (size_t)ab_knob("CKZG_HIP_QUAD_MAX", 8192);
    if (n <= 4 * quad_max) {
        hipLaunchKernelGGL(k_subgroup_g1_quad, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, stream ? stream : ctx->stream,
                           d_status, d_pts, n);
    } else {
        hipLaunchKernelGGL(k_subgroup_g1, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream ? stream : ctx->stream,
                           d_status, d_pts, n);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// variable-base linear combination: out = sum_i k_i * P_i  (k_i canonical 8 x u32)
// ------------------------------------------------------------------------------------------

constexpr int LC_THREADS = 64;

// Two lanes per term: the scalar is split (GLV, the points were subgroup-checked) and lane 2i takes
// [k1]P_i, lane 2i+1 takes [k2]phi(P_i) -- 128 doublings + 32 additions each instead of one lane doing
// 128 + 64 -- then the workgroup's 64 products (32 terms) are folded in LDS (28-bit-limb domain).
__global__ __launch_bounds__(LC_THREADS) void k_lincomb_partial(G1XYZZ *partials, const G1Affine *pts,
                                                                const uint32_t *scalars, size_t n) {
    __shared__ uint32_t sh[57][LC_THREADS / 2];
    const size_t g = blockIdx.x * (size_t)LC_THREADS + threadIdx.x;
    const size_t term = g >> 1;
    const bool second = (g & 1) != 0;
    XYZZ28 acc;
    bool inf = true;
    if (term < n) {
        G1Affine a = pts[term];
        if (!a.is_inf()) {
            uint32_t k[8], glv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = scalars[term * 8 + i];
            glv_split(k, glv, glv + 4);
            XYZZ28 p;
            p.x = widen<1, 10>(f28_from_fp(a.x));
            if (second) p.x = widen<1, 10>(mul(p.x, f28_const<1, 1>(FP28_BETA_LAMBDA)));
            p.y = widen<1, 6>(f28_from_fp(a.y));
            p.zz = widen<1, 2>(f28_one());
            p.zzz = p.zz;
            xyzz28_mul_w4_128(acc, inf, p, false, second ? glv + 4 : glv);
        }
    }
    block_reduce_xyzz28<LC_THREADS>(acc, inf, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = xyzz28_to_xyzz(acc, inf);
}

// The latency form of the same kernel: FOUR lanes per GLV half-term (g1_quad.hpp), i.e. 8 terms per 64-lane
// workgroup and one partial per 8 terms.  A ladder is ~2.3x shorter and the call's work spreads over four
// times as many waves; used while that still fits the chip in about two waves per SIMD.
__global__ __launch_bounds__(LC_THREADS) void k_lincomb_partial_quad(G1XYZZ *partials, const G1Affine *pts,
                                                                     const uint32_t *scalars, size_t n) {
    __shared__ uint32_t sh[57][LC_THREADS / 2];
    const size_t g = blockIdx.x * (size_t)LC_THREADS + threadIdx.x;
    const int ql = (int)(threadIdx.x & 3);
    const size_t half_term = g >> 2, term_raw = half_term >> 1;
    const bool second = (half_term & 1) != 0;
    XYZZ28 acc;
    bool inf = true;
    // quads past the end repeat the last term (their product is dropped below): no partly masked waves
    const bool live = term_raw < n;
    const size_t term = live ? term_raw : n - 1;
    bool fin = false;
    if (n != 0) {
        G1Affine a = pts[term];
        // a point at infinity (the padding of a job is made of them) runs the ladder on the generator and its product
        // is dropped: the wave stays fully active
        fin = !a.is_inf();
        if (!fin) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                a.x.l[i] = G1_GEN_X[i];
                a.y.l[i] = G1_GEN_Y[i];
            }
        }
        {
            uint32_t k[8], glv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = fin ? scalars[term * 8 + i] : (i == 7 ? 0x1e3779b9u : 0x9e3779b9u);
            glv_split(k, glv, glv + 4);
            XYZZ28 p;
            p.x = widen<1, 10>(f28_from_fp(a.x));
            if (second) p.x = widen<1, 10>(mul(p.x, f28_const<1, 1>(FP28_BETA_LAMBDA)));
            p.y = widen<1, 6>(f28_from_fp(a.y));
            p.zz = widen<1, 2>(f28_one());
            p.zzz = p.zz;
            quad::xyzz28_mul_w4_128_quad(acc, inf, p, false, second ? glv + 4 : glv, ql);
        }
    }
    if (ql != 0 || !live || !fin) inf = true;  // the four lanes of a quad hold the same product: count it once
    block_reduce_xyzz28<LC_THREADS>(acc, inf, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = xyzz28_to_xyzz(acc, inf);
}

// job j owns partials[part_off[j] .. part_off[j+1]): one 64-lane workgroup per job folds them
// (lane-strided partial sums, then the LDS tree) and normalises the result
__global__ __launch_bounds__(64) void k_lincomb_final(G1Affine *out, const G1XYZZ *partials,
                                                     const uint32_t *part_off, int njobs) {
    __shared__ uint32_t sh[57][32];
    const int j = blockIdx.x;
    XYZZ28 acc;
    bool inf = true;
    for (uint32_t i = part_off[j] + threadIdx.x; i < part_off[j + 1]; i += 64) {
        bool oinf;
        XYZZ28 o = xyzz28_from_xyzz(partials[i], oinf);
        xyzz28_add(acc, inf, o, oinf);
    }
    block_reduce_xyzz28<64>(acc, inf, sh);
    if (threadIdx.x == 0) out[j] = xyzz28_to_affine(acc, inf);
}

// `total` (a multiple of 64) points/scalars laid out job after job; h_part_off has njobs+1 entries, in units of
// one partial = 32 terms (quad == false) or 8 terms (quad == true); d_partials holds total/32 resp. total/8 points
int lincomb_multi_device(DeviceCtx *ctx, G1Affine *d_out, G1XYZZ *d_partials, uint32_t *d_off, const G1Affine *d_pts,
                         const uint32_t *d_scalars, size_t total, const uint32_t *h_part_off, int njobs, bool quad) {
    HIP_TRY(hipMemcpyAsync(d_off, h_part_off, (njobs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (quad) {
        hipLaunchKernelGGL(k_lincomb_partial_quad, dim3((unsigned)(8 * total / LC_THREADS)), dim3(LC_THREADS), 0,
                           ctx->stream, d_partials, d_pts, d_scalars, total);
    } else {
        hipLaunchKernelGGL(k_lincomb_partial, dim3((unsigned)(2 * total / LC_THREADS)), dim3(LC_THREADS), 0, ctx->stream,
                           d_partials, d_pts, d_scalars, total);
    }
    hipLaunchKernelGGL(k_lincomb_final, dim3(njobs), dim3(64), 0, ctx->stream, d_out, d_partials, d_off, njobs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(dev::sync_stream(ctx->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------
// cell helpers of recover_cells_and_kzg_proofs / verify_cell_kzg_proof_batch
// ------------------------------------------------------------------------------------------

// cells[b][j] (2048 B each) -> image[b][cell_indices[j]]; the image is zero elsewhere
__global__ void k_scatter_cells(uint4 *image, const uint4 *cells, const uint32_t *idx, uint32_t num_cells,
                                size_t total_u4) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= total_u4) return;
    constexpr uint32_t U4 = 2048 / 16;
    size_t cell = g / U4;
    uint32_t w = (uint32_t)(g % U4);
    size_t b = cell / num_cells;
    uint32_t j = (uint32_t)(cell % num_cells);
    image[(b * 128 + idx[j]) * U4 + w] = cells[g];
}

// interp[k] = sum_c col[c][k] * (h_c^-1)^k with h_c^-1 = w^(8192 - brp7(c))  (eip7594.c:549-566,
// 713-752).  One workgroup per coefficient k, one thread per column: a product each and a seven-level fold in LDS --
// the one-thread-per-coefficient form was 128 dependent multiply-adds (225 us of every cell verification).
__global__ __launch_bounds__(128) void k_interp_sum(Fr *interp, const Fr *cols, const Fr *roots) {
    __shared__ Fr sh[128];
    const uint32_t k = blockIdx.x, c = threadIdx.x;
    const uint32_t rb = __brev(c) >> 25;
    const uint32_t idx = ((8192u - rb) * k) & 8191u;
    sh[c] = mul(vld_fr(cols + c * 64 + k), vld_fr(roots + idx));
    __syncthreads();
    for (uint32_t s2 = 64; s2 >= 1; s2 >>= 1) {
        if (c < s2) sh[c] = add(sh[c], sh[c + s2]);
        __syncthreads();
    }
    if (c == 0) {
        uint32_t raw[8];
        to_raw<FrParams>(raw, sh[0]);  // canonical limbs: this vector is used as MSM scalars
        for (int i = 0; i < 8; i++) reinterpret_cast<uint32_t *>(interp + k)[i] = raw[i];
    }
}

// agg[c][j] = sum over the cells i of column c of r^i * cell_i[j]  (eip7594.c:661-683).
// order[col_start[c] .. col_start[c+1]) lists the cells of column c.  One workgroup per column, thread (j, p): position
// j over every parts-th cell of the list, then a fold over the parts in LDS (a large batch has 64+ cells per column:
// that many dependent multiply-adds per thread in the one-part form).
__global__ void k_cell_aggregate(Fr *agg, const Fr *cell_fr, const Fr *rp, const uint32_t *col_start,
                                 const uint32_t *order, uint32_t parts) {
    extern __shared__ Fr sh_agg[];   // [parts][64]
    const uint32_t c = blockIdx.x, j = threadIdx.x & 63u, p = threadIdx.x >> 6;
    Fr acc = Fr::zero();
    for (uint32_t t = col_start[c] + p; t < col_start[c + 1]; t += parts) {
        const uint32_t i = order[t];
        acc = add(acc, mul(vld_fr(cell_fr + (size_t)i * 64 + j), vld_fr(rp + i)));
    }
    if (parts > 1) {
        sh_agg[p * 64 + j] = acc;
        __syncthreads();
        for (uint32_t s2 = parts >> 1; s2 >= 1; s2 >>= 1) {
            if (p < s2) sh_agg[p * 64 + j] = add(sh_agg[p * 64 + j], sh_agg[(p + s2) * 64 + j]);
            __syncthreads();
        }
        acc = sh_agg[j];
    }
    if (p == 0) vst_fr(agg + c * 64 + j, acc);
}

int scatter_cells_device(DeviceCtx *ctx, uint8_t *d_image, const uint8_t *d_cells, const uint32_t *d_idx,
                         uint32_t num_cells, size_t num_rows) {
    const size_t u4 = num_rows * num_cells * (2048 / 16);
    if (!u4) return 0;
    hipLaunchKernelGGL(k_scatter_cells, dim3((unsigned)((u4 + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<uint4 *>(d_image), reinterpret_cast<const uint4 *>(d_cells), d_idx, num_cells, u4);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The scalars of a cell batch's sums over its call-time table, made from the batch challenge where their digits are
// needed (eip7594.c:926 sum r^i proof_i, :784-812 sum r^i h_k^64 proof_i): lane i raises r to its index, leaves it in
// Montgomery form for k_cell_aggregate and in canonical limbs -- plain and times the coset factor of its cell's
// column -- in the two scalar vectors.  Only r (a kernel argument) crosses PCIe after the transcript.
__global__ void k_cell_rlc_scalars(Fr *rp, uint32_t *vec_rp, uint32_t *vec_wrp, const uint32_t *cell_idx, const Fr *roots,
                                   RPow2 rp2, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr pw = rpow_at(rp2, i);
    rp[i] = pw;
    const uint32_t rb = __brev(cell_idx[i]) >> 25;   // 7-bit reversal of the column index (128 columns)
    uint32_t a[8], c[8];
    to_raw<FrParams>(a, pw);
    to_raw<FrParams>(c, mul(pw, roots[(size_t)rb * N_CELL]));
#pragma unroll
    for (int k = 0; k < 8; k++) {
        vec_rp[(size_t)i * 8 + k] = a[k];
        vec_wrp[(size_t)i * 8 + k] = c[k];
    }
}

// Weight of each distinct commitment (eip7594.c:494-539): the sum of r^i over the cells that name it.  One 64-lane
// workgroup per commitment over the member list the host grouped while the transcript was being hashed.
__global__ __launch_bounds__(64) void k_commit_weights(uint32_t *vec_w, const Fr *rp, const uint32_t *grp_start,
                                                      const uint32_t *members) {
    __shared__ Fr sh[64];
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    Fr acc = Fr::zero();
    for (uint32_t m = grp_start[j] + tid; m < grp_start[j + 1]; m += 64) acc = add(acc, rp[members[m]]);
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t s2 = 32; s2 >= 1; s2 >>= 1) {
        if (tid < s2) sh[tid] = add(sh[tid], sh[tid + s2]);
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t a[8];
        to_raw<FrParams>(a, sh[0]);
#pragma unroll
        for (int k = 0; k < 8; k++) vec_w[(size_t)j * 8 + k] = a[k];
    }
}

int cell_rlc_scalars_enqueue(DeviceCtx *ctx, Fr *d_rp, uint32_t *d_vec_rp, uint32_t *d_vec_wrp, uint32_t *d_vec_w,
                             const uint32_t *d_cell_idx, const uint32_t *d_grp_start, const uint32_t *d_members,
                             const Fr &r, size_t n, size_t nc) {
    if (!n) return 0;
    if (n >= ((size_t)1 << 24)) return 2;
    hipLaunchKernelGGL(k_cell_rlc_scalars, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_rp, d_vec_rp, d_vec_wrp,
                       d_cell_idx, ctx->d_roots, rpow2_of(r), (uint32_t)n);
    hipLaunchKernelGGL(k_commit_weights, dim3((unsigned)nc), dim3(64), 0, ctx->stream, d_vec_w, d_rp, d_grp_start, d_members);
    HIP_TRY(hipGetLastError());
    return 0;
}

int cell_aggregate_device(DeviceCtx *ctx, Fr *d_agg, const Fr *d_cell_fr, const Fr *d_rp, const uint32_t *d_col_start,
                          const uint32_t *d_order, size_t n_cells) {
    // parts: a power of two, ~4 cells per thread, at most 16 (1024 threads, 32 KB of LDS)
    uint32_t parts = 1;
    while (parts < 16 && (size_t)parts * 128 * 4 < n_cells) parts <<= 1;
    hipLaunchKernelGGL(k_cell_aggregate, dim3(128), dim3(64 * parts), parts > 1 ? parts * 64 * sizeof(Fr) : 0, ctx->stream, d_agg,
                       d_cell_fr, d_rp, d_col_start, d_order, parts);
    HIP_TRY(hipGetLastError());
    return 0;
}

int interp_sum_device(DeviceCtx *ctx, Fr *d_interp, const Fr *d_cols) {
    hipLaunchKernelGGL(k_interp_sum, dim3(64), dim3(128), 0, ctx->stream, d_interp, d_cols, ctx->d_roots);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// element-wise Fr helpers
// ------------------------------------------------------------------------------------------

// a[i] *= b[i mod period]
__global__ void k_fr_mul_inplace(Fr *a, const Fr *b, size_t n, size_t period) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= n) return;
    vst_fr(a + g, mul(vld_fr(a + g), vld_fr(b + (g % period))));
}

// a[i] = a[i] / b[i] with Montgomery's trick inside each thread's run of 16 (recovery.c:322-328
// does 8192 separate inversions; division by zero yields 0 like blst_fr_eucl_inverse)
__global__ void k_fr_div_inplace(Fr *a, const Fr *b, size_t n) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t base = t * 16;
    if (base >= n) return;
    Fr den[16], pre[16];
    Fr acc = Fr::one();
    int cnt = (int)(n - base < 16 ? n - base : 16);
    for (int k = 0; k < cnt; k++) {
        den[k] = vld_fr(b + base + k);
        pre[k] = acc;
        if (!den[k].is_zero()) acc = mul(acc, den[k]);
    }
    Fr inv = fr_inv_dev(acc);
    for (int k = cnt - 1; k >= 0; k--) {
        Fr q = Fr::zero();
        if (!den[k].is_zero()) {
            Fr di = mul(inv, pre[k]);
            inv = mul(inv, den[k]);
            q = mul(vld_fr(a + base + k), di);
        }
        vst_fr(a + base + k, q);
    }
}

int fr_mul_inplace_device(DeviceCtx *ctx, Fr *d_a, const Fr *d_b, size_t n, size_t period) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_fr_mul_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_a,
                       d_b, n, period);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// Fiat-Shamir challenges on the GPU: z_i = SHA-256("FSBLOBVERIFY_V1_" | u64be 0 | u64be 4096 |
// blob_i | commitment_i) mod r  (compute_challenge, src/eip4844/eip4844.c:147-178).
// SHA-256 is sequential per message -- 2050 dependent compressions per blob -- and a wave on its own already issues
// one VALU instruction per four cycles, which is all a SIMD has: the kernel's run time IS the instruction count of
// the wave that carries the chaining state.  Of the 1,405 instructions a compression took in round 3, a third (the
// message schedule W[16..63], the byte swaps, the + K[t]) do not depend on that state.  So a workgroup is TWO waves
// over the same 64 blobs: the producer wave loads block i of its lane's blob, expands the schedule and stores
// W[t] + K[t] to LDS (64 words per lane, double-buffered: 32 KB) while the consumer wave runs the 64 rounds of
// block i - 1 from LDS -- 14 instructions a round, ~920 per block -- one barrier per block.  The two waves sit on
// different SIMDs of the CU.  The 32-byte header shifts the blob by half a block, so block b (1 <= b <= 2047) is
// the 64 contiguous, 32-byte aligned bytes blob[64b-32, 64b+32); the producer requests the next block's four
// 16-byte loads before it expands the current one.
// ------------------------------------------------------------------------------------------

__device__ __constant__ uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_rotateright32(x, n); }
// gfx950's v_bitop3_b32 evaluates any three-input boolean function in one instruction (truth table = the function
// applied to A = 0xF0, B = 0xCC, C = 0xAA): the three-way XORs of the sigma functions, Ch and Maj are one
// instruction each instead of two or three.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

__device__ __forceinline__ void be_words(uint32_t *w, const uint4 &v) {
    w[0] = __builtin_bswap32(v.x);
    w[1] = __builtin_bswap32(v.y);
    w[2] = __builtin_bswap32(v.z);
    w[3] = __builtin_bswap32(v.w);
}

// producer: w[16] holds a block as big-endian words; writes W[t] + K[t], t = 0..63, as 16 uint4 to kw[.][lane]
__device__ __forceinline__ void sha256_expand_to_lds(uint4 (*kw)[64], int lane, uint32_t (&w)[16]) {
#pragma unroll
    for (int t4 = 0; t4 < 16; t4++) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = 4 * t4 + k;
            if (t >= 16) {
                uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
                uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
                w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
            }
            o[k] = w[t & 15] + SHA256_K[t];
        }
        kw[t4][lane] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// consumer: the 64 rounds of one block from the W + K words the producer left in LDS
__device__ __forceinline__ void sha256_rounds_from_lds(uint32_t (&h)[8], const uint4 (*kw)[64], int lane) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int t4 = 0; t4 < 16; t4++) {
        const uint4 q = kw[t4][lane];
        const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
            uint32_t t1 = hh + S1 + sha_ch(e, f, g) + x[k];
            uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
            uint32_t t2 = S0 + sha_maj(a, b, c);
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

constexpr int SHA_BLOCKS = 2050;   // 32-byte header + 131,072-byte blob + 48-byte commitment + padding

__global__ __launch_bounds__(128) void k_sha256_challenges(Fr *z_out, const uint8_t *blobs, const uint8_t *commit48,
                                                           size_t n) {
    __shared__ uint4 kw[2][16][64];
    const int lane = threadIdx.x & 63;
    const bool producer = threadIdx.x >= 64;
    const size_t g_real = blockIdx.x * (size_t)64 + lane;
    const size_t g = g_real < n ? g_real : n - 1;   // lanes past the end repeat the last blob (no partly masked wave)
    const uint4 *bp = reinterpret_cast<const uint4 *>(blobs + g * (size_t)(N_BLOB * 32));
    const uint32_t *cp = reinterpret_cast<const uint32_t *>(commit48 + g * 48);
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint4 n0, n1, n2, n3;   // producer: the next block's bytes, requested one block ahead
    if (producer) {
        n0 = bp[0];
        n1 = bp[1];
    }
    for (int i = 0; i <= SHA_BLOCKS; i++) {
        if (producer) {
            if (i < SHA_BLOCKS) {
                uint32_t w[16];
                if (i == 0) {
                    // block 0: "FSBLOBVERIFY_V1_" | 0^8 | u64be(4096) | blob[0,32)
                    w[0] = 0x4653424c; w[1] = 0x4f425645; w[2] = 0x52494659; w[3] = 0x5f56315f;
                    w[4] = 0; w[5] = 0; w[6] = 0; w[7] = N_BLOB;
                    be_words(w + 8, n0);
                    be_words(w + 12, n1);
                    n0 = bp[2]; n1 = bp[3]; n2 = bp[4]; n3 = bp[5];
                } else if (i < 2048) {
                    // blocks 1..2047: blob[64i-32, 64i+32) = uint4 index 4i-2 .. 4i+1
                    be_words(w, n0);
                    be_words(w + 4, n1);
                    be_words(w + 8, n2);
                    be_words(w + 12, n3);
                    const int nb = 4 * (i + 1) - 2;  // next block; the last of these requests the 32-byte tail
                    n0 = bp[nb];
                    n1 = bp[nb + 1];
                    if (i < 2047) {
                        n2 = bp[nb + 2];
                        n3 = bp[nb + 3];
                    }
                } else if (i == 2048) {
                    // block 2048: blob[131040, 131072) | commitment[0, 32)
                    be_words(w, n0);
                    be_words(w + 4, n1);
#pragma unroll
                    for (int k = 0; k < 8; k++) w[8 + k] = __builtin_bswap32(cp[k]);
                } else {
                    // block 2049: commitment[32, 48) | 0x80 | zeros | bit length (131152 bytes)
#pragma unroll
                    for (int k = 0; k < 4; k++) w[k] = __builtin_bswap32(cp[8 + k]);
                    w[4] = 0x80000000u;
#pragma unroll
                    for (int k = 5; k < 15; k++) w[k] = 0;
                    w[15] = (uint32_t)((32 + N_BLOB * 32 + 48) * 8);
                }
                sha256_expand_to_lds(kw[i & 1], lane, w);
            }
        } else if (i >= 1) {
            sha256_rounds_from_lds(h, kw[(i - 1) & 1], lane);
        }
        // block i is in LDS and block i - 1 has been consumed: the producer may overwrite the latter's buffer
        __syncthreads();
    }
    if (producer || g_real >= n) return;
    // digest as a big-endian 256-bit integer, reduced mod r (hash_to_bls_field, eip4844.c:121-127)
    uint32_t raw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) raw[k] = h[7 - k];
    vst_fr(z_out + g_real, from_raw<FrParams>(raw));
}

int sha256_challenges_device(DeviceCtx *ctx, Fr *d_z, const uint8_t *d_blobs, const uint8_t *d_commit48, size_t n,
                             hipStream_t stream) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_sha256_challenges, dim3((unsigned)((n + 63) / 64)), dim3(128), 0, stream ? stream : ctx->stream, d_z,
                       d_blobs, d_commit48, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// The batch transcript's rows, assembled where their parts already are (resident verification): row i =
// commitment_i | z_i | y_i | proof_i as the 160 bytes compute_r_powers_for_verify_kzg_proof_batch hashes per blob
// (src/eip4844/eip4844.c:597-680) -- the host then runs ONE SHA-256 over one contiguous buffer that arrived in one
// copy, instead of converting 2 n field elements and stitching four arrays.  pts48: commitments [0, n), proofs [n, 2n).
// ------------------------------------------------------------------------------------------
__global__ void k_batch_transcript_rows(uint4 *out, const uint4 *pts48, const Fr *z, const Fr *y, size_t n) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint4 *row = out + g * 10;
    const uint4 *c = pts48 + g * 3, *p = pts48 + (n + g) * 3;
    row[0] = c[0]; row[1] = c[1]; row[2] = c[2];
    uint32_t raw[8];
    to_raw<FrParams>(raw, vld_fr(z + g));
    row[3] = make_uint4(__builtin_bswap32(raw[7]), __builtin_bswap32(raw[6]), __builtin_bswap32(raw[5]), __builtin_bswap32(raw[4]));
    row[4] = make_uint4(__builtin_bswap32(raw[3]), __builtin_bswap32(raw[2]), __builtin_bswap32(raw[1]), __builtin_bswap32(raw[0]));
    to_raw<FrParams>(raw, vld_fr(y + g));
    row[5] = make_uint4(__builtin_bswap32(raw[7]), __builtin_bswap32(raw[6]), __builtin_bswap32(raw[5]), __builtin_bswap32(raw[4]));
    row[6] = make_uint4(__builtin_bswap32(raw[3]), __builtin_bswap32(raw[2]), __builtin_bswap32(raw[1]), __builtin_bswap32(raw[0]));
    row[7] = p[0]; row[8] = p[1]; row[9] = p[2];
}

int batch_transcript_rows_device(DeviceCtx *ctx, uint8_t *d_rows, const uint8_t *d_pts48, const Fr *d_z, const Fr *d_y, size_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_batch_transcript_rows, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream,
                       reinterpret_cast<uint4 *>(d_rows), reinterpret_cast<const uint4 *>(d_pts48), d_z, d_y, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int fr_div_inplace_device(DeviceCtx *ctx, Fr *d_a, const Fr *d_b, size_t n) {
    if (!n) return 0;
    size_t threads = (n + 15) / 16;
    hipLaunchKernelGGL(k_fr_div_inplace, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, ctx->stream,
                       d_a, d_b, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace dev
}  // namespace ckzg
