// pippenger.hip -- variable-base multi-scalar multiplication over BLS12-381 G1 by bucket accumulation.
//
// IN THE PRODUCT SINCE ROUND 6, ON REQUEST ONLY: ckzg_hip_g1_lincomb(algo = 2) runs these kernels (the bucket method is
// what the reference's g1_lincomb_fast is, src/common/lincomb.c:65-123 -> blst_p1s_mult_pippenger, and what north_star
// names); the library's OWN variable-base sums keep the per-term GLV ladders of verify.hip.  Why: at every size the
// c-kzg API can produce (n <= ~10^4 terms per sum, three or four sums per call) both algorithms end in the same ~128
// sequential doublings (~1.4 ms of one lane), and the ladders won the same-box A/B at every n from 128 to 65,536
// (profiles/r02_lincomb_ab.txt: n = 8192 13.0 vs 15.0 ms, n = 65,536 86.6 vs 99.6 ms for the call).  A regime where
// buckets win needs n >> 10^5 terms over shared points AND a sort-based scatter instead of the gather-by-comparison
// below (whose sweeps cost n x buckets comparisons); no caller of this library has such sums.  (Rounds 2-5 kept the
// kernels in a second library, libckzg_hip_buckets.so, and the product answered C_KZG_BADARGS to algo = 2.)
//
// Replaces g1_lincomb_fast -> blst_p1s_mult_pippenger (src/common/lincomb.c:65-123) for sums whose bases
// are NOT fixed by the trusted setup: the proofs / commitments of verify_cell_kzg_proof_batch
// (src/eip7594/eip7594.c:530,807,926) and of verify_blob_kzg_proof_batch (src/eip4844/eip4844.c:731-746).
// Fixed bases never come here (msm.hip: tables).  Small sums keep the per-term GLV ladders of
// verify.hip (k_lincomb_partial): see the hand-over note at gpu_lincomb_multi() in ckzg_api2.hip.
//
// A CPU Pippenger scatters points into buckets with data-dependent stores.  Here nothing is scattered:
//   1. k_pip_prepare  one lane per term: balanced GLV split (the points were subgroup-checked), signed c-bit
//                     digits of both 128-bit halves -> digits[job][hw][i] (int16), and the point re-encoded once
//                     in the 28-bit-limb / 2^392 domain the adders work in.
//   2. k_pip_buckets  eight lanes per (job, half-window hw, bucket b): they sweep the hw-th digit row and record
//                     the terms whose digit is +-(b+1) ("gather by comparison": no sort, no atomics), add their
//                     recorded points in lockstep into private XYZZ accumulators, and an LDS tree folds the
//                     eight lane sums.  Duplicates, P / -P pairs and points at infinity need no special cases
//                     beyond the complete addition law.
//   3. k_pip_bits     sum_b (b+1) B_b without the sequential running sum: for every bit t of the bucket weight,
//                     T_t = sum of the buckets whose weight has bit t (one wave per (job, hw, t): lane-strided
//                     sums + LDS tree), so that W_hw = sum_t 2^t T_t.
//   4. k_pip_combine  one workgroup per job: lane (h, t) runs Horner over the windows of its half,
//                     S_{h,t} = sum_w 2^(c w) T_{h,w,t} (c doublings + one addition per window); then
//                     H_h = sum_t 2^t S_{h,t}, result = H_1 + phi(H_2), normalised to affine.
// Sequential depth: ~8 + ~10 additions, then 128 doublings + ~25 additions: the doublings of the last step
// are inherent to any 128-bit scalar and set the latency floor (~1.3 ms of one lane); the total work is
// 2n * ceil(128/c) mixed additions instead of 2n * (128 doublings + 46 additions).
#include "device.hpp"
#include "dev_inline.hpp"
#include "g1_28.hpp"


namespace ckzg {
namespace dev {

bool bucket_msm_available() { return true; }

namespace {

// XYZZ28 + infinity flag as stored between the kernels of this file (limbs as they are: no domain change)
struct P28 {
    uint32_t w[57];
};

__device__ __forceinline__ void st28(P28 *dst, const XYZZ28 &a, bool inf) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&a);
#pragma unroll
    for (int k = 0; k < 56; k++) dst->w[k] = src[k];
    dst->w[56] = inf ? 1u : 0u;
}

__device__ __forceinline__ XYZZ28 ld28(const P28 *src, bool &inf) {
    XYZZ28 a;
    uint32_t *dst = reinterpret_cast<uint32_t *>(&a);
#pragma unroll
    for (int k = 0; k < 56; k++) dst[k] = src->w[k];
    inf = src->w[56] != 0;
    return a;
}

}  // namespace

// digits[(job_hw_row)][i]: row = hw, i = term index inside the whole padded term array (jobs back to back)
__global__ void k_pip_prepare(int16_t *digits, G1Affine *pts392, const G1Affine *pts, const uint32_t *scalars,
                              size_t total, int wbits, int twin) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    G1Affine a = pts[i];
    uint32_t k[8];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = scalars[i * 8 + j];
    if (a.is_inf()) {
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = 0;  // contributes nothing; its digits are all zero
    }
    glv_digits(digits + i, total, k, wbits, twin);  // rows 0..twin-1: k2 (phi half), twin..2twin-1: k1
    Fp k8;
#pragma unroll
    for (int j = 0; j < 12; j++) k8.l[j] = FP_MONT_2POW8[j];
    pts392[i] = {mul(a.x, k8), mul(a.y, k8)};  // fully reduced, 2^392 domain, packed like a table entry
}

// grid: (buckets / (64 / LPB), nhw, njobs); 64 lanes = 64/LPB buckets of LPB lanes each.
// Two phases, because a mixed addition executed whenever ANY lane of the wave has a match would serialise
// the wave's matches (measured: 10.5 ms at n = 8192 for the one-phase form): first every lane sweeps its
// stride of the digit row and only *records* its matches (index + sign, up to CAP per lane, in LDS); then the
// lanes add their j-th recorded point in lockstep, j = 0, 1, ... -- about n / 2^(c-2) / LPB rounds -- and an
// LDS tree of log2(LPB) levels folds each bucket.  Matches beyond CAP (rare) are added on the spot.
template <int LPB>
__global__ __launch_bounds__(64) void k_pip_buckets(P28 *buckets, const int16_t *digits, const G1Affine *pts392,
                                                   const uint32_t *job_off, size_t total, uint32_t nbuckets) {
    __shared__ uint32_t sh[57][32];
    constexpr int GROUPS = 64 / LPB, CAP = 6;
    __shared__ uint32_t lst[CAP][64];
    const int tid = threadIdx.x, grp = tid / LPB, l = tid % LPB;
    const uint32_t b = blockIdx.x * GROUPS + grp, hw = blockIdx.y, job = blockIdx.z;
    const uint32_t i0 = job_off[job], i1 = job_off[job + 1];
    const int16_t *row = digits + (size_t)hw * total;
    const int want = (int)b + 1;
    XYZZ28 acc;
    bool inf = true;
    auto add_point = [&](uint32_t rec) {
        const uint32_t i = rec & 0x7fffffffu;
        const uint4 *src = reinterpret_cast<const uint4 *>(pts392 + i);
        uint32_t wd[24];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            uint4 v = src[k];
            wd[4 * k] = v.x; wd[4 * k + 1] = v.y; wd[4 * k + 2] = v.z; wd[4 * k + 3] = v.w;
        }
        xyzz28_madd(acc, inf, f28_unpack<1>(wd), cneg_reduced(f28_unpack<1>(wd + 12), (rec >> 31) != 0));
    };
    int cnt = 0;
    if (b < nbuckets) {
        for (uint32_t i = i0 + l; i < i1; i += LPB) {
            const int d = row[i];
            const int mag = d < 0 ? -d : d;
            if (mag == want) {
                const uint32_t rec = i | (d < 0 ? 0x80000000u : 0u);
                if (cnt < CAP) {
                    lst[cnt][tid] = rec;
                    cnt++;
                } else {
                    add_point(rec);
                }
            }
        }
    }
    for (int j = 0; j < CAP; j++) {
        if (__ballot(j < cnt) == 0) break;  // wave-uniform
        if (j < cnt) add_point(lst[j][tid]);
    }
    for (int s = LPB / 2; s >= 1; s >>= 1) {
        if (l >= s && l < 2 * s) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
            const int slot = grp * (LPB / 2) + (l - s);
#pragma unroll
            for (int k = 0; k < 56; k++) sh[k][slot] = src[k];
            sh[56][slot] = inf ? 1u : 0u;
        }
        __syncthreads();
        if (l < s) {
            XYZZ28 o;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&o);
            const int slot = grp * (LPB / 2) + l;
#pragma unroll
            for (int k = 0; k < 56; k++) dst[k] = sh[k][slot];
            xyzz28_add(acc, inf, o, sh[56][slot] != 0);
        }
        __syncthreads();
    }
    if (l == 0 && b < nbuckets) st28(buckets + ((size_t)job * gridDim.y + hw) * nbuckets + b, acc, inf);
}

// grid: (wbits, nhw, njobs); T[job][hw][t] = sum of buckets b with bit t of (b+1) set
__global__ __launch_bounds__(64) void k_pip_bits(P28 *tsum, const P28 *buckets, uint32_t nbuckets) {
    __shared__ uint32_t sh[57][32];
    const uint32_t t = blockIdx.x, hw = blockIdx.y, job = blockIdx.z;
    const P28 *bk = buckets + ((size_t)job * gridDim.y + hw) * nbuckets;
    XYZZ28 acc;
    bool inf = true;
    for (uint32_t b = threadIdx.x; b < nbuckets; b += 64) {
        if (((b + 1) >> t) & 1u) {
            bool oi;
            XYZZ28 o = ld28(bk + b, oi);
            xyzz28_add(acc, inf, o, oi);
        }
    }
    block_reduce_xyzz28<64>(acc, inf, sh);
    if (threadIdx.x == 0) st28(tsum + ((size_t)job * gridDim.y + hw) * gridDim.x + t, acc, inf);
}

// grid: njobs; 64 lanes: lane = h * wbits + t for t < wbits, h < 2 (wbits <= 16)
__global__ __launch_bounds__(64) void k_pip_combine(G1Affine *out, const P28 *tsum, int wbits, int twin) {
    __shared__ uint32_t sh[57][32];
    const uint32_t job = blockIdx.x;
    const int lane = threadIdx.x, h = lane / wbits, t = lane - h * wbits;
    XYZZ28 acc;
    bool inf = true;
    if (h < 2) {
        // rows: h = 0 -> k1 half = rows twin..2twin-1, h = 1 -> k2 (phi) half = rows 0..twin-1
        const int row0 = h == 0 ? twin : 0;
        const P28 *base = tsum + ((size_t)job * 2 * twin) * wbits;
        for (int w = twin - 1; w >= 0; w--) {
            if (!inf) {
                for (int k = 0; k < wbits; k++) xyzz28_dbl(acc);
            }
            bool oi;
            XYZZ28 o = ld28(base + (size_t)(row0 + w) * wbits + t, oi);
            xyzz28_add(acc, inf, o, oi);
        }
    }
    // H_h = sum_t 2^t S_{h,t}: Horner across lanes through LDS, led by lanes t == 0
    // (slot s of the exchange buffer holds lane s's value)
    for (int step = wbits - 1; step >= 1; step--) {
        // lanes with t == step publish; lanes with t == step - 1 take 2 * incoming + own
        __syncthreads();
        if (h < 2 && t == step && lane < 64) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
            const int slot = h;  // one value per half at a time
#pragma unroll
            for (int k = 0; k < 56; k++) sh[k][slot] = src[k];
            sh[56][slot] = inf ? 1u : 0u;
        }
        __syncthreads();
        if (h < 2 && t == step - 1) {
            XYZZ28 o;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
            for (int k = 0; k < 56; k++) dst[k] = sh[k][h];
            bool oi = sh[56][h] != 0;
            if (!oi) xyzz28_dbl(o);
            // acc = own + 2 * incoming
            xyzz28_add(acc, inf, o, oi);
        }
    }
    // lane 0 holds H_1 (k1 half), lane wbits holds H_2 (k2 half): result = H_1 + phi(H_2)
    __syncthreads();
    if (lane == wbits) {
        if (!inf) acc.x = widen<1, 10>(mul(acc.x, f28_const<1, 1>(FP28_BETA_LAMBDA)));
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
#pragma unroll
        for (int k = 0; k < 56; k++) sh[k][0] = src[k];
        sh[56][0] = inf ? 1u : 0u;
    }
    __syncthreads();
    if (lane == 0) {
        XYZZ28 o;
        uint32_t *dst = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
        for (int k = 0; k < 56; k++) dst[k] = sh[k][0];
        xyzz28_add(acc, inf, o, sh[56][0] != 0);
        out[job] = xyzz28_to_affine(acc, inf);
    }
}

size_t bucket_msm_scratch_bytes(size_t total, int njobs, int wbits) {
    const size_t twin = FixedBaseTable::twin_for(wbits), nhw = 2 * twin, nb = (size_t)1 << (wbits - 1);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    return al(total * nhw * sizeof(int16_t)) + al(total * sizeof(G1Affine)) + al((size_t)njobs * nhw * nb * sizeof(P28)) +
           al((size_t)njobs * nhw * wbits * sizeof(P28)) + al((size_t)(njobs + 1) * 4);
}

// Window width for the largest job: about log2(n) - 3 balances the 2n*nwin bucket additions against the
// buckets' own reduction; clamped to [5, 12] (measured on MI355X: tools/bench_lincomb.py).
int bucket_msm_wbits(size_t max_job_terms) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= max_job_terms) lg++;
    int c = lg - 3;
    return c < 5 ? 5 : (c > 12 ? 12 : c);
}

// out[j] = sum over the terms [h_job_off[j], h_job_off[j+1]) of scalars[i] * pts[i], for njobs jobs laid out
// back to back in pts / scalars (`total` entries; canonical 8 x u32 scalars; points subgroup-checked affine or
// (0,0) = infinity).  `scratch` holds bucket_msm_scratch_bytes(total, njobs, wbits).  Enqueue-only.
int bucket_msm_enqueue(DeviceCtx *ctx, G1Affine *d_out, const G1Affine *d_pts, const uint32_t *d_scalars, size_t total,
                       const uint32_t *h_job_off, int njobs, int wbits, uint8_t *scratch) {
    if (njobs <= 0 || total == 0) return 0;
    if (wbits < 4 || wbits > 16) return 2;
    const size_t twin = FixedBaseTable::twin_for(wbits), nhw = 2 * twin, nb = (size_t)1 << (wbits - 1);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t off = 0;
    int16_t *d_digits = reinterpret_cast<int16_t *>(scratch + off);
    off += al(total * nhw * sizeof(int16_t));
    G1Affine *d_pts392 = reinterpret_cast<G1Affine *>(scratch + off);
    off += al(total * sizeof(G1Affine));
    P28 *d_buckets = reinterpret_cast<P28 *>(scratch + off);
    off += al((size_t)njobs * nhw * nb * sizeof(P28));
    P28 *d_tsum = reinterpret_cast<P28 *>(scratch + off);
    off += al((size_t)njobs * nhw * wbits * sizeof(P28));
    uint32_t *d_job_off = reinterpret_cast<uint32_t *>(scratch + off);
    HIP_TRY(hipMemcpyAsync(d_job_off, h_job_off, (size_t)(njobs + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pip_prepare, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, ctx->stream, d_digits, d_pts392,
                       d_pts, d_scalars, total, wbits, (int)twin);
    // lanes per bucket: a whole wave (coalesced sweep of the digit row, 6-level fold).  Measured at n = 8192,
    // c = 10 (tools/bench_lincomb.sh): 64 lanes one-phase 3.2 ms, 8 lanes one-phase 10.5 ms (matches serialise),
    // 8 lanes two-phase 5.7 ms (each lane's sweep is 8x longer and latency-bound), 64 lanes two-phase: see DESIGN.md
    constexpr int LPB = 64;
    hipLaunchKernelGGL(k_pip_buckets<LPB>, dim3((unsigned)((nb + 64 / LPB - 1) / (64 / LPB)), (unsigned)nhw, (unsigned)njobs),
                       dim3(64), 0, ctx->stream, d_buckets, d_digits, d_pts392, d_job_off, total, (uint32_t)nb);
    hipLaunchKernelGGL(k_pip_bits, dim3((unsigned)wbits, (unsigned)nhw, (unsigned)njobs), dim3(64), 0, ctx->stream, d_tsum,
                       d_buckets, (uint32_t)nb);
    hipLaunchKernelGGL(k_pip_combine, dim3((unsigned)njobs), dim3(64), 0, ctx->stream, d_out, d_tsum, wbits, (int)twin);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace dev
}  // namespace ckzg

