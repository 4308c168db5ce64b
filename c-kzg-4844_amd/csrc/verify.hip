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

__device__ __noinline__ Fr fr_inv_dev(const Fr &a) { return fr_inv(a); }

// ------------------------------------------------------------------------------------------
// barycentric evaluation: one 256-thread workgroup per polynomial
//   y = (z^4096 - 1)/4096 * sum_i p_i w_i / (z - w_i),   or p_m if z == w_m
// ------------------------------------------------------------------------------------------

constexpr int EV_THREADS = 256;
constexpr int EV_PER = N_BLOB / EV_THREADS;  // 16 terms per thread

// QUOT: additionally write the quotient polynomial of the KZG opening at z in evaluation form,
//   q_i = (p_i - y)/(w_i - z) = (y - p_i) * 1/(z - w_i)          (eip4844.c:441-456)
// as canonical little-endian scalars ready for the MSM recoding.  The inverses are parked in the
// output buffer until y is known.  hit_out[blob] = index of the domain point equal to z, or -1;
// for such a blob (eip4844.c:458-481) q is not produced here and the caller takes the scalar path.
template <bool QUOT>
__global__ __launch_bounds__(EV_THREADS) void k_eval_barycentric(Fr *y_out, uint32_t *q_raw, int *hit_out,
                                                                 const Fr *poly, const Fr *zs,
                                                                 const Fr *brp_roots) {
    __shared__ uint32_t sh[8][EV_THREADS];
    __shared__ int hit;
    const int tid = threadIdx.x;
    const Fr *p = poly + (size_t)blockIdx.x * N_BLOB;
    Fr *qinv = reinterpret_cast<Fr *>(q_raw) + (size_t)blockIdx.x * N_BLOB;
    const Fr z = vld_fr(zs + blockIdx.x);
    if (tid == 0) hit = -1;
    __syncthreads();
    Fr den[EV_PER], pre[EV_PER];
    Fr acc = Fr::one();
#pragma unroll
    for (int k = 0; k < EV_PER; k++) {
        int i = tid + k * EV_THREADS;
        den[k] = sub(z, vld_fr(brp_roots + i));
        if (den[k].is_zero()) hit = i;  // at most one domain point equals z
        pre[k] = acc;
        acc = mul(acc, den[k]);
    }
    __syncthreads();
    if (hit >= 0) {
        if (tid == 0) {
            vst_fr(y_out + blockIdx.x, vld_fr(p + hit));
            if (hit_out) hit_out[blockIdx.x] = hit;
        }
        return;
    }
    if (tid == 0 && hit_out) hit_out[blockIdx.x] = -1;
    Fr inv = fr_inv_dev(acc);
    Fr sum = Fr::zero();
#pragma unroll
    for (int k = EV_PER - 1; k >= 0; k--) {
        int i = tid + k * EV_THREADS;
        Fr di = mul(inv, pre[k]);  // 1/(z - w_i)
        inv = mul(inv, den[k]);
        if (QUOT) vst_fr(qinv + i, di);
        sum = add(sum, mul(mul(di, vld_fr(brp_roots + i)), vld_fr(p + i)));
    }
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
        Fr zn = z;
        for (int k = 0; k < 12; k++) zn = sqr(zn);  // z^4096
        Fr f = sub(zn, Fr::one());
        Fr n_inv = Fr::one();                        // 1/4096 by halving
        uint32_t m[8];
        mod_limbs<FrParams>(m);
        for (int k = 0; k < 12; k++) {
            uint32_t t[9];
            uint32_t c = 0;
            if (n_inv.l[0] & 1u) {
                c = limbs_add<8>(t, n_inv.l, m);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) t[i] = n_inv.l[i];
            }
            t[8] = c;
#pragma unroll
            for (int i = 0; i < 8; i++) n_inv.l[i] = (t[i] >> 1) | (t[i + 1] << 31);
        }
        Fr y = mul(mul(sum, n_inv), f);
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
#pragma unroll
        for (int k = 0; k < EV_PER; k++) {
            int i = tid + k * EV_THREADS;
            Fr q = mul(sub(y, vld_fr(p + i)), vld_fr(qinv + i));
            uint32_t raw[8];
            to_raw<FrParams>(raw, q);
            uint4 *dst = reinterpret_cast<uint4 *>(q_raw + ((size_t)blockIdx.x * N_BLOB + i) * 8);
            dst[0] = make_uint4(raw[0], raw[1], raw[2], raw[3]);
            dst[1] = make_uint4(raw[4], raw[5], raw[6], raw[7]);
        }
    }
}

int eval_poly_batch_device(DeviceCtx *ctx, Fr *d_y, const Fr *d_poly, const Fr *d_z, size_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_eval_barycentric<false>, dim3((unsigned)n), dim3(EV_THREADS), 0, ctx->stream, d_y,
                       (uint32_t *)nullptr, (int *)nullptr, d_poly, d_z, ctx->d_brp_roots);
    HIP_TRY(hipGetLastError());
    return 0;
}

// y_i = p_i(z_i) and the quotient scalars q (canonical limbs, [n][4096][8]); d_hit[i] >= 0 flags a
// blob whose z lies in the evaluation domain
int eval_quotient_batch_device(DeviceCtx *ctx, Fr *d_y, uint32_t *d_q_raw, int *d_hit, const Fr *d_poly,
                               const Fr *d_z, size_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_eval_barycentric<true>, dim3((unsigned)n), dim3(EV_THREADS), 0, ctx->stream, d_y,
                       d_q_raw, d_hit, d_poly, d_z, ctx->d_brp_roots);
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

__global__ void k_validate_g1(G1Affine *out, uint8_t *status, const uint8_t *in48, size_t n) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint8_t buf[48];
    for (int k = 0; k < 48; k++) buf[k] = in48[g * 48 + k];
    G1Affine a;
    int rc = g1_uncompress(a, buf);
    uint8_t st = rc ? 1 : 0;
    if (!rc && !a.is_inf()) {
        uint32_t r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = FR_R[k];
        G1XYZZ t = affine_mul_w4(a, r);
        if (!t.is_inf()) st = 1;
    }
    if (st) a = G1Affine::inf();
    out[g] = a;
    status[g] = st;
}

int validate_g1_batch_device(DeviceCtx *ctx, G1Affine *d_out, uint8_t *d_status, const uint8_t *d_in48,
                             size_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_validate_g1, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_out,
                       d_status, d_in48, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// variable-base linear combination: out = sum_i k_i * P_i  (k_i canonical 8 x u32)
// ------------------------------------------------------------------------------------------

constexpr int LC_THREADS = 64;

__global__ __launch_bounds__(LC_THREADS) void k_lincomb_partial(G1XYZZ *partials, const G1Affine *pts,
                                                                const uint32_t *scalars, size_t n) {
    __shared__ uint32_t sh[48][LC_THREADS / 2];
    size_t g = blockIdx.x * (size_t)LC_THREADS + threadIdx.x;
    G1XYZZ acc = G1XYZZ::inf();
    if (g < n) {
        uint32_t k[8];
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = scalars[g * 8 + i];
        G1Affine a = pts[g];
        if (!a.is_inf()) acc = affine_mul_w4(a, k);
    }
    const int tid = threadIdx.x;
    for (int s = LC_THREADS / 2; s >= 1; s >>= 1) {
        if (tid >= s && tid < 2 * s) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
#pragma unroll
            for (int k = 0; k < 48; k++) sh[k][tid - s] = src[k];
        }
        __syncthreads();
        if (tid < s) {
            G1XYZZ o;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
            for (int k = 0; k < 48; k++) dst[k] = sh[k][tid];
            acc = xyzz_add(acc, o);
        }
        __syncthreads();
    }
    if (tid == 0) partials[blockIdx.x] = acc;
}

// job j owns partials[part_off[j] .. part_off[j+1]); one lane per job
__global__ void k_lincomb_final(G1Affine *out, const G1XYZZ *partials, const uint32_t *part_off, int njobs) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    G1XYZZ acc = G1XYZZ::inf();
    for (uint32_t i = part_off[j]; i < part_off[j + 1]; i++) acc = xyzz_add(acc, partials[i]);
    bool inf;
    XYZZ28 a = xyzz28_from_xyzz(acc, inf);
    out[j] = xyzz28_to_affine(a, inf);
}

// `total` (a multiple of 64) points/scalars laid out job after job; h_part_off has njobs+1 entries
int lincomb_multi_device(DeviceCtx *ctx, G1Affine *d_out, G1XYZZ *d_partials, const G1Affine *d_pts,
                         const uint32_t *d_scalars, size_t total, const uint32_t *h_part_off, int njobs) {
    uint32_t *d_off = nullptr;
    HIP_TRY(hipMalloc(&d_off, (njobs + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(d_off, h_part_off, (njobs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_lincomb_partial, dim3((unsigned)(total / LC_THREADS)), dim3(LC_THREADS), 0, ctx->stream,
                       d_partials, d_pts, d_scalars, total);
    hipLaunchKernelGGL(k_lincomb_final, dim3((njobs + 63) / 64), dim3(64), 0, ctx->stream, d_out, d_partials,
                       d_off, njobs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_off));
    return 0;
}

// d_out: one affine point; d_partials: scratch for ceil(n/64) XYZZ points
int lincomb_var_device(DeviceCtx *ctx, G1Affine *d_out, G1XYZZ *d_partials, const G1Affine *d_pts,
                       const uint32_t *d_scalars, size_t n) {
    size_t nb = (n + LC_THREADS - 1) / LC_THREADS;
    if (nb == 0) nb = 1;
    uint32_t off[2] = {0, (uint32_t)nb};
    uint32_t *d_off = nullptr;
    HIP_TRY(hipMalloc(&d_off, sizeof off));
    HIP_TRY(hipMemcpyAsync(d_off, off, sizeof off, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_lincomb_partial, dim3((unsigned)nb), dim3(LC_THREADS), 0, ctx->stream, d_partials,
                       d_pts, d_scalars, n);
    hipLaunchKernelGGL(k_lincomb_final, dim3(1), dim3(64), 0, ctx->stream, d_out, d_partials, d_off, 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_off));
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
