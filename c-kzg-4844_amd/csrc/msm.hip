// msm.hip -- fixed-base multi-scalar multiplication over BLS12-381 G1 for gfx950.
//
// Replaces g1_lincomb_fast -> blst_p1s_mult_pippenger (src/common/lincomb.c:65-123) for the MSMs
// whose bases are fixed by the trusted setup: poly_to_kzg_commitment (src/eip4844/eip4844.c:253),
// the quotient commitment in compute_kzg_proof_impl (:484) and the 128 FK20 column MSMs
// (src/eip7594/fk20.c:222-247, where the reference itself switches to fixed-base tables when
// precompute > 0).
//
// MI355X-first design.  A CPU Pippenger spends its time scattering points into data-dependent
// buckets; on a 64-wide machine that scatter is divergence and atomics.  The bases here never
// change and the card has 288 GB of HBM, so the table is widened instead: for every base P_i and
// window w it holds e*2^(c*w)*P_i for e = 1..2^(c-1) in affine form.  After signed-digit
// recoding a scalar vector selects exactly one entry per (window, base), and the MSM is a plain,
// perfectly regular sum of nwin*npoints table entries per blob: coalesced digit reads, one
// 96-byte gather per addition, 8M+2S mixed additions into an XYZZ accumulator held in VGPRs, an
// LDS tree to fold a workgroup, no buckets, no doublings, no atomics.  Arithmetic intensity is
// ~3000 integer multiply-adds per 96-byte gather: the kernel is VALU-bound, not HBM-bound.
#include <chrono>
#include <cstring>
#include <vector>
#include "device.hpp"
#include "dev_inline.hpp"
#include "g1_28.hpp"
#include "g1_quad.hpp"

namespace ckzg {
namespace dev {

int scratch_reserve(DeviceCtx *ctx, size_t bytes) {
    if (ctx->scratch.cap >= bytes) return 0;
    // A slot that has to grow AGAIN grows geometrically: coalesced batches creep upwards with the number of callers
    // (42 -> 84 -> 160 units), and every regrowth is a device-synchronising hipFree plus a hipMalloc -- tens of
    // milliseconds during which every caller of the slot's batch waits (the 65-70 ms worst calls of the round-4
    // 256-caller rows).  Doubling bounds the number of such events per slot by log2(largest / first).
    const size_t old_cap = ctx->scratch.cap;
    if (ctx->scratch.ptr) {
        HIP_TRY(dev::sync_stream(ctx->stream));
        HIP_TRY(hipFree(ctx->scratch.ptr));
        ctx->scratch.ptr = nullptr;
        ctx->scratch.cap = 0;
    }
    size_t want = bytes + (bytes >> 2);
    if (reserve_scaled(bytes) > want) want = reserve_scaled(bytes);   // a batch of a coalescing operation (device.hpp)
    if (old_cap) {
        const size_t step = old_cap < ((size_t)2 << 30) ? old_cap : ((size_t)2 << 30);   // double, by at most 2 GB
        if (want < old_cap + step) want = old_cap + step;
    }
    want = speculative_bytes(bytes + (bytes >> 4), want);   // ahead of need only while HBM is plentiful (device.hpp)
    hipError_t e = hipMalloc(&ctx->scratch.ptr, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the failed attempt must not surface at the next hipGetLastError()
        want = bytes;
        HIP_TRY(hipMalloc(&ctx->scratch.ptr, want));
    }
    ctx->scratch.cap = want;
    return 0;
}

// ------------------------------------------------------------------------------------------
// table construction (setup time)
// ------------------------------------------------------------------------------------------

// wb[w][i] = 2^(wbits*w) * P_i on the four lanes of a DPP quad per point (g1_quad.hpp), in the 28-bit field: the
// (nwin - 1) * wbits sequential doublings are the latency floor of every call-time table build (126 of them for a 6-bit
// table: 2.4 ms with one lane per point whatever the number of points, 0.75-1.0 ms here).  A doubling is three product steps on four lanes instead of
// seven products on one; the conversion of a window base to the XYZZ form the chain builder reads is three more steps,
// each lane finishing one coordinate: zz = z^2 | x | y | zzz = z^3, all multiplied into the 2^384 domain on the way.
__global__ __launch_bounds__(64) void k_window_bases_quad(G1XYZZ *wb, const G1Affine *bases, int npoints, int wbits, int nwin) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 2, ql = gid & 3;
    if (i >= npoints) return;   // (whole quads leave together)
    const G1Affine b = bases[i];
    const bool inf = b.is_inf();
    JAC28 a;
    a.x = widen<1, 34>(f28_from_fp(b.x));
    a.y = widen<1, 34>(f28_from_fp(b.y));
    a.z = widen<2, 4>(f28_one());
    const auto to384 = f28_const<1, 1>(FP28_TO384);
    const int field = ql == 1 ? 0 : (ql == 2 ? 1 : (ql == 0 ? 2 : 3));   // x, y, zz, zzz of G1XYZZ
    for (int w = 0; w < nwin; w++) {
        {
            const auto x = widen<2, 34>(a.x), y = widen<2, 34>(a.y), z = widen<2, 34>(a.z), t = widen<2, 34>(to384);
            const auto p1 = mul(quad::qsel(ql, z, x, y, z), quad::qsel(ql, z, t, t, z));          // zz | x' | y' | zz
            const auto z4 = widen<2, 4>(a.z), t4 = widen<2, 4>(to384);
            const auto p2 = mul(p1, quad::qsel(ql, t4, t4, t4, z4));                               // zz' | - | - | zzz
            const auto p3 = mul(p2, to384);                                                       // -   | - | - | zzz'
            F28<1, 2> mine;
#pragma unroll
            for (int j = 0; j < 14; j++) mine.l[j] = ql == 0 ? p2.l[j] : (ql == 3 ? p3.l[j] : p1.l[j]);
            const Fp out = f28_finish_fp(mine);
            uint32_t *dst = reinterpret_cast<uint32_t *>(wb + (size_t)w * npoints + i) + field * 12;
#pragma unroll
            for (int k = 0; k < 12; k++) dst[k] = inf ? 0u : out.l[k];
        }
        if (w + 1 < nwin) {
            for (int k = 0; k < wbits; k++) quad::jac28_dbl_quad(a, ql);
        }
    }
}

static void enqueue_window_bases(hipStream_t stream, G1XYZZ *d_wb, const G1Affine *d_bases, int npoints, int wbits, int nwin) {
    hipLaunchKernelGGL(k_window_bases_quad, dim3((unsigned)(((size_t)npoints * 4 + 63) / 64)), dim3(64), 0, stream, d_wb, d_bases,
                       npoints, wbits, nwin);
}

// Montgomery simultaneous inversion: each thread normalises a run of L points
// to392: store the coordinates multiplied by 2^8, i.e. in the 2^392 Montgomery domain that the
// 28-bit-limb accumulate kernel works in (fp28.hpp); used for table entries only.
__global__ void k_batch_to_affine(G1Affine *out, const G1XYZZ *in, Fp *prefix, size_t n, int L,
                                  int to392) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t b = t * (size_t)L;
    if (b >= n) return;
    size_t e = b + L < n ? b + L : n;
    Fp acc = Fp::one();
    for (size_t k = b; k < e; k++) {
        prefix[k] = acc;
        Fp z = in[k].zzz;
        if (!z.is_zero()) acc = mul(acc, z);
    }
    // the run's one inversion: safegcd in the 28-bit-limb domain (fp28_inv.hpp), ~12x fewer instructions than the
    // 381-squaring Fermat ladder -- which was 1.4 ms of latency at the end of every small FK20 batch and a quarter
    // of the table-build kernel
    Fp inv = fp_inv_safegcd(acc);
    for (size_t k = e; k-- > b;) {
        G1XYZZ p = in[k];
        if (p.zz.is_zero()) {
            out[k] = G1Affine::inf();
            continue;
        }
        Fp ti = mul(inv, prefix[k]);  // 1/zzz_k
        inv = mul(inv, p.zzz);
        Fp u = mul(p.zz, ti);  // 1/z
        Fp ax = mul(p.x, sqr(u)), ay = mul(p.y, ti);
        if (to392) {
            Fp k8;
#pragma unroll
            for (int i = 0; i < 12; i++) k8.l[i] = FP_MONT_2POW8[i];
            ax = mul(ax, k8);
            ay = mul(ay, k8);
        }
        out[k] = {ax, ay};
    }
}

// ------------------------------------------------------------------------------------------
// table construction, round 3: affine chains with a shared inversion, in the 28-bit-limb field
//
// The entries of one (window, point) chain are an arithmetic progression e*B, e = 1..half, and they are wanted in
// AFFINE form.  The round-1/2 builder walked the chain in XYZZ coordinates (12M+2S per entry on 12 x 32-bit limbs)
// and normalised afterwards (k_batch_to_affine: ~12 more products per entry and a 240-byte temporary).  Here every
// thread owns L independent chain segments and advances all of them one entry per step directly in affine
// coordinates: (e+1)B = eB + B costs lambda = (yB - y)/(xB - x), x3 = lambda^2 - xB - x, y3 = lambda (x - x3) - y,
// i.e. 2M + 1S plus the share of ONE inversion of the product of the L denominators (Montgomery's trick: 3 more
// products per entry; safegcd inversion ~40 product-equivalents per step).  ~6 + 40/L products per entry instead of
// ~26, on the faster field, no temporaries: 238 GB of tables in ~1 s instead of ~3.6 s.  This is batch-affine
// addition in the one place of this library where its operands are streamed exactly once (the accumulate kernels
// cannot use it: DESIGN.md section 10).
// Segments start from seeds (e0+1)*B computed by a short double-and-add (k_table_seeds); segment 0 also gets its
// second entry 2B from there, so that the step kernel never meets the doubling case (e*B = +-B only for e = 1).
// ------------------------------------------------------------------------------------------

// canonical representative in [0, p) of a value < V*p with normalised limbs
template <int V>
__device__ __forceinline__ F28<1, 1> f28_canon(const F28<1, V> &a) {
    static_assert(V <= 16, "value bound");
    uint32_t t[14];
#pragma unroll
    for (int j = 0; j < 14; j++) t[j] = a.l[j];
#pragma unroll
    for (int K = f28detail::pow2_above(V - 1) / 2; K >= 1; K >>= 1) {
        const f28detail::Limbs14 m = f28detail::spread_multiple(K, 0);   // K*p, normalised limbs
        uint32_t d[14], br = 0;
#pragma unroll
        for (int j = 0; j < 14; j++) {
            uint32_t v = t[j] - m.v[j] - br;
            br = v >> 31;   // limbs < 2^28: a negative difference sets bit 31
            d[j] = (j == 13) ? v : (v & M28);
        }
#pragma unroll
        for (int j = 0; j < 14; j++) t[j] = br ? t[j] : d[j];
    }
    F28<1, 1> r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = t[j];
    return r;
}

__device__ __forceinline__ void tb_load12(uint32_t *w, const Fp *src) {
    const uint4 *q = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint4 v = q[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
}
__device__ __forceinline__ void tb_store_point(G1Affine *dst, const F28<1, 1> &x, const F28<1, 1> &y) {
    uint32_t w[24];
    f28_pack<1>(w, x);
    f28_pack<1>(w + 12, y);
    uint4 *q = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int k = 0; k < 6; k++) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// bases[c] (affine, 2^392 domain, canonical; (0,0) = infinity) -> table[c*half + e0] = (e0+1)*B for every segment
// start e0 = s*seg, and table[c*half + 1] = 2B.  One thread per (chain, segment).
__global__ void k_table_seeds(G1Affine *table, const G1Affine *bases, uint32_t nchains, uint32_t half, uint32_t seg) {
    const uint32_t nseg = half / seg;
    const size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (u >= (size_t)nchains * nseg) return;
    const uint32_t c = (uint32_t)(u / nseg), sgm = (uint32_t)(u % nseg);
    uint32_t wx[12], wy[12];
    tb_load12(wx, &bases[c].x);
    tb_load12(wy, &bases[c].y);
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) any |= wx[k] | wy[k];
    G1Affine *row = table + (size_t)c * half;
    if (any == 0) {   // a base at infinity: its whole row is infinity; the step kernel skips it
        G1Affine z = G1Affine::inf();
        row[(size_t)sgm * seg] = z;
        if (sgm == 0 && half > 1) row[1] = z;
        return;
    }
    const F28<1, 1> xb = f28_unpack<1>(wx), yb = f28_unpack<1>(wy);
    const F28<4, 2> yb4 = cneg_reduced(yb, false);
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t m = pass == 0 ? sgm * seg + 1 : 2u;
        if (pass == 1 && !(sgm == 0 && half > 1 && seg > 1)) break;
        XYZZ28 acc;
        bool inf = true;
        for (int bit = 31 - __builtin_clz(m); bit >= 0; bit--) {
            if (!inf) xyzz28_dbl(acc);
            if ((m >> bit) & 1u) xyzz28_madd(acc, inf, xb, yb4);
        }
        // m*B is never infinity (the base has prime order r > m)
        auto t = f28_inv(acc.zzz);   // 1/z^3
        auto zi = mul(acc.zz, t);    // 1/z
        auto ax = mul(acc.x, sqr(zi));
        auto ay = mul(acc.y, t);
        tb_store_point(row + (m - 1), f28_canon<2>(ax), f28_canon<2>(ay));
    }
}

// Every thread advances L chain segments by seg - 1 entries (segment 0: seg - 2, its first two entries are seeds).
template <int L>
__global__ __launch_bounds__(64) void k_table_steps(G1Affine *table, const G1Affine *bases, uint32_t nchains, uint32_t half,
                                                    uint32_t seg) {
    const uint32_t nseg = half / seg;
    const size_t total = (size_t)nchains * nseg;
    const size_t u0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * L;
    if (u0 >= total) return;
    G1Affine *cur[L];          // the last entry written for unit k
    const G1Affine *base[L];
    uint32_t left[L];          // entries still to write
#pragma unroll
    for (int k = 0; k < L; k++) {
        const size_t u = u0 + k;
        left[k] = 0;
        cur[k] = table;
        base[k] = bases;
        if (u < total) {
            const uint32_t c = (uint32_t)(u / nseg), sgm = (uint32_t)(u % nseg);
            const uint32_t first = sgm == 0 && seg > 1 ? 1u : 0u;   // entries [e0, e0 + first] exist already
            base[k] = bases + c;
            cur[k] = table + (size_t)c * half + (size_t)sgm * seg + first;
            uint32_t wb[12], wy[12], any = 0;
            tb_load12(wb, &bases[c].x);
            tb_load12(wy, &bases[c].y);
#pragma unroll
            for (int j = 0; j < 12; j++) any |= wb[j] | wy[j];
            left[k] = any ? seg - 1 - first : 0;
            if (!any) {   // a base at infinity (never in a ceremony file, legal in the format): its row is all infinity
                const G1Affine z = G1Affine::inf();
                for (uint32_t e = first + 1; e < seg; e++) cur[k][e - first] = z;
            }
        }
    }
    for (uint32_t step = 0; step + 1 < seg; step++) {
        F28<1, 2> pre[L];
        // forward: running product of the denominators xB - x
#pragma unroll
        for (int k = 0; k < L; k++) {
            F28<4, 3> d = widen<4, 3>(f28_one());
            if (step < left[k]) {
                uint32_t wx[12], wb[12];
                tb_load12(wx, &cur[k]->x);
                tb_load12(wb, &base[k]->x);
                d = sub(f28_unpack<1>(wb), f28_unpack<1>(wx));
            }
            pre[k] = k == 0 ? mul(d, f28_one()) : mul(pre[k - 1], d);
        }
        F28<1, 2> inv = f28_inv(pre[L - 1]);
        // backward: peel the inverses off, finish each addition, write the next entry
#pragma unroll
        for (int k = L - 1; k >= 0; k--) {
            if (step < left[k]) {   // (inactive units contributed d = 1: nothing to peel)
                uint32_t wx[12], wy[12], wbx[12], wby[12];
                tb_load12(wx, &cur[k]->x);
                tb_load12(wy, &cur[k]->y);
                tb_load12(wbx, &base[k]->x);
                tb_load12(wby, &base[k]->y);
                const F28<1, 1> x = f28_unpack<1>(wx), y = f28_unpack<1>(wy), xb = f28_unpack<1>(wbx), yb = f28_unpack<1>(wby);
                const F28<4, 3> d = sub(xb, x);
                const F28<1, 2> dinv = k == 0 ? inv : mul(inv, pre[k - 1]);
                if (k > 0) inv = mul(inv, d);
                const F28<1, 2> lam = mul(sub(yb, y), dinv);             // (yB - y) / (xB - x)
                const F28<1, 1> x3 = f28_canon<6>(norm(sub(sub(sqr(lam), xb), x)));   // <7,6> -> [0, p)
                const F28<1, 1> y3 = f28_canon<4>(norm(sub(mul(lam, sub(x, x3)), y)));   // <4,4> -> [0, p)
                cur[k] += 1;
                tb_store_point(cur[k], x3, y3);
            }
        }
    }
}

int batch_to_affine_device(DeviceCtx *ctx, G1Affine *d_out, const G1XYZZ *d_in, Fp *d_prefix, size_t n) {
    if (n == 0) return 0;
    // short runs when there are few points (latency), long runs when there are many (throughput); up to 4096 points
    // (the proofs of a batch of <= 32 blobs) every point has a lane and an inversion of its own: the lanes are idle
    // anyway and the run of four cost 207 us against ~36 us for the one inversion it contains
    static const size_t solo_max = (size_t)ab_knob("CKZG_HIP_TO_AFFINE_SOLO_MAX", 4096);
    int L = n >= ((size_t)1 << 20) ? 128 : (n >= ((size_t)1 << 14) ? 16 : (n > solo_max ? 4 : 1));
    size_t threads = (n + L - 1) / L;
    hipLaunchKernelGGL(k_batch_to_affine, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, ctx->stream,
                       d_out, d_in, d_prefix, n, L, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}


// The launches of the affine-chain builder on an explicit stream: window bases (already in d_wb, XYZZ) to affine in
// the 2^392 domain, segment seeds, segment steps.  d_wba: twin*npoints affine points, d_prefix: as many Fp.
static int enqueue_affine_chain_table(hipStream_t stream, const FixedBaseTable &t, G1Affine *d_table, const G1XYZZ *d_wb,
                                      G1Affine *d_wba, Fp *d_prefix, const std::atomic<bool> *cancel) {
    const size_t nchains = (size_t)t.twin * t.npoints;
    {
        const int Lb = 16;
        const size_t th = (nchains + Lb - 1) / Lb;
        hipLaunchKernelGGL(k_batch_to_affine, dim3((unsigned)((th + 63) / 64)), dim3(64), 0, stream, d_wba, d_wb, d_prefix,
                           nchains, Lb, 1);
    }
    // segment length: ~4096 waves of 8-segment threads when the table is large enough, 16..512 entries
    constexpr int LSEG = 8;
    const size_t entries = nchains * t.half;
    size_t seg = entries / ((size_t)262144 * LSEG);
    uint32_t seg2 = 16;
    while (seg2 < 512 && seg2 * 2 <= seg) seg2 *= 2;
    if (seg2 > t.half) seg2 = (uint32_t)t.half;
    const uint32_t nseg = (uint32_t)(t.half / seg2);
    // a background build that may be cancelled goes four windows at a time (a launch of the whole table cannot be
    // abandoned; one window alone -- 512 waves at 16 bits -- leaves half the chip idle)
    const size_t chains_per_launch = cancel ? (size_t)4 * t.npoints : nchains;
    for (size_t c0 = 0; c0 < nchains; c0 += chains_per_launch) {
        if (cancel && cancel->load(std::memory_order_relaxed)) {
            (void)dev::sync_stream(stream);
            return 5;
        }
        if (cancel && c0) HIP_TRY(dev::sync_stream(stream));
        const size_t nc = nchains - c0 < chains_per_launch ? nchains - c0 : chains_per_launch;
        const size_t units = nc * nseg, threads = (units + LSEG - 1) / LSEG;
        hipLaunchKernelGGL(k_table_seeds, dim3((unsigned)((units + 63) / 64)), dim3(64), 0, stream, d_table + c0 * t.half,
                           d_wba + c0, (uint32_t)nc, (uint32_t)t.half, seg2);
        if (seg2 > 1)
            hipLaunchKernelGGL(k_table_steps<LSEG>, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, stream,
                               d_table + c0 * t.half, d_wba + c0, (uint32_t)nc, (uint32_t)t.half, seg2);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int build_fixed_base_table(DeviceCtx *ctx, FixedBaseTable *t, const G1Affine *d_bases, int npoints,
                           int wbits, double *times_ms, const std::atomic<bool> *cancel) {
    if (wbits < 2 || wbits > 16) return 1;
    const auto t_start = std::chrono::steady_clock::now();
    t->npoints = npoints;
    t->wbits = wbits;
    t->twin = FixedBaseTable::twin_for(wbits);
    t->nwin = 2 * t->twin;
    t->half = (size_t)1 << (wbits - 1);
    DevTmp wb, prefix, table, wba;   // `table` is handed to *t only when the build has completed
    t->d_table = nullptr;
    HIP_TRY(hipMalloc(&table.p, t->bytes()));
    G1Affine *d_table = static_cast<G1Affine *>(table.p);
    HIP_TRY(hipMalloc(&wb.p, (size_t)t->twin * npoints * sizeof(G1XYZZ)));
    G1XYZZ *d_wb = static_cast<G1XYZZ *>(wb.p);
    const auto t_alloc = std::chrono::steady_clock::now();
    enqueue_window_bases(ctx->stream, d_wb, d_bases, npoints, wbits, t->twin);
    {
        // affine chains with a shared inversion (k_table_seeds / k_table_steps): entries are written once, in their
        // final form; 13x faster than the round-1/2 builder (XYZZ chains + a normalisation pass, profiles/r03_load_ab.jsonl)
        const size_t nchains = (size_t)t->twin * npoints;
        HIP_TRY(hipMalloc(&wba.p, nchains * sizeof(G1Affine)));
        HIP_TRY(hipMalloc(&prefix.p, nchains * sizeof(Fp)));
        int rc = enqueue_affine_chain_table(ctx->stream, *t, d_table, d_wb, static_cast<G1Affine *>(wba.p),
                                            static_cast<Fp *>(prefix.p), cancel);
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(dev::sync_stream(ctx->stream));
    t->d_table = d_table;
    table.p = nullptr;
    if (times_ms) {
        const auto t_end = std::chrono::steady_clock::now();
        times_ms[0] += std::chrono::duration<double, std::milli>(t_alloc - t_start).count();
        times_ms[1] += std::chrono::duration<double, std::milli>(t_end - t_alloc).count();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// scalar recoding
// ------------------------------------------------------------------------------------------

// One thread per field element of the batch: big-endian bytes -> canonical check (blob.c:31-38 /
// bytes.c:64-70: a value >= r makes the whole blob BADARGS) -> balanced GLV split -> signed digits
// digits[blob][w][i]: windows 0..twin-1 from k2 (the phi half), twin..2*twin-1 from k1.
__global__ void k_blob_digits(int16_t *digits, uint32_t *bad, const uint8_t *blobs, size_t total,
                              int wbits, int twin) {
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= total) return;
    size_t blob = gid >> 12;
    uint32_t i = (uint32_t)(gid & 4095);
    uint32_t s[8], r[8];
    load_be256(s, blobs + gid * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = FR_R[k];
    if (limbs_geq<8>(s, r)) {
        atomicOr(&bad[blob], 1u);
        // the blob's output is unspecified, but its digits must stay inside the table
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = 0;
    }
    glv_digits(digits + blob * (size_t)(2 * twin) * N_BLOB + i, N_BLOB, s, wbits, twin);
}

// Same for scalars that are already canonical little-endian integers ([n][4096][8] u32)
__global__ void k_raw_digits(int16_t *digits, const uint32_t *scalars, size_t total, int wbits,
                             int twin) {
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= total) return;
    size_t vec = gid >> 12;
    uint32_t i = (uint32_t)(gid & 4095);
    uint32_t s[8];
    const uint4 *q = reinterpret_cast<const uint4 *>(scalars + gid * 8);
    uint4 a = q[0], b = q[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
    s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    glv_digits(digits + vec * (size_t)(2 * twin) * N_BLOB + i, N_BLOB, s, wbits, twin);
}

// ------------------------------------------------------------------------------------------
// accumulate: the dominant kernel
// ------------------------------------------------------------------------------------------


// phi(X, Y, ZZ, ZZZ) = (beta X, Y, ZZ, ZZZ) = [lambda](X, Y, ZZ, ZZZ) on G1: turns the sum of the k2-half
// terms, which were gathered from the plain table, into the sum over the phi-mapped bases.
__device__ __forceinline__ void msm_apply_phi(XYZZ28 &acc28) {
    acc28.x = widen<1, 10>(mul(acc28.x, f28_const<1, 1>(FP28_BETA_LAMBDA)));
}

// Alternating priority slices between the two waves of a SIMD.
// What the per-wave trace of round 3 showed (tools/msm_trace.py, profiles/r03_msm_trace_*.json): of two resident
// waves the OLDER one is served first -- it gets ~88 % of the issue slots and finishes its 256 additions in 2.8 ms,
// its partner needs 5.3 ms -- so a launch of exactly two rounds of workgroups spends its last quarter with one wave
// per SIMD (SQ_WAVE_CYCLES: 1.72 waves resident on average).  But that lone wave runs at 98 % of the pair's combined
// rate (0.392 vs 0.401 wave-additions per ms and SIMD): the SIMD's issue slots are already full with one wave, and
// residency is NOT what bounds this kernel (finer work items raise it to 1.89 and gain nothing, profiles/r03_ppb_ab.txt).
// What does help, measurably: s_setprio slices.  A wave raises its priority in the time slices whose parity equals
// the parity of its wave slot in the SIMD (two resident waves sit in slots 0 and 1) and lowers it in the others, so
// the two waves take turns at running nearly alone.  Slices of 2^15..2^17 ticks of the 100 MHz s_memrealtime counter
// (0.3-1.3 ms) give -3 % kernel time in interleaved same-box runs; 2^13 and shorter, or 2^19 and longer, give
// nothing (profiles/r03_prio_ab.txt).  The mechanism is not established -- the likeliest is instruction-cache
// locality: the loop body is ~37 KB of code, and two waves at unrelated positions in it compete for the cache their
// CU pair shares, while a wave that runs nearly alone for a slice streams through it undisturbed.
// prio_bit = log2 of the slice width in ticks; 0 switches the scheme off.
__device__ __forceinline__ uint32_t msm_wave_slot_parity() {
    return __builtin_amdgcn_s_getreg(4) & 1u;   // hwreg(HW_REG_HW_ID, 0, 1): bit 0 of WAVE_ID
}
__device__ __forceinline__ void msm_fair_prio(uint32_t prio_bit, uint32_t parity) {
    if (prio_bit == 0) return;
    const uint32_t slice = (uint32_t)(wall_clock64() >> prio_bit) & 1u;
    if (slice == parity)
        __builtin_amdgcn_s_setprio(2);
    else
        __builtin_amdgcn_s_setprio(0);
}

// One lane's share of a fixed-base sum: pairs q = first, first + STRIDE, ... < q1 of one vector (q = w*ppv + i;
// pairs below phi_pairs = twin*ppv belong to the k2 half, so the accumulator goes through phi once, when the lane
// crosses that boundary or at the end if it never does).
// Software pipeline: in the iteration that adds pair q the table entry of pair q + STRIDE and the digit of pair
// q + 2 STRIDE are already requested, so a wave never waits for its own gather (first iteration: nothing to add,
// it only fills the pipeline).  e[] is written only by the copies at the END of an iteration: that is where the
// compiler's s_waitcnt vmcnt(0) lands (tools/isa_waits.py) -- a first version that loaded e[] in a prologue got the
// wait in FRONT of the addition, because vmcnt counts in order, and measured slower than no prefetch at all.
// Same-box A/B on the headline kernel: 9.76 -> 9.50 ms per 1024 blobs (profiles/r02_fp28_ab.txt).
template <int STRIDE>
__device__ __forceinline__ void msm_sum_pairs(XYZZ28 &acc28, bool &inf, bool &yneg, const G1Affine *table,
                                              const int16_t *dg, uint32_t first, uint32_t q1, uint32_t phi_pairs,
                                              uint32_t twin, uint32_t ppv, uint32_t npoints, uint32_t voff,
                                              int half_shift, uint32_t prio_bit = 0) {
    bool phi_pending = first < phi_pairs;
    const uint32_t slot_parity = prio_bit ? msm_wave_slot_parity() : 0u;
    uint32_t qn = first;
    int d = 0, d1 = qn < q1 ? dg[qn] : 0;
    uint4 e[6];
#pragma unroll
    for (int k = 0; k < 6; k++) e[k] = make_uint4(0, 0, 0, 0);
    for (; qn < q1 + STRIDE; qn += STRIDE) {
        msm_fair_prio(prio_bit, slot_parity);
        uint4 nx[6];
#pragma unroll
        for (int k = 0; k < 6; k++) nx[k] = make_uint4(0, 0, 0, 0);
        if (d1 != 0) {
            uint32_t w = qn / ppv, i = qn - w * ppv;
            uint32_t tw = w >= twin ? w - twin : w;
            uint32_t mag = (uint32_t)(d1 < 0 ? -d1 : d1);
            const uint4 *src = reinterpret_cast<const uint4 *>(
                table + ((((size_t)tw * npoints + voff + i) << half_shift) + (mag - 1)));
#pragma unroll
            for (int k = 0; k < 6; k++) nx[k] = src[k];
        }
        int d2 = qn + STRIDE < q1 ? dg[qn + STRIDE] : 0;
        if (phi_pending && qn >= phi_pairs + STRIDE) {
            if (!inf) msm_apply_phi(acc28);
            phi_pending = false;
        }
        if (d != 0) {
            uint32_t wd[24];
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                wd[4 * k] = e[k].x; wd[4 * k + 1] = e[k].y; wd[4 * k + 2] = e[k].z; wd[4 * k + 3] = e[k].w;
                any |= e[k].x | e[k].y | e[k].z | e[k].w;
            }
            // (0,0) encodes a table entry at infinity
            if (any != 0) xyzz28_madd_alt(acc28, inf, yneg, f28_unpack<1>(wd), f28_unpack<1>(wd + 12), d < 0);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) e[k] = nx[k];
        d = d1;
        d1 = d2;
    }
    if (phi_pending && !inf) msm_apply_phi(acc28);
}

#ifdef CKZG_MSM_TRACE
// Diagnostic build (tools/build_variant.sh trace -DCKZG_MSM_TRACE; never in the product): every wave of
// k_msm_accumulate records when and where it ran -- {s_memrealtime at entry, at exit, HW_ID, XCC_ID} -- so that
// tools/msm_trace.py can draw the occupancy of every SIMD over the launch (why SQ_WAVE_CYCLES says 1.7 waves
// per SIMD where the register budget allows 2).  Dumped to $CKZG_HIP_MSM_TRACE_FILE by commit_blobs_device.
__device__ uint64_t *g_msm_trace = nullptr;
__device__ __forceinline__ void msm_trace_mark(uint32_t wave_slot, int which) {
    if ((threadIdx.x & 63) != 0 || !g_msm_trace) return;
    uint64_t *rec = g_msm_trace + (size_t)wave_slot * 4;
    rec[which] = wall_clock64();
    if (which == 0) {
        rec[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        rec[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
}
#endif

// grid: nvec * blocks_per_vec workgroups.  A "vector" is one MSM: ppv (points per vector) scalars
// recoded to digits[vec][w][i], i < ppv, w < nwin = 2*twin.  Its bases are points voff..voff+ppv of a
// table over npoints bases, voff = (vec % vecs_per_group) * ppv  (commitment: ppv = npoints = 4096,
// one group; FK20: ppv = 64, 128 vectors per blob over the 8192 x_ext_fft points).
// Workgroup (v, c) sums the table entries selected by pairs q in [c*ppb, (c+1)*ppb) of vector v,
// q = w*ppv + i, into partials[v*blocks_per_vec + c].  Pairs below phi_pairs = twin*ppv belong to the
// k2 half: a thread walks its pairs in ascending order, so it maps its accumulator through phi once,
// when it crosses that boundary (or at the end, if it never does).
// RAW is a template parameter, not a run-time branch: with `if (raw_out)` in one kernel the register allocator
// kept the LDS fold's result live across both stores and the throughput form went from 249 to 307 unified VGPRs
// (one wave per SIMD instead of two; round-4 regression, bisected by the judge).  tests/test_kernel_resources.py
// pins both instantiations at <= 256.
template <int THREADS, bool RAW>
__global__ __launch_bounds__(THREADS) void k_msm_accumulate(
    G1XYZZ *partials, const G1Affine *table, const int16_t *digits, uint32_t pairs_per_vec,
    uint32_t pairs_per_block, int half_shift, uint32_t blocks_per_vec, uint32_t ppv,
    uint32_t npoints, uint32_t vecs_per_group, uint32_t part_stride, uint32_t prio_bit) {
    __shared__ uint32_t sh[57][THREADS];
#ifdef CKZG_MSM_TRACE
    msm_trace_mark(blockIdx.x * (THREADS / 64) + threadIdx.x / 64, 0);
#endif
    const uint32_t vec = blockIdx.x / blocks_per_vec, chunk = blockIdx.x % blocks_per_vec;
    const uint32_t q0 = chunk * pairs_per_block;
    const uint32_t q1 = q0 + pairs_per_block < pairs_per_vec ? q0 + pairs_per_block : pairs_per_vec;
    const uint32_t voff = (vec % vecs_per_group) * ppv;
    const uint32_t phi_pairs = pairs_per_vec >> 1, twin = phi_pairs / ppv;
    const int16_t *dg = digits + (size_t)vec * pairs_per_vec;
    // accumulator in the 28-bit-limb / 2^392 domain (fp28.hpp), infinity tracked by a flag
    XYZZ28 acc28;
    bool inf = true, yneg = false;  // yneg: acc28.y currently holds -Y (xyzz28_madd_alt)
    // (Measured alternatives: unrolling by two -- 14.4 ms, two copies of the ~40 KB addition body thrash the
    // instruction cache; capping VGPRs for 3 or 4 waves per SIMD -- 11.4 / 13.2 ms.)
    msm_sum_pairs<THREADS>(acc28, inf, yneg, table, dg, q0 + threadIdx.x, q1, phi_pairs, twin, ppv, npoints, voff,
                           half_shift, prio_bit);
    if (prio_bit) __builtin_amdgcn_s_setprio(0);
    xyzz28_fix_sign(acc28, inf, yneg);
    quad::block_reduce_xyzz28_quad<THREADS>(acc28, inf, sh);   // four lanes per pair: the fold is ~3x shorter
    if constexpr (RAW) {
        // latency form (k_msm_fold_finalize reads it): the sum as it stands in LDS -- 56 limbs of the 28-bit domain and
        // the infinity flag -- with no conversion to the reduced 12-limb form and back; `partials` is the raw buffer
        uint32_t *raw_out = reinterpret_cast<uint32_t *>(partials);
        if (threadIdx.x < 57) raw_out[((size_t)vec * part_stride + chunk) * 57 + threadIdx.x] = sh[threadIdx.x][0];
    } else {
        if (threadIdx.x == 0) partials[(size_t)vec * part_stride + chunk] = xyzz28_to_xyzz(acc28, inf);
    }
#ifdef CKZG_MSM_TRACE
    msm_trace_mark(blockIdx.x * (THREADS / 64) + threadIdx.x / 64, 1);
#endif
}

// Many small MSMs (FK20: 128 vectors of 64 points per blob).  A 64-lane workgroup serves 64/LPV
// vectors, LPV lanes each: with LPV = 16 a lane does ~90 additions before the 4-level fold instead
// of ~22 before a 6-level fold, so the reduction tree shrinks from ~40 % to ~6 % of the work.
// Results stay in (fully reduced) XYZZ form in out[nvec].
template <int LPV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_msm_small(G1XYZZ *out, const G1Affine *table, const int16_t *digits,
                                                 uint32_t nvec, uint32_t pairs_per_vec, int half_shift,
                                                 uint32_t ppv, uint32_t npoints, uint32_t vecs_per_group, uint32_t prio_bit) {
    __shared__ uint32_t sh[57][LPV == 64 ? 64 : 32];
    constexpr int GROUPS = 64 / LPV;
    const int tid = threadIdx.x, grp = tid / LPV, l = tid % LPV;
    const uint32_t vec = blockIdx.x * GROUPS + grp;
    const uint32_t phi_pairs = pairs_per_vec >> 1, twin = phi_pairs / ppv;
    XYZZ28 acc28;
    bool inf = true, yneg = false;
    if (vec < nvec) {
        const uint32_t voff = (vec % vecs_per_group) * ppv;
        const int16_t *dg = digits + (size_t)vec * pairs_per_vec;
        msm_sum_pairs<LPV>(acc28, inf, yneg, table, dg, (uint32_t)l, pairs_per_vec, phi_pairs, twin, ppv, npoints,
                           voff, half_shift, prio_bit);
    }
    if (prio_bit) __builtin_amdgcn_s_setprio(0);
    xyzz28_fix_sign(acc28, inf, yneg);
    if constexpr (LPV == 64) {
        // one wave per vector is the latency form (few vectors): fold on four lanes per pair.  (In the throughput
        // forms the quad fold pushes the kernel past 256 VGPRs: measured slower, profiles/r02_fp28_ab.txt.)
        quad::block_reduce_xyzz28_quad<64>(acc28, inf, sh);
        if (l == 0 && vec < nvec) out[vec] = xyzz28_to_xyzz(acc28, inf);
        return;
    }
    for (int s = LPV / 2; s >= 1; s >>= 1) {
        if (l >= s && l < 2 * s) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc28);
            const int slot = grp * (LPV / 2) + (l - s);
#pragma unroll
            for (int k = 0; k < 56; k++) sh[k][slot] = src[k];
            sh[56][slot] = inf ? 1u : 0u;
        }
        __syncthreads();
        if (l < s) {
            XYZZ28 o;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&o);
            const int slot = grp * (LPV / 2) + l;
#pragma unroll
            for (int k = 0; k < 56; k++) dst[k] = sh[k][slot];
            xyzz28_add(acc28, inf, o, sh[56][slot] != 0);
        }
        __syncthreads();
    }
    if (l == 0 && vec < nvec) out[vec] = xyzz28_to_xyzz(acc28, inf);
}

// Fold many per-workgroup partials of one vector into one: a 64-lane workgroup per vector,
// lane-parallel accumulation then an LDS tree.  Only used when a vector has more than a few
// partials (small batches that were split finely to fill the chip).
__global__ __launch_bounds__(64) void k_msm_reduce_partials(G1XYZZ *sums, const G1XYZZ *partials,
                                                           uint32_t blocks_per_vec) {
    __shared__ uint32_t sh[57][64];
    const size_t v = blockIdx.x;
    const int tid = threadIdx.x;
    XYZZ28 acc;
    bool inf = true;
    for (uint32_t j = tid; j < blocks_per_vec; j += 64) {
        bool oinf;
        XYZZ28 o = xyzz28_from_xyzz(partials[v * blocks_per_vec + j], oinf);
        xyzz28_add(acc, inf, o, oinf);
    }
    quad::block_reduce_xyzz28_quad<64>(acc, inf, sh);
    if (tid == 0) sums[v] = xyzz28_to_xyzz(acc, inf);
}

// One LANE per vector, packed into full waves: sum the (few) partials, normalise with a Fermat
// inversion, compress (bytes.c:42-44).  Measured (tools/ubench/inv_bench.hip): 1024 inversions
// take 0.80 ms as 16 dense waves but 1.8-3.1 ms as 1024 single-lane waves, so the inversions are
// packed even though that serialises the short partial sum.
__global__ __launch_bounds__(64) void k_msm_finalize(uint8_t *out48, uint8_t *status, const G1XYZZ *partials,
                                                    const uint32_t *bad, uint32_t blocks_per_vec, size_t n) {
    size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (v >= n) return;
    XYZZ28 acc;
    bool inf = true;
    for (uint32_t j = 0; j < blocks_per_vec; j++) {
        bool oinf;
        XYZZ28 o = xyzz28_from_xyzz(partials[v * blocks_per_vec + j], oinf);
        xyzz28_add(acc, inf, o, oinf);
    }
    G1Affine a = xyzz28_to_affine(acc, inf);
    uint8_t buf[48];
    g1_compress_affine(buf, a);
    for (int k = 0; k < 48; k++) out48[v * 48 + k] = buf[k];
    if (status) status[v] = (bad && bad[v]) ? 1 : 0;
}

// The same with LPV lanes per vector: each lane takes one of the (2..8) partials, an LDS tree folds them in
// log2(LPV) additions instead of a chain of bpv - 1, lane 0 of the group normalises.  For the few hundred
// vectors of a latency-bound call (one blob's 128 cell proofs: 4 partials each) the chain was a third of the
// kernel's time.
template <int LPV>
__global__ __launch_bounds__(64) void k_msm_finalize_tree(uint8_t *out48, uint8_t *status, const G1XYZZ *partials,
                                                         const uint32_t *bad, uint32_t blocks_per_vec, size_t n) {
    __shared__ uint32_t sh[57][32];
    constexpr int GROUPS = 64 / LPV;
    const int tid = threadIdx.x, grp = tid / LPV, l = tid % LPV;
    const size_t v = blockIdx.x * (size_t)GROUPS + grp;
    XYZZ28 acc;
    bool inf = true;
    if (v < n && (uint32_t)l < blocks_per_vec) acc = xyzz28_from_xyzz(partials[v * blocks_per_vec + l], inf);
    for (int s = LPV / 2; s >= 1; s >>= 1) {
        if (l >= s && l < 2 * s) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&acc);
            const int slot = grp * (LPV / 2) + (l - s);
#pragma unroll
            for (int k = 0; k < 56; k++) sh[k][slot] = src[k];
            sh[56][slot] = inf ? 1u : 0u;
        }
        __syncthreads();
        if (l < s) {
            XYZZ28 o;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&o);
            const int slot = grp * (LPV / 2) + l;
#pragma unroll
            for (int k = 0; k < 56; k++) dst[k] = sh[k][slot];
            xyzz28_add(acc, inf, o, sh[56][slot] != 0);
        }
        __syncthreads();
    }
    if (l == 0 && v < n) {
        G1Affine a = xyzz28_to_affine(acc, inf);
        uint8_t buf[48];
        g1_compress_affine(buf, a);
        for (int k = 0; k < 48; k++) out48[v * 48 + k] = buf[k];
        if (status) status[v] = (bad && bad[v]) ? 1 : 0;
    }
}

// Latency form of the two kernels above, for the few vectors of a one-blob call whose sums were cut into hundreds of
// partial sums to fill the chip: ONE 512-thread workgroup per vector folds its bpv <= 512 raw partial sums
// (k_msm_accumulate's raw_out) on 128 DPP quads -- every level of the tree is a four-step quad addition, ~4 us,
// where k_msm_reduce_partials opened with two one-lane additions of 15 us each -- and its first lane normalises and
// compresses.  One launch instead of two, no conversion of the partial sums out of the 28-bit domain and back.
#ifdef CKZG_FOLD_TRACE
__device__ uint64_t g_fold_trace[16];
#define FOLD_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_fold_trace[i] = wall_clock64(); } while (0)
#else
#define FOLD_STAMP(i) ((void)0)
#endif
constexpr int FOLD_THREADS = 512;   // 128 quads, two waves per SIMD: the full register file per lane (no spills in the inversion)
__global__ __launch_bounds__(FOLD_THREADS) void k_msm_fold_finalize(uint8_t *out48, uint8_t *status, const uint32_t *raw,
                                                                   const uint32_t *bad, uint32_t bpv) {
    __shared__ uint32_t sh[57][256];
    FOLD_STAMP(0);
    const int tid = threadIdx.x, ql = tid & 3, quad_id = tid >> 2;
    const size_t v = blockIdx.x;
    const uint32_t *rv = raw + v * (size_t)bpv * 57;
    uint32_t cnt = bpv, h = (cnt + 1) / 2;
    for (uint32_t pr = (uint32_t)quad_id; pr < h; pr += FOLD_THREADS / 4) {
        XYZZ28 x, y;
        uint32_t *dx = reinterpret_cast<uint32_t *>(&x), *dy = reinterpret_cast<uint32_t *>(&y);
        const uint32_t *px = rv + (size_t)pr * 57;
        const bool has_y = pr + h < cnt;
        const uint32_t *py = rv + (size_t)(has_y ? pr + h : pr) * 57;
#pragma unroll
        for (int k = 0; k < 56; k++) {
            dx[k] = px[k];
            dy[k] = py[k];
        }
        bool xi = px[56] != 0;
        const bool yi = !has_y || py[56] != 0;
        quad::xyzz28_add_quad(x, xi, y, yi, ql);
        if (ql == 0) {
#pragma unroll
            for (int k = 0; k < 56; k++) sh[k][pr] = dx[k];
            sh[56][pr] = xi ? 1u : 0u;
        }
    }
    __syncthreads();
    FOLD_STAMP(1);
    cnt = h;
    while (cnt > 1) {
        h = (cnt + 1) / 2;
        for (uint32_t pr = (uint32_t)quad_id; pr < cnt / 2; pr += FOLD_THREADS / 4) {
            XYZZ28 x, y;
            uint32_t *dx = reinterpret_cast<uint32_t *>(&x), *dy = reinterpret_cast<uint32_t *>(&y);
#pragma unroll
            for (int k = 0; k < 56; k++) {
                dx[k] = sh[k][pr];
                dy[k] = sh[k][pr + h];
            }
            bool xi = sh[56][pr] != 0;
            const bool yi = sh[56][pr + h] != 0;
            quad::xyzz28_add_quad(x, xi, y, yi, ql);
            if (ql == 0) {
#pragma unroll
                for (int k = 0; k < 56; k++) sh[k][pr] = dx[k];
                sh[56][pr] = xi ? 1u : 0u;
            }
        }
        __syncthreads();
        cnt = h;
    }
    FOLD_STAMP(2);
    if (tid == 0) {
        XYZZ28 acc;
        uint32_t *dst = reinterpret_cast<uint32_t *>(&acc);
#pragma unroll
        for (int k = 0; k < 56; k++) dst[k] = sh[k][0];
#ifdef CKZG_FOLD_TRACE
        auto tinv = f28_inv(acc.zzz);
        FOLD_STAMP(3);
        if (tinv.l[0] == 0xdeadbeef) out48[0] = 1;   // keep it alive
#endif
        G1Affine a = xyzz28_to_affine(acc, sh[56][0] != 0);
        FOLD_STAMP(4);
        uint8_t buf[48];
        g1_compress_affine(buf, a);
        FOLD_STAMP(5);
        for (int k = 0; k < 48; k++) out48[v * 48 + k] = buf[k];
        if (status) status[v] = (bad && bad[v]) ? 1 : 0;
        FOLD_STAMP(6);
    }
}

// How many (window, point) pairs one 256-thread workgroup of k_msm_accumulate sums.  The chip holds
// 512 such workgroups at once (2 per CU at ~200 VGPRs); a launch is `rounds` waves of resident
// workgroups, each costing its threads' additions plus the 8-level LDS tree, so the choice trades
// tail effect against tree overhead:  cost = rounds * (pairs_per_thread * ADD + TREE).
static uint32_t pick_pairs_per_block(size_t nvec, uint32_t pairs_per_vec) {
    static const long forced = ab_knob("CKZG_HIP_PPB", 0);
    if (forced >= 256) return (uint32_t)forced;
    // the fold: 9 passes of a four-step quad addition (g1_quad.hpp), ~400 multiply-adds a step; it overlaps with
    // the CU's other workgroup (the one-lane tree of round 1 cost 20440)
    static const double TREE = (double)ab_knob("CKZG_HIP_TREE_COST", 9 * 4 * 400 / 2);
    const double ADD = 3542.0;
    const size_t resident = 512;
    uint32_t best_ppb = pairs_per_vec;
    double best = 1e300;
    for (uint32_t bpv = 1; bpv <= 512; bpv++) {
        uint32_t ppb = ((pairs_per_vec + bpv - 1) / bpv + 255) / 256 * 256;
        if (ppb < 512 && bpv > 1) break;
        uint32_t real_bpv = (pairs_per_vec + ppb - 1) / ppb;
        size_t rounds = (nvec * real_bpv + resident - 1) / resident;
        double cost = (double)rounds * ((double)(ppb / 256) * ADD + TREE) + real_bpv * 300.0;
        if (cost < best) {
            best = cost;
            best_ppb = ppb;
        }
    }
    return best_ppb;
}

// The fixed-base sums of a call with a handful of vectors (the reference-shaped one-blob calls: blob_to_kzg_commitment,
// compute_kzg_proof, compute_blob_kzg_proof) are latency: nothing else is on the device, so the cut is the FINEST
// that keeps every workgroup resident at once (512 of them) -- with one pair per thread a lane's "addition" is a copy
// and the workgroup is its nine-round quad fold; the partial sums then go through k_msm_fold_finalize.  Larger
// batches are throughput and take the cost model above.
static uint32_t pick_pairs_per_block_fixed(size_t nvec, uint32_t pairs_per_vec) {
    for (uint32_t ppb = 256; ppb <= 1024; ppb *= 2) {
        const size_t bpv = (pairs_per_vec + ppb - 1) / ppb;
        if (nvec * bpv <= 512 && bpv > 8) return ppb;
    }
    return pick_pairs_per_block(nvec, pairs_per_vec);
}
// bytes of partial sums of nvec vectors cut into bpv blocks each, plus one reduced sum per vector: raw 57-word records
// when k_msm_fold_finalize folds them (bpv > 8), reduced XYZZ points otherwise
static size_t partials_bytes(size_t nvec, uint32_t bpv) {
    return bpv > 8 ? nvec * ((size_t)bpv * 57 * sizeof(uint32_t) + sizeof(G1XYZZ)) : nvec * ((size_t)bpv + 1) * sizeof(G1XYZZ);
}

// slice width (log2 ticks of 100 MHz) of the fair-priority scheme of msm_fair_prio; 0 = off.
// Measured on the headline launch (1024 blobs, same box, interleaved runs, profiles/r03_prio_ab.txt): off 10.03 ms,
// 2^9..2^13 ticks no change, 2^15 9.72-9.75 ms, 2^17 9.68 ms, 2^19 10.07 ms -> 2^16 ticks = 0.66 ms per slice.
// (k_msm_small showed no gain at any slice width and keeps the scheme off.)
static uint32_t msm_prio_bit() {
    static const uint32_t v = (uint32_t)ab_knob("CKZG_HIP_MSM_PRIO_BIT", 16);
    return v;
}
static uint32_t small_prio_bit() {
    static const uint32_t v = (uint32_t)ab_knob("CKZG_HIP_SMALL_PRIO_BIT", 0);
    return v;
}

// digits already in scratch; runs accumulate + finalize.  Scratch layout is owned by callers.
static int run_msm(DeviceCtx *ctx, const FixedBaseTable &t, uint8_t *d_out48, uint8_t *d_status,
                   const int16_t *d_digits, const uint32_t *d_bad, G1XYZZ *d_partials, size_t nvec,
                   uint32_t ppb) {
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    // A handful of vectors cut finely by pick_pairs_per_block_fixed (a one-blob call, the smallest coalesced batches):
    // raw partial sums, one fold + finalize launch.  Only there: a 1024-thread fold workgroup needs a compute unit to
    // itself (16 waves of 128 registers, 58 KB of LDS), and in a mid-size batch of the throughput regime -- where two
    // launches of concurrent callers overlap -- it would wait for the OTHER launch's accumulate workgroups to drain
    // (measured: 128 callers 80.6 -> 64.6 k commitments/s when every bpv > 8 launch took this form).
    if (bpv > 8 && bpv <= 512 && ppb <= 1024 && nvec * bpv <= 512) {
        uint32_t *d_raw = reinterpret_cast<uint32_t *>(d_partials);
        hipLaunchKernelGGL((k_msm_accumulate<256, true>), dim3((unsigned)(nvec * bpv)), dim3(256), 0, ctx->stream,
                           d_partials, t.d_table, d_digits, pairs_per_vec, ppb, t.wbits - 1, bpv,
                           (uint32_t)t.npoints, (uint32_t)t.npoints, 1u, bpv, msm_prio_bit());
        HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
        hipLaunchKernelGGL(k_msm_fold_finalize, dim3((unsigned)nvec), dim3(FOLD_THREADS), 0, ctx->stream, d_out48, d_status, d_raw,
                           d_bad, bpv);
        HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
        HIP_TRY(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL((k_msm_accumulate<256, false>), dim3((unsigned)(nvec * bpv)), dim3(256), 0, ctx->stream,
                       d_partials, t.d_table, d_digits, pairs_per_vec, ppb, t.wbits - 1, bpv,
                       (uint32_t)t.npoints, (uint32_t)t.npoints, 1u, bpv, msm_prio_bit());
    HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    if (bpv > 8) {
        // mid-size batches split finely by the cost model: fold per vector on one wave, then dense one-lane finalize
        // (partials[nvec*bpv ..] is free: the callers size d_partials with partials_bytes())
        G1XYZZ *d_sums = d_partials + nvec * (size_t)bpv;
        hipLaunchKernelGGL(k_msm_reduce_partials, dim3((unsigned)nvec), dim3(64), 0, ctx->stream, d_sums,
                           d_partials, bpv);
        hipLaunchKernelGGL(k_msm_finalize, dim3((unsigned)((nvec + 63) / 64)), dim3(64), 0, ctx->stream,
                           d_out48, d_status, d_sums, d_bad, 1u, nvec);
    } else if (bpv >= 2 && nvec <= 4096) {
        // few vectors: fold the partials in a tree (latency); many: one lane per vector (dense inversions)
        if (bpv <= 2)
            hipLaunchKernelGGL(k_msm_finalize_tree<2>, dim3((unsigned)((nvec + 31) / 32)), dim3(64), 0, ctx->stream, d_out48,
                               d_status, d_partials, d_bad, bpv, nvec);
        else if (bpv <= 4)
            hipLaunchKernelGGL(k_msm_finalize_tree<4>, dim3((unsigned)((nvec + 15) / 16)), dim3(64), 0, ctx->stream, d_out48,
                               d_status, d_partials, d_bad, bpv, nvec);
        else
            hipLaunchKernelGGL(k_msm_finalize_tree<8>, dim3((unsigned)((nvec + 7) / 8)), dim3(64), 0, ctx->stream, d_out48,
                               d_status, d_partials, d_bad, bpv, nvec);
    } else {
        hipLaunchKernelGGL(k_msm_finalize, dim3((unsigned)((nvec + 63) / 64)), dim3(64), 0, ctx->stream,
                           d_out48, d_status, d_partials, d_bad, bpv, nvec);
    }
    HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void commit_collect_times(DeviceCtx *ctx) {
    float ms;
#ifdef CKZG_FOLD_TRACE
    {
        uint64_t h[16] = {};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fold_trace), sizeof h) == hipSuccess && getenv("CKZG_FOLD_TRACE_PRINT"))
            fprintf(stderr, "[fold trace, 10 ns ticks] level0 %llu  tree %llu  inv(extra) %llu  to_affine %llu  compress %llu  store %llu\n",
                    (unsigned long long)(h[1] - h[0]), (unsigned long long)(h[2] - h[1]), (unsigned long long)(h[3] - h[2]),
                    (unsigned long long)(h[4] - h[3]), (unsigned long long)(h[5] - h[4]), (unsigned long long)(h[6] - h[5]));
    }
#endif
    if (hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) ctx->last_ms[0] = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ms[1] = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]) == hipSuccess) ctx->last_ms[2] = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[4]) == hipSuccess) ctx->last_ms[3] = ms;
    (void)hipGetLastError();   // an event that cannot be read is no error of the call: do not leave it for the next hipGetLastError()
}

// Many small MSMs against sub-ranges of one table (FK20: 128 vectors of 64 points per blob).
// One 64-lane workgroup per vector; results stay in XYZZ form in d_out[nvec].
int msm_small_vectors_device(DeviceCtx *ctx, const FixedBaseTable &t, G1XYZZ *d_out,
                             const int16_t *d_digits, size_t nvec, uint32_t ppv,
                             uint32_t vecs_per_group) {
    if (nvec == 0) return 0;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * ppv;
    HIP_TRY(hipEventRecord(ctx->ev[5], ctx->stream));
    // few vectors: one wave each (latency); many vectors: 16 lanes each (throughput); a chip full several times
    // over: 8 lanes each, whose fold is one level shorter
    static const long forced_lpv = ab_knob("CKZG_HIP_SMALL_LPV", 0);
    const long lpv = forced_lpv ? forced_lpv : (nvec >= 65536 ? 8 : 16);
    // one wave per vector below this many vectors (measured, 8-bit table: 4096 vectors = 32 blobs take 2.4 ms with 16
    // lanes per vector -- 1024 waves, 128 additions per lane -- against two rounds of the one-wave form)
    static const size_t wave_max = (size_t)ab_knob("CKZG_HIP_SMALL_WAVE_MAX", 8192);
    if (nvec < wave_max) {
        hipLaunchKernelGGL(k_msm_small<64>, dim3((unsigned)nvec), dim3(64), 0, ctx->stream, d_out, t.d_table,
                           d_digits, (uint32_t)nvec, pairs_per_vec, t.wbits - 1, ppv, (uint32_t)t.npoints,
                           vecs_per_group, small_prio_bit());
    } else if (nvec >= 4096 && lpv == 4) {
        hipLaunchKernelGGL(k_msm_small<4>, dim3((unsigned)((nvec + 15) / 16)), dim3(64), 0, ctx->stream, d_out,
                           t.d_table, d_digits, (uint32_t)nvec, pairs_per_vec, t.wbits - 1, ppv,
                           (uint32_t)t.npoints, vecs_per_group, small_prio_bit());
    } else if (nvec >= 4096 && lpv == 8) {
        hipLaunchKernelGGL(k_msm_small<8>, dim3((unsigned)((nvec + 7) / 8)), dim3(64), 0, ctx->stream, d_out,
                           t.d_table, d_digits, (uint32_t)nvec, pairs_per_vec, t.wbits - 1, ppv,
                           (uint32_t)t.npoints, vecs_per_group, small_prio_bit());
    } else if (nvec >= 4096) {
        hipLaunchKernelGGL(k_msm_small<16>, dim3((unsigned)((nvec + 3) / 4)), dim3(64), 0, ctx->stream, d_out,
                           t.d_table, d_digits, (uint32_t)nvec, pairs_per_vec, t.wbits - 1, ppv,
                           (uint32_t)t.npoints, vecs_per_group, small_prio_bit());
    } else {
        hipLaunchKernelGGL(k_msm_small<64>, dim3((unsigned)nvec), dim3(64), 0, ctx->stream, d_out, t.d_table,
                           d_digits, (uint32_t)nvec, pairs_per_vec, t.wbits - 1, ppv, (uint32_t)t.npoints,
                           vecs_per_group, small_prio_bit());
    }
    HIP_TRY(hipEventRecord(ctx->ev[6], ctx->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

// Full-size MSMs (ppv == npoints) for nvec digit vectors already in HBM; writes nvec compressed
// points.  d_partials must hold msm_partials_needed() XYZZ points.
size_t msm_partials_needed(const FixedBaseTable &t, size_t nvec) {
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block_fixed(nvec, pairs_per_vec);
    return (partials_bytes(nvec, (pairs_per_vec + ppb - 1) / ppb) + sizeof(G1XYZZ) - 1) / sizeof(G1XYZZ);
}

int msm_from_digits_device(DeviceCtx *ctx, const FixedBaseTable &t, uint8_t *d_out48,
                           const int16_t *d_digits, G1XYZZ *d_partials, size_t nvec) {
    if (nvec == 0) return 0;
    if (!t.d_table) return 2;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block_fixed(nvec, pairs_per_vec);
    return run_msm(ctx, t, d_out48, nullptr, d_digits, nullptr, d_partials, nvec, ppb);
}

// Enqueue digits + MSM + finalize for n blobs on the context stream without waiting.  The scratch
// arena must already hold commit_scratch_bytes(n); successive enqueues may share it because the
// stream executes them in order.
size_t commit_scratch_bytes(const DeviceCtx *ctx, size_t n) {
    const FixedBaseTable &t = ctx->commit;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block_fixed(n, pairs_per_vec);
    uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    return align_up(n * (size_t)pairs_per_vec * sizeof(int16_t), 256) + align_up(n * sizeof(uint32_t), 256) +
           align_up(partials_bytes(n, bpv), 256);
}

int commit_blobs_enqueue(DeviceCtx *ctx, uint8_t *d_out48, uint8_t *d_status, const uint8_t *d_blobs,
                         size_t n) {
    if (n == 0) return 0;
    const FixedBaseTable &t = ctx->commit;
    if (!t.d_table) return 2;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block_fixed(n, pairs_per_vec);
    uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    size_t dig_bytes = align_up(n * (size_t)pairs_per_vec * sizeof(int16_t), 256);
    size_t bad_bytes = align_up(n * sizeof(uint32_t), 256);
    if (ctx->scratch.cap < dig_bytes + bad_bytes + align_up(partials_bytes(n, bpv), 256)) return 2;
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch.ptr);
    int16_t *d_digits = reinterpret_cast<int16_t *>(base);
    uint32_t *d_bad = reinterpret_cast<uint32_t *>(base + dig_bytes);
    G1XYZZ *d_partials = reinterpret_cast<G1XYZZ *>(base + dig_bytes + bad_bytes);
    HIP_TRY(hipMemsetAsync(d_bad, 0, n * sizeof(uint32_t), ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    size_t total = n * N_BLOB;
    hipLaunchKernelGGL(k_blob_digits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_digits, d_bad, d_blobs, total, t.wbits, t.twin);
    return run_msm(ctx, t, d_out48, d_status, d_digits, d_bad, d_partials, n, ppb);
}

// The lone one-blob commitment (blob_to_kzg_commitment, eip4844.c:264-280) as an EXPLICITLY built graph: copy in,
// flag reset, recoding, accumulate (raw partial sums), fold + finalize, copy out -- the six dependent nodes of
// commit_blobs_enqueue(n = 1) added one by one (hipGraphAddMemcpyNode1D / MemsetNode / KernelNode) with the same
// launch geometry and arguments.  No stream capture: a capture is invalidated (and the capturing thread faults inside
// libamdhip64 on this runtime) by HIP calls other threads make meanwhile -- round 4 therefore captured only while the
// calling thread was alone in the library and could not protect against HIP users outside it (RCCL's watchdog,
// PyTorch).  Node-by-node construction touches no stream and no global capture state, so it needs no quiet section
// and serves busy processes too.  Returns 0 and the instantiated graph, 4 if this table geometry does not take the
// raw-partials form (the caller then keeps plain launches), 2 on a HIP error.
int commit_one_graph_build(DeviceCtx *ctx, hipGraphExec_t *exec_out, uint8_t *d_out48, uint8_t *d_status,
                           const uint8_t *d_blob, const void *h_in, void *h_res) {
    const FixedBaseTable &t = ctx->commit;
    if (!t.d_table) return 2;
    const size_t n = 1;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block_fixed(n, pairs_per_vec);
    uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    if (!(bpv > 8 && bpv <= 512 && ppb <= 1024 && n * bpv <= 512)) return 4;   // run_msm's raw-partials condition
    size_t dig_bytes = align_up(n * (size_t)pairs_per_vec * sizeof(int16_t), 256);
    size_t bad_bytes = align_up(n * sizeof(uint32_t), 256);
    if (ctx->scratch.cap < dig_bytes + bad_bytes + align_up(partials_bytes(n, bpv), 256)) return 2;
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch.ptr);
    int16_t *d_digits = reinterpret_cast<int16_t *>(base);
    uint32_t *d_bad = reinterpret_cast<uint32_t *>(base + dig_bytes);
    G1XYZZ *d_partials = reinterpret_cast<G1XYZZ *>(base + dig_bytes + bad_bytes);
    const uint32_t *d_raw = reinterpret_cast<const uint32_t *>(d_partials);

    struct GraphOwner {   // destroyed on every exit path; the instantiated executable is independent of it
        hipGraph_t g = nullptr;
        ~GraphOwner() {
            if (g) (void)hipGraphDestroy(g);
        }
    } own;
    HIP_TRY(hipGraphCreate(&own.g, 0));
    hipGraphNode_t n_in, n_zero, n_digits, n_acc, n_fold, n_out;
    HIP_TRY(hipGraphAddMemcpyNode1D(&n_in, own.g, nullptr, 0, const_cast<uint8_t *>(d_blob), h_in, (size_t)N_BLOB * 32,
                                    hipMemcpyHostToDevice));
    hipMemsetParams zp;
    memset(&zp, 0, sizeof zp);
    zp.dst = d_bad;
    zp.elementSize = 4;
    zp.width = n;
    zp.height = 1;
    zp.pitch = n * 4;
    zp.value = 0;
    HIP_TRY(hipGraphAddMemsetNode(&n_zero, own.g, nullptr, 0, &zp));

    hipKernelNodeParams kp;
    // k_blob_digits(digits, bad, blobs, total, wbits, twin)
    size_t total = n * N_BLOB;
    int wbits = t.wbits, twin = t.twin;
    const uint8_t *blob_arg = d_blob;
    void *a_digits[] = {&d_digits, &d_bad, &blob_arg, &total, &wbits, &twin};
    memset(&kp, 0, sizeof kp);
    kp.func = reinterpret_cast<void *>(k_blob_digits);
    kp.gridDim = dim3((unsigned)((total + 255) / 256));
    kp.blockDim = dim3(256);
    kp.kernelParams = a_digits;
    hipGraphNode_t dep_digits[] = {n_in, n_zero};
    HIP_TRY(hipGraphAddKernelNode(&n_digits, own.g, dep_digits, 2, &kp));

    // k_msm_accumulate<256, true>(partials, table, digits, pairs_per_vec, pairs_per_block, half_shift, blocks_per_vec,
    //                             ppv, npoints, vecs_per_group, part_stride, prio_bit)
    const G1Affine *table = t.d_table;
    const int16_t *digits_arg = d_digits;
    int half_shift = t.wbits - 1;
    uint32_t npts = (uint32_t)t.npoints, one = 1u, prio = msm_prio_bit();
    void *a_acc[] = {&d_partials, &table, &digits_arg, &pairs_per_vec, &ppb, &half_shift, &bpv, &npts, &npts, &one, &bpv, &prio};
    memset(&kp, 0, sizeof kp);
    kp.func = reinterpret_cast<void *>(k_msm_accumulate<256, true>);
    kp.gridDim = dim3((unsigned)(n * bpv));
    kp.blockDim = dim3(256);
    kp.kernelParams = a_acc;
    HIP_TRY(hipGraphAddKernelNode(&n_acc, own.g, &n_digits, 1, &kp));

    // k_msm_fold_finalize(out48, status, raw, bad, bpv)
    const uint32_t *bad_arg = d_bad;
    void *a_fold[] = {&d_out48, &d_status, &d_raw, &bad_arg, &bpv};
    memset(&kp, 0, sizeof kp);
    kp.func = reinterpret_cast<void *>(k_msm_fold_finalize);
    kp.gridDim = dim3((unsigned)n);
    kp.blockDim = dim3(FOLD_THREADS);
    kp.kernelParams = a_fold;
    HIP_TRY(hipGraphAddKernelNode(&n_fold, own.g, &n_acc, 1, &kp));

    HIP_TRY(hipGraphAddMemcpyNode1D(&n_out, own.g, &n_fold, 1, h_res, d_out48, n * 49, hipMemcpyDeviceToHost));
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipGraphInstantiate(&exec, own.g, nullptr, nullptr, 0));
    *exec_out = exec;
    return 0;
}

// The host-pointer pipeline's form: only recoding + accumulation for a chunk of k blobs, partial sums left in
// d_part8[blob][8] (unused slots stay all-zero = infinity; the caller zeroed the array) and the per-blob flags in
// d_bad; ONE commit_finalize8_enqueue over the whole batch follows the last chunk.  A finalize per chunk would put
// a latency-bound launch (~0.2 ms, one lane per blob) between the accumulations of consecutive chunks.
// Returns 4 if this chunk size splits a blob into more than 8 partial sums (callers then use the plain form).
int commit_accumulate8_enqueue(DeviceCtx *ctx, G1XYZZ *d_part8, uint32_t *d_bad, const uint8_t *d_blobs, size_t k) {
    if (k == 0) return 0;
    const FixedBaseTable &t = ctx->commit;
    if (!t.d_table) return 2;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block(k, pairs_per_vec);
    uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    if (bpv > 8) return 4;
    size_t dig_bytes = align_up(k * (size_t)pairs_per_vec * sizeof(int16_t), 256);
    if (ctx->scratch.cap < dig_bytes) return 2;
    int16_t *d_digits = reinterpret_cast<int16_t *>(ctx->scratch.ptr);
    size_t total = k * N_BLOB;
    hipLaunchKernelGGL(k_blob_digits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_digits, d_bad, d_blobs, total, t.wbits, t.twin);
    hipLaunchKernelGGL((k_msm_accumulate<256, false>), dim3((unsigned)(k * bpv)), dim3(256), 0, ctx->stream,
                       d_part8, t.d_table, d_digits, pairs_per_vec, ppb, t.wbits - 1, bpv,
                       (uint32_t)t.npoints, (uint32_t)t.npoints, 1u, 8u, msm_prio_bit());
    HIP_TRY(hipGetLastError());
    return 0;
}

bool commit_chunk_fits8(const DeviceCtx *ctx, size_t k) {
    const FixedBaseTable &t = ctx->commit;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block(k, pairs_per_vec);
    return (pairs_per_vec + ppb - 1) / ppb <= 8;
}

int commit_finalize8_enqueue(DeviceCtx *ctx, uint8_t *d_out48, uint8_t *d_status, const G1XYZZ *d_part8,
                             const uint32_t *d_bad, size_t n) {
    if (n == 0) return 0;
    if (n <= 4096)
        hipLaunchKernelGGL(k_msm_finalize_tree<8>, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, ctx->stream, d_out48, d_status,
                           d_part8, d_bad, 8u, n);
    else
        hipLaunchKernelGGL(k_msm_finalize, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_out48, d_status,
                           d_part8, d_bad, 8u, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int commit_blobs_device(DeviceCtx *ctx, uint8_t *d_out48, uint8_t *d_status, const uint8_t *d_blobs,
                        size_t n) {
    if (n == 0) return 0;
    int rc = scratch_reserve(ctx, commit_scratch_bytes(ctx, n));
    if (rc) return rc;
#ifdef CKZG_MSM_TRACE
    const char *trace_file = getenv("CKZG_HIP_MSM_TRACE_FILE");
    uint64_t *d_trace = nullptr;
    size_t trace_waves = 0;
    if (trace_file && *trace_file) {
        const FixedBaseTable &t = ctx->commit;
        uint32_t ppv = (uint32_t)t.nwin * t.npoints, ppb = pick_pairs_per_block_fixed(n, ppv);
        trace_waves = n * ((ppv + ppb - 1) / ppb) * 4;
        HIP_TRY(hipMalloc(&d_trace, trace_waves * 32));
        HIP_TRY(hipMemset(d_trace, 0, trace_waves * 32));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_msm_trace), &d_trace, sizeof d_trace));
    }
#endif
    rc = commit_blobs_enqueue(ctx, d_out48, d_status, d_blobs, n);
    if (rc) return rc;
    HIP_TRY(dev::sync_stream(ctx->stream));
    commit_collect_times(ctx);
#ifdef CKZG_MSM_TRACE
    if (d_trace) {
        std::vector<uint64_t> h(trace_waves * 4);
        HIP_TRY(hipMemcpy(h.data(), d_trace, trace_waves * 32, hipMemcpyDeviceToHost));
        uint64_t *null_ptr = nullptr;
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_msm_trace), &null_ptr, sizeof null_ptr));
        (void)hipFree(d_trace);
        if (FILE *f = fopen(trace_file, "wb")) {   // the last traced launch wins
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
#endif
    return 0;
}

int msm_commit_table_raw_device(DeviceCtx *ctx, uint8_t *d_out48, const uint32_t *d_scalars, size_t n) {
    if (n == 0) return 0;
    const FixedBaseTable &t = ctx->commit;
    if (!t.d_table) return 2;
    uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    uint32_t ppb = pick_pairs_per_block_fixed(n, pairs_per_vec);
    uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    size_t dig_bytes = align_up(n * (size_t)pairs_per_vec * sizeof(int16_t), 256);
    size_t part_bytes = align_up(partials_bytes(n, bpv), 256);
    int rc = scratch_reserve(ctx, dig_bytes + part_bytes);
    if (rc) return rc;
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch.ptr);
    int16_t *d_digits = reinterpret_cast<int16_t *>(base);
    G1XYZZ *d_partials = reinterpret_cast<G1XYZZ *>(base + dig_bytes);
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    size_t total = n * N_BLOB;
    hipLaunchKernelGGL(k_raw_digits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_digits, d_scalars, total, t.wbits, t.twin);
    rc = run_msm(ctx, t, d_out48, nullptr, d_digits, nullptr, d_partials, n, ppb);
    if (rc) return rc;
    HIP_TRY(dev::sync_stream(ctx->stream));
    commit_collect_times(ctx);
    return 0;
}

// ------------------------------------------------------------------------------------------
// call-time tables: fixed-base sums over points that are only known when the call arrives
//
// The random-linear-combination sums of a verification batch (sum r^i proof_i, sum r^i z_i proof_i, sum r^i C_i;
// eip4844.c:731-746; the four sums of eip7594.c:926,530,807,758) are variable-base sums -- 128 sequential doublings per
// term on a ladder -- but their POINTS are known long before their scalars: the commitments and proofs are 96 bytes
// per blob and validated while the blobs (128 KB each) are still crossing PCIe, whereas r hashes every evaluation and
// arrives last.  So the doublings are done early: a narrow fixed-base table over the validated points (22 windows of
// 6 bits) is built on a side stream under the copy / the transcript hash, and once r exists the sums are digit vectors
// over that table: 0.45 ms instead of 1.0-2.45 ms after the last byte.  The table is in accumulator form (below): its
// build is ~1 ms of latency plus 0.15 us per point, short enough to pay from 8 blobs / 128 cells upwards
// (ckzg_api2.hip: verify_blobs_core, verify_cells_on decide; a table that does not fit the arena is simply not built).
// ------------------------------------------------------------------------------------------

// Same as k_raw_digits for vectors of any length: scalars[vec][i][8] (canonical, little-endian words), i < npoints
__global__ void k_raw_digits_n(int16_t *digits, const uint32_t *scalars, uint32_t npoints, size_t total, int wbits, int twin) {
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= total) return;
    size_t vec = gid / npoints;
    uint32_t i = (uint32_t)(gid - vec * npoints);
    uint32_t s[8];
    const uint4 *q = reinterpret_cast<const uint4 *>(scalars + gid * 8);
    uint4 a = q[0], b = q[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
    s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    glv_digits(digits + vec * (size_t)(2 * twin) * npoints + i, npoints, s, wbits, twin);
}

// ---- the call-time table is kept in ACCUMULATOR form ("X28": XYZZ28 -- x, y, zz, zzz as 14 limbs of 28 bits in the
// 2^392 domain, 224 bytes an entry; zz all-zero = infinity).  The setup tables are affine because they are built
// once and read for ever; a call-time table is built and read once, and what its affine form costs -- an inversion
// per entry, shared by Montgomery's trick or not: 5.3 of the 7.7 ms of an 8192-point 6-bit build -- buys a 9-product
// mixed addition instead of a 14-product full one in sums that are 0.1 ms of arithmetic either way.  In this form
// the build is the window-base ladders on quad lanes (0.75 ms) and one chain of full additions per (window, point).
struct X28Entry {
    uint4 q[14];
};

__device__ __forceinline__ void x28_store(X28Entry *dst, const XYZZ28 &p, bool inf) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < 14; k++)
        dst->q[k] = inf ? make_uint4(0, 0, 0, 0) : make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ XYZZ28 x28_load(const X28Entry *src, bool &inf) {
    XYZZ28 p;
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const uint4 v = src->q[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        if (k >= 7 && k < 11) any |= v.x | v.y | v.z | v.w;   // words 28..41 are zz (42, 43: the first limbs of zzz)
    }
    inf = any == 0;
    return p;
}
static_assert(sizeof(XYZZ28) == 224 && sizeof(X28Entry) == 224, "XYZZ28 is 4 x 14 words");

// wb[w][i] = 2^(wbits*w) * P_i in X28 form, four lanes per point (k_window_bases_quad without the way back to the
// 2^384 domain): after each window's doublings the lanes make z^2 | x | y | z^2 and then z^3, and lane k stores
// coordinate k.
__global__ __launch_bounds__(64) void k_window_bases_quad_x28(X28Entry *wb, const G1Affine *bases, int npoints, int wbits, int nwin) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 2, ql = gid & 3;
    if (i >= npoints) return;   // (whole quads leave together)
    const G1Affine b = bases[i];
    const bool inf = b.is_inf();
    JAC28 a;
    a.x = widen<1, 34>(f28_from_fp(b.x));
    a.y = widen<1, 34>(f28_from_fp(b.y));
    a.z = widen<2, 4>(f28_one());
    const int field = ql == 1 ? 0 : (ql == 2 ? 1 : (ql == 0 ? 2 : 3));   // x, y, zz, zzz
    for (int w = 0; w < nwin; w++) {
        {
            const auto x = widen<2, 34>(a.x), y = widen<2, 34>(a.y), z = widen<2, 34>(a.z), one = widen<2, 34>(f28_one());
            const auto p1 = mul(quad::qsel(ql, z, x, y, z), quad::qsel(ql, z, one, one, z));       // zz | x | y | zz
            const auto p2 = mul(p1, widen<2, 4>(a.z));                                             // -  | - | - | zzz
            uint32_t *dst = reinterpret_cast<uint32_t *>(wb + (size_t)w * npoints + i) + field * 14;
#pragma unroll
            for (int j = 0; j < 14; j++) dst[j] = inf ? 0u : (ql == 3 ? p2.l[j] : p1.l[j]);
        }
        if (w + 1 < nwin) {
            for (int k = 0; k < wbits; k++) quad::jac28_dbl_quad(a, ql);
        }
    }
}

// table[c][e] = (e + 1) * B_c, c = window * npoints + point: one lane per chain, `half` entries, full additions
__global__ __launch_bounds__(64) void k_table_chain_x28(X28Entry *table, const X28Entry *wb, uint32_t nchains, uint32_t half) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    bool binf;
    const XYZZ28 b = x28_load(wb + c, binf);
    X28Entry *out = table + (size_t)c * half;
    XYZZ28 acc = b;
    bool inf = binf;
    x28_store(out, acc, inf);
    for (uint32_t e = 1; e < half; e++) {
        xyzz28_add(acc, inf, b, binf);
        x28_store(out + e, acc, inf);
    }
}

// k_msm_accumulate over an X28 table: the same pair walk (k2 half first, phi when the boundary is crossed), full
// additions, plain loads -- the three or four vectors of a verification are a few additions per lane and the fold.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_msm_accumulate_x28(G1XYZZ *partials, const X28Entry *table, const int16_t *digits,
                                                               uint32_t pairs_per_vec, uint32_t pairs_per_block, int half_shift,
                                                               uint32_t blocks_per_vec, uint32_t npoints) {
    __shared__ uint32_t sh[57][THREADS];
    const uint32_t vec = blockIdx.x / blocks_per_vec, chunk = blockIdx.x % blocks_per_vec;
    const uint32_t q0 = chunk * pairs_per_block;
    const uint32_t q1 = q0 + pairs_per_block < pairs_per_vec ? q0 + pairs_per_block : pairs_per_vec;
    const uint32_t phi_pairs = pairs_per_vec >> 1, twin = phi_pairs / npoints;
    const int16_t *dg = digits + (size_t)vec * pairs_per_vec;
    XYZZ28 acc;
    bool inf = true;
    const uint32_t first = q0 + threadIdx.x;
    bool phi_pending = first < phi_pairs;
    for (uint32_t q = first; q < q1; q += THREADS) {
        if (phi_pending && q >= phi_pairs) {
            if (!inf) msm_apply_phi(acc);
            phi_pending = false;
        }
        const int d = dg[q];
        if (d == 0) continue;
        const uint32_t w = q / npoints, i = q - w * npoints;
        const uint32_t tw = w >= twin ? w - twin : w;
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        bool oinf;
        XYZZ28 o = x28_load(table + ((((size_t)tw * npoints + i) << half_shift) + (mag - 1)), oinf);
        if (oinf) continue;
        if (d < 0) o = xyzz28_neg(o);
        xyzz28_add(acc, inf, o, false);
    }
    if (phi_pending && !inf) msm_apply_phi(acc);
    quad::block_reduce_xyzz28_quad<THREADS>(acc, inf, sh);
    if (threadIdx.x == 0) partials[(size_t)vec * blocks_per_vec + chunk] = xyzz28_to_xyzz(acc, inf);
}

void call_table_geometry(FixedBaseTable *t, int npoints, int wbits) {
    t->d_table = nullptr;
    t->npoints = npoints;
    t->wbits = wbits;
    t->twin = FixedBaseTable::twin_for(wbits);
    t->nwin = 2 * t->twin;
    t->half = (size_t)1 << (wbits - 1);
}

size_t call_table_bytes(const FixedBaseTable &t) { return (size_t)t.twin * t.npoints * t.half * sizeof(X28Entry); }

// temporaries of call_table_enqueue: the window bases
size_t call_table_tmp_bytes(const FixedBaseTable &t) {
    return align_up((size_t)t.twin * t.npoints * sizeof(X28Entry), 256);
}

// Enqueue the construction of t (geometry set by call_table_geometry) into d_table (call_table_bytes(t)) on `stream`.
// d_bases: npoints affine points (2^384 domain, (0,0) = infinity) that must lie in the prime-order subgroup.
int call_table_enqueue(hipStream_t stream, FixedBaseTable *t, void *d_table, uint8_t *d_tmp, const G1Affine *d_bases) {
    const size_t nchains = (size_t)t->twin * t->npoints;
    X28Entry *d_wb = reinterpret_cast<X28Entry *>(d_tmp);
    hipLaunchKernelGGL(k_window_bases_quad_x28, dim3((unsigned)(((size_t)t->npoints * 4 + 63) / 64)), dim3(64), 0, stream, d_wb,
                       d_bases, t->npoints, t->wbits, t->twin);
    hipLaunchKernelGGL(k_table_chain_x28, dim3((unsigned)((nchains + 63) / 64)), dim3(64), 0, stream,
                       static_cast<X28Entry *>(d_table), d_wb, (uint32_t)nchains, (uint32_t)t->half);
    HIP_TRY(hipGetLastError());
    t->d_table = static_cast<G1Affine *>(d_table);   // (an X28 table: only table_sums_enqueue reads it)
    return 0;
}

size_t table_sums_scratch_bytes(const FixedBaseTable &t, size_t nvec) {
    const uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    const uint32_t ppb = pick_pairs_per_block(nvec, pairs_per_vec);
    const uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    return align_up(nvec * (size_t)pairs_per_vec * sizeof(int16_t), 256) + align_up(nvec * (size_t)bpv * sizeof(G1XYZZ), 256);
}

// d_sums[v] = sum_i scalars[v][i] * base_i over the whole (call-time, X28) table, v < nvec, as fully reduced XYZZ
// points.  Enqueue-only.
int table_sums_enqueue(hipStream_t stream, const FixedBaseTable &t, G1XYZZ *d_sums, const uint32_t *d_scalars, size_t nvec,
                       uint8_t *scratch) {
    if (!t.d_table || nvec == 0) return 2;
    const uint32_t pairs_per_vec = (uint32_t)t.nwin * t.npoints;
    const uint32_t ppb = pick_pairs_per_block(nvec, pairs_per_vec);
    const uint32_t bpv = (pairs_per_vec + ppb - 1) / ppb;
    int16_t *d_digits = reinterpret_cast<int16_t *>(scratch);
    G1XYZZ *d_partials = reinterpret_cast<G1XYZZ *>(scratch + align_up(nvec * (size_t)pairs_per_vec * sizeof(int16_t), 256));
    const size_t total = nvec * (size_t)t.npoints;
    hipLaunchKernelGGL(k_raw_digits_n, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_digits, d_scalars,
                       (uint32_t)t.npoints, total, t.wbits, t.twin);
    hipLaunchKernelGGL(k_msm_accumulate_x28<256>, dim3((unsigned)(nvec * bpv)), dim3(256), 0, stream, d_partials,
                       reinterpret_cast<const X28Entry *>(t.d_table), d_digits, pairs_per_vec, ppb, t.wbits - 1, bpv,
                       (uint32_t)t.npoints);
    hipLaunchKernelGGL(k_msm_reduce_partials, dim3((unsigned)nvec), dim3(64), 0, stream, d_sums, d_partials, bpv);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace dev
}  // namespace ckzg
