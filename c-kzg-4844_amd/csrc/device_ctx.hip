// device_ctx.hip -- creation / destruction of the per-KZGSettings GPU context: device selection,
// stream + timing events, upload of setup points and twiddles, fixed-base table construction.
// This is the GPU half of load_trusted_setup (src/setup/setup.c:392-505): the reference builds
// x_ext_fft_columns and (optionally) blst fixed-base tables on the CPU (setup.c:238-330); here the
// 64 G1 FFTs and all tables are produced by kernels and stay resident in HBM.
#include "api_common.hpp"

namespace ckzg {
namespace api {

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

// Largest window width <= wbits (but >= floor_bits) whose table, plus the construction scratch of
// build_fixed_base_table and a margin for per-call scratch, fits the HBM that is free right now.
// A wide table is an optimisation, never a reason for load_trusted_setup to fail.
static int fit_wbits(const char *what, int wbits, int floor_bits, int npoints) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return wbits;
    const size_t margin = (size_t)6 << 30;
    const int asked = wbits;
    while (wbits > floor_bits) {
        size_t twin = dev::FixedBaseTable::twin_for(wbits), half = (size_t)1 << (wbits - 1);
        // table + window bases + construction scratch (bounded at ~2 GiB, msm.hip) + per-call scratch
        size_t need = twin * npoints * half * sizeof(G1Affine) + twin * npoints * sizeof(G1XYZZ) + ((size_t)3 << 30) + margin;
        if (need <= free_b) break;
        wbits--;
    }
    if (wbits != asked)
        fprintf(stderr, "[ckzg-hip] %s table: window %d bits does not fit %.1f GB of free HBM, using %d bits\n",
                what, asked, free_b / 1e9, wbits);
    return wbits;
}

void destroy_device_ctx(dev::DeviceCtx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->commit.d_table) (void)hipFree(ctx->commit.d_table);
    if (ctx->fk20.d_table) (void)hipFree(ctx->fk20.d_table);
    if (ctx->mono.d_table) (void)hipFree(ctx->mono.d_table);
    if (ctx->d_xext) (void)hipFree(ctx->d_xext);
    if (ctx->d_roots) (void)hipFree(ctx->d_roots);
    if (ctx->d_brp_roots) (void)hipFree(ctx->d_brp_roots);
    if (ctx->d_roots_raw) (void)hipFree(ctx->d_roots_raw);
    if (ctx->d_mono) (void)hipFree(ctx->d_mono);
    if (ctx->d_shift) (void)hipFree(ctx->d_shift);
    if (ctx->d_unshift) (void)hipFree(ctx->d_unshift);
    if (ctx->scratch.ptr) (void)hipFree(ctx->scratch.ptr);
    ctx->api_arena.release();
    ctx->lc_arena.release();
    for (auto &e : ctx->ev) {
        if (e) (void)hipEventDestroy(e);
    }
    for (auto &e : ctx->stage_ev) {
        if (e) (void)hipEventDestroy(e);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (auto &h : ctx->h_stage) {
        if (h) (void)hipHostFree(h);
    }
    delete static_cast<PreparedG2 *>(ctx->host_prepared);
    delete ctx;
}

#define CTX_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            fprintf(stderr, "[ckzg-hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e),   \
                    __FILE__, __LINE__);                                                         \
            destroy_device_ctx(ctx);                                                             \
            return _e == hipErrorOutOfMemory ? C_KZG_MALLOC : C_KZG_ERROR;                       \
        }                                                                                        \
    } while (0)

C_KZG_RET create_device_ctx(KZGSettings *s, const G1Affine *lagrange_brp_affine,
                            const G1Affine *monomial_affine) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "[ckzg-hip] no HIP device available: this build has no CPU fallback for the MSM/FFT hot path\n");
        return C_KZG_ERROR;
    }
    int device = g_opts.device;
    if (device < 0) device = env_int("CKZG_HIP_DEVICE", -1);
    if (device < 0) device = env_int("LOCAL_RANK", 0) % ndev;
    if (device >= ndev) {
        fprintf(stderr, "[ckzg-hip] device %d out of range (%d visible)\n", device, ndev);
        return C_KZG_ERROR;
    }
    dev::DeviceCtx *ctx = new dev::DeviceCtx();
    ctx->device = device;
    CTX_TRY(hipSetDevice(device));
    CTX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CTX_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (auto &e : ctx->ev) CTX_TRY(hipEventCreate(&e));

    // Fr twiddles
    CTX_TRY(hipMalloc(&ctx->d_roots, (dev::N_EXT + 1) * sizeof(Fr)));
    CTX_TRY(hipMalloc(&ctx->d_brp_roots, dev::N_EXT * sizeof(Fr)));
    CTX_TRY(hipMemcpy(ctx->d_roots, s->roots_of_unity, (dev::N_EXT + 1) * sizeof(Fr), hipMemcpyHostToDevice));
    CTX_TRY(hipMemcpy(ctx->d_brp_roots, s->brp_roots_of_unity, dev::N_EXT * sizeof(Fr), hipMemcpyHostToDevice));

    // coset shift factors for recovery: 7^i and 7^-i
    {
        std::vector<Fr> sh(dev::N_EXT), ush(dev::N_EXT);
        Fr seven, seven_inv;
        for (int i = 0; i < 8; i++) {
            seven.l[i] = FR_SEVEN_MONT[i];
            seven_inv.l[i] = FR_SEVEN_INV_MONT[i];
        }
        sh[0] = ush[0] = Fr::one();
        for (int i = 1; i < dev::N_EXT; i++) {
            sh[i] = mul(sh[i - 1], seven);
            ush[i] = mul(ush[i - 1], seven_inv);
        }
        CTX_TRY(hipMalloc(&ctx->d_shift, dev::N_EXT * sizeof(Fr)));
        CTX_TRY(hipMalloc(&ctx->d_unshift, dev::N_EXT * sizeof(Fr)));
        CTX_TRY(hipMemcpy(ctx->d_shift, sh.data(), dev::N_EXT * sizeof(Fr), hipMemcpyHostToDevice));
        CTX_TRY(hipMemcpy(ctx->d_unshift, ush.data(), dev::N_EXT * sizeof(Fr), hipMemcpyHostToDevice));
    }

    // commitment table over the bit-reversed Lagrange points
    {
        int wbits = env_int("CKZG_HIP_COMMIT_WBITS", g_opts.commit_wbits);
        if (wbits < 4 || wbits > 16) wbits = 10;
        wbits = fit_wbits("commit", wbits, 10, (int)NUM_G1_POINTS);
        DeviceBuffer d_bases;
        if (!d_bases.alloc(NUM_G1_POINTS * sizeof(G1Affine))) {
            destroy_device_ctx(ctx);
            return C_KZG_MALLOC;
        }
        CTX_TRY(hipMemcpy(d_bases.p, lagrange_brp_affine, NUM_G1_POINTS * sizeof(G1Affine), hipMemcpyHostToDevice));
        int rc = dev::build_fixed_base_table(ctx, &ctx->commit, (const G1Affine *)d_bases.p, (int)NUM_G1_POINTS, wbits);
        if (rc) {
            destroy_device_ctx(ctx);
            return (C_KZG_RET)rc;
        }
    }
    // FK20: x_ext_fft columns by 64 G1 FFTs on the GPU, mirrored into the host struct, then the
    // fixed-base table over those 8192 points
    {
        CTX_TRY(hipMalloc(&ctx->d_mono, NUM_G1_POINTS * sizeof(G1Affine)));
        CTX_TRY(hipMemcpy(ctx->d_mono, monomial_affine, NUM_G1_POINTS * sizeof(G1Affine), hipMemcpyHostToDevice));
        std::vector<G1Affine> h_xext((size_t)dev::N_CELLS_EXT * dev::N_CELL);
        int rc = dev::fk20_setup_device(ctx, ctx->d_mono, h_xext.data());
        if (rc) {
            destroy_device_ctx(ctx);
            return (C_KZG_RET)rc;
        }
        s->x_ext_fft_columns = (g1_t **)calloc(dev::N_CELLS_EXT, sizeof(g1_t *));
        if (!s->x_ext_fft_columns) {
            destroy_device_ctx(ctx);
            return C_KZG_MALLOC;
        }
        for (int j = 0; j < dev::N_CELLS_EXT; j++) {
            s->x_ext_fft_columns[j] = (g1_t *)calloc(dev::N_CELL, sizeof(g1_t));
            if (!s->x_ext_fft_columns[j]) {
                destroy_device_ctx(ctx);
                return C_KZG_MALLOC;
            }
            for (int i = 0; i < dev::N_CELL; i++) {
                *as_g1(&s->x_ext_fft_columns[j][i]) = jac_from_affine(h_xext[(size_t)j * dev::N_CELL + i]);
            }
        }
        int wbits = env_int("CKZG_HIP_FK20_WBITS", g_opts.fk20_wbits);
        if (wbits == 0) wbits = s->wbits > 8 ? (s->wbits > 13 ? 13 : (int)s->wbits) : 8;
        if (wbits < 4 || wbits > 15) wbits = 8;
        wbits = fit_wbits("fk20", wbits, 8, dev::N_CELLS_EXT * dev::N_CELL);
        rc = dev::build_fixed_base_table(ctx, &ctx->fk20, ctx->d_xext, dev::N_CELLS_EXT * dev::N_CELL, wbits);
        if (rc) {
            destroy_device_ctx(ctx);
            return (C_KZG_RET)rc;
        }
    }
    // table over the monomial points for the low-latency (direct) cell-proof path
    {
        int wbits = env_int("CKZG_HIP_PROOF_WBITS", g_opts.proof_wbits);
        ctx->direct_max = env_int("CKZG_HIP_DIRECT_MAX", g_opts.direct_max);
        if (wbits != 0 && ctx->direct_max != 0) {
            if (wbits < 4 || wbits > 16) wbits = 8;
            wbits = fit_wbits("proof", wbits, 8, (int)NUM_G1_POINTS);
            int rc = dev::build_fixed_base_table(ctx, &ctx->mono, ctx->d_mono, (int)NUM_G1_POINTS, wbits);
            if (rc) {
                destroy_device_ctx(ctx);
                return (C_KZG_RET)rc;
            }
            // Automatic hand-over point (measured, tools/bench_direct_vs_fk20.py): FK20 costs ~28 ms for any
            // batch of up to ~48 blobs (13 dependent ladder launches), the direct path 3.5 ms for one blob
            // plus 2.5 / 1.9 / 1.5 ms per further blob with an 8 / 13 / 16-bit table.
            if (ctx->direct_max < 0) ctx->direct_max = wbits >= 15 ? 18 : (wbits >= 11 ? 14 : 10);
        }
        if (ctx->direct_max < 0) ctx->direct_max = 0;
    }
    {
        PreparedG2 *pg = new PreparedG2();
        host::g2_prepare(pg->gen, host::g2_to_affine(host::g2_generator()));
        host::g2_prepare(pg->s1, host::g2_to_affine(*as_g2(&s->g2_values_monomial[1])));
        host::g2_prepare(pg->s64, host::g2_to_affine(*as_g2(&s->g2_values_monomial[dev::N_CELL])));
        ctx->host_prepared = pg;
    }
    header_of(s)->ctx = ctx;
    return C_KZG_OK;
}

}  // namespace api
}  // namespace ckzg
