// ckzg_api.hip -- the extern "C" boundary: ckzg.h (reference ABI) + ckzg_hip.h (batch/device
// extensions).  Host-side protocol logic lives here; every MSM / NTT goes to the HIP kernels via
// device.hpp.  There is no CPU fallback for the hot path: if no GPU context is attached to the
// KZGSettings (or the HIP runtime fails) the call returns C_KZG_ERROR and says why on stderr.
#include <atomic>
#include <thread>

#include "api_common.hpp"

using namespace ckzg;
using namespace ckzg::host;
using namespace ckzg::api;

// ------------------------------------------------------------------------------------------
// options
// ------------------------------------------------------------------------------------------

namespace ckzg {
namespace api {
Options g_opts;
}
}  // namespace ckzg

extern "C" C_KZG_RET ckzg_hip_set_option(const char *key, int64_t value) {
    if (!key) return C_KZG_BADARGS;
    if (!strcmp(key, "device")) {
        if (value < -1 || value > 1023) return C_KZG_BADARGS;
        g_opts.device = (int)value;
    } else if (!strcmp(key, "commit_wbits")) {
        if (value < 4 || value > 16) return C_KZG_BADARGS;
        g_opts.commit_wbits = (int)value;
    } else if (!strcmp(key, "fk20_wbits")) {
        if (value != 0 && (value < 4 || value > 15)) return C_KZG_BADARGS;
        g_opts.fk20_wbits = (int)value;
    } else if (!strcmp(key, "proof_wbits")) {
        if (value != 0 && (value < 4 || value > 16)) return C_KZG_BADARGS;
        g_opts.proof_wbits = (int)value;
    } else if (!strcmp(key, "gpu_sha_min")) {
        if (value < 0 || value > (1 << 30)) return C_KZG_BADARGS;
        g_opts.gpu_sha_min = (int)value;  // read at call time, unlike the table options
    } else if (!strcmp(key, "direct_max")) {
        if (value < -1 || value > 4096) return C_KZG_BADARGS;
        g_opts.direct_max = (int)value;
    } else {
        return C_KZG_BADARGS;
    }
    return C_KZG_OK;
}

extern "C" int ckzg_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// trusted setup (src/setup/setup.c)
// ------------------------------------------------------------------------------------------

static const uint32_t ROOT_8192[8] = {FR_ROOT_8192_MONT[0], FR_ROOT_8192_MONT[1], FR_ROOT_8192_MONT[2],
                                      FR_ROOT_8192_MONT[3], FR_ROOT_8192_MONT[4], FR_ROOT_8192_MONT[5],
                                      FR_ROOT_8192_MONT[6], FR_ROOT_8192_MONT[7]};

static C_KZG_RET compute_roots_of_unity(KZGSettings *s) {  // setup.c:99-153
    Fr *roots = as_fr(s->roots_of_unity);
    Fr root;
    for (int i = 0; i < 8; i++) root.l[i] = ROOT_8192[i];
    roots[0] = Fr::one();
    roots[1] = root;
    size_t i;
    for (i = 2; i <= FIELD_ELEMENTS_PER_EXT_BLOB; i++) {
        roots[i] = mul(roots[i - 1], root);
        if (roots[i] == Fr::one()) break;
    }
    if (i != FIELD_ELEMENTS_PER_EXT_BLOB || roots[FIELD_ELEMENTS_PER_EXT_BLOB] != Fr::one()) return C_KZG_BADARGS;
    memcpy(s->brp_roots_of_unity, s->roots_of_unity, sizeof(fr_t) * FIELD_ELEMENTS_PER_EXT_BLOB);
    bit_reversal_permutation(s->brp_roots_of_unity, sizeof(fr_t), FIELD_ELEMENTS_PER_EXT_BLOB);
    for (i = 0; i <= FIELD_ELEMENTS_PER_EXT_BLOB; i++) {
        s->reverse_roots_of_unity[i] = s->roots_of_unity[FIELD_ELEMENTS_PER_EXT_BLOB - i];
    }
    return C_KZG_OK;
}

extern "C" void free_trusted_setup(KZGSettings *s) {  // setup.c:162-190
    if (s == NULL) return;
    if (s->roots_of_unity) {
        SettingsHeader *h = header_of(s);
        if (h && h->ctx) destroy_device_ctx(h->ctx);
        free(h ? (void *)h : (void *)s->roots_of_unity);
    }
    free(s->brp_roots_of_unity);
    free(s->reverse_roots_of_unity);
    free(s->g1_values_monomial);
    free(s->g1_values_lagrange_brp);
    free(s->g2_values_monomial);
    if (s->x_ext_fft_columns) {
        for (size_t i = 0; i < CELLS_PER_EXT_BLOB; i++) free(s->x_ext_fft_columns[i]);
    }
    free(s->x_ext_fft_columns);
    memset(s, 0, sizeof *s);
}

extern "C" C_KZG_RET load_trusted_setup(KZGSettings *out, const uint8_t *g1_monomial_bytes,
                                        uint64_t num_g1_monomial_bytes,
                                        const uint8_t *g1_lagrange_bytes,
                                        uint64_t num_g1_lagrange_bytes,
                                        const uint8_t *g2_monomial_bytes,
                                        uint64_t num_g2_monomial_bytes, uint64_t precompute) {
    // setup.c:392-505
    C_KZG_RET ret = C_KZG_OK;
    std::vector<G1Affine> lagr_affine(NUM_G1_POINTS), mono_affine(NUM_G1_POINTS);
    memset(out, 0, sizeof *out);
    if (precompute > 15) return C_KZG_BADARGS;
    out->wbits = precompute;
    if (num_g1_monomial_bytes != NUM_G1_POINTS * 48 || num_g1_lagrange_bytes != NUM_G1_POINTS * 48 ||
        num_g2_monomial_bytes != NUM_G2_POINTS * 96) {
        return C_KZG_BADARGS;
    }
    // roots_of_unity carries the hidden header that links this struct to its GPU context
    {
        size_t bytes = sizeof(SettingsHeader) + (FIELD_ELEMENTS_PER_EXT_BLOB + 1) * sizeof(fr_t);
        SettingsHeader *h = (SettingsHeader *)calloc(1, bytes);
        if (!h) return C_KZG_MALLOC;
        h->magic = SETTINGS_MAGIC;
        h->ctx = nullptr;
        out->roots_of_unity = (fr_t *)(h + 1);
    }
    out->brp_roots_of_unity = (fr_t *)calloc(FIELD_ELEMENTS_PER_EXT_BLOB, sizeof(fr_t));
    out->reverse_roots_of_unity = (fr_t *)calloc(FIELD_ELEMENTS_PER_EXT_BLOB + 1, sizeof(fr_t));
    out->g1_values_monomial = (g1_t *)calloc(NUM_G1_POINTS, sizeof(g1_t));
    out->g1_values_lagrange_brp = (g1_t *)calloc(NUM_G1_POINTS, sizeof(g1_t));
    out->g2_values_monomial = (g2_t *)calloc(NUM_G2_POINTS, sizeof(g2_t));
    if (!out->brp_roots_of_unity || !out->reverse_roots_of_unity || !out->g1_values_monomial ||
        !out->g1_values_lagrange_brp || !out->g2_values_monomial) {
        ret = C_KZG_MALLOC;
        goto fail;
    }
    // The file is trusted: curve membership only, no subgroup check (setup.c:447-477)
    // (8192 square roots: ~0.25 s on one core, spread over up to 16 threads)
    {
        unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = hw >= 16 ? 16 : (hw ? hw : 1);
        std::atomic<int> bad(0);
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++) {
            th.emplace_back([&, t]() {
                for (size_t i = t; i < NUM_G1_POINTS; i += nt) {
                    if (g1_uncompress(mono_affine[i], g1_monomial_bytes + 48 * i) != 0 ||
                        g1_uncompress(lagr_affine[i], g1_lagrange_bytes + 48 * i) != 0) {
                        bad.store(1);
                        return;
                    }
                    *as_g1(&out->g1_values_monomial[i]) = jac_from_affine(mono_affine[i]);
                    *as_g1(&out->g1_values_lagrange_brp[i]) = jac_from_affine(lagr_affine[i]);
                }
            });
        }
        for (auto &x : th) x.join();
        if (bad.load()) {
            ret = C_KZG_BADARGS;
            goto fail;
        }
    }
    for (size_t i = 0; i < NUM_G2_POINTS; i++) {
        G2Affine a;
        if (g2_uncompress(a, g2_monomial_bytes + 96 * i) != 0) {
            ret = C_KZG_BADARGS;
            goto fail;
        }
        G2Jac j = a.is_inf() ? G2Jac::inf() : G2Jac{a.x, a.y, Fp2::one()};
        memcpy(&out->g2_values_monomial[i], &j, sizeof j);
    }
    // setup.c:339-358: reject a setup whose "Lagrange" points are really the monomial ones
    if (pairings_verify(*as_g1(&out->g1_values_lagrange_brp[1]), *as_g2(&out->g2_values_monomial[0]),
                        *as_g1(&out->g1_values_lagrange_brp[0]), *as_g2(&out->g2_values_monomial[1]))) {
        ret = C_KZG_BADARGS;
        goto fail;
    }
    ret = compute_roots_of_unity(out);
    if (ret != C_KZG_OK) goto fail;
    bit_reversal_permutation(out->g1_values_lagrange_brp, sizeof(g1_t), NUM_G1_POINTS);
    bit_reversal_permutation(lagr_affine.data(), sizeof(G1Affine), NUM_G1_POINTS);
    // GPU context: commitment tables, NTT twiddles, FK20 columns and tables (setup.c:238-330)
    ret = create_device_ctx(out, lagr_affine.data(), mono_affine.data());
    if (ret != C_KZG_OK) goto fail;
    return C_KZG_OK;
fail:
    free_trusted_setup(out);
    return ret;
}

extern "C" C_KZG_RET load_trusted_setup_file(KZGSettings *out, FILE *in, uint64_t precompute) {
    // setup.c:519-600: "<n_g1> <n_g2>" then hex of: G1 Lagrange, G2 monomial, G1 monomial
    uint64_t n1 = 0, n2 = 0;
    memset(out, 0, sizeof *out);
    std::vector<uint8_t> mono(NUM_G1_POINTS * 48), lagr(NUM_G1_POINTS * 48), g2(NUM_G2_POINTS * 96);
    if (fscanf(in, "%" SCNu64, &n1) != 1 || n1 != NUM_G1_POINTS) return C_KZG_BADARGS;
    if (fscanf(in, "%" SCNu64, &n2) != 1 || n2 != NUM_G2_POINTS) return C_KZG_BADARGS;
    for (auto &b : lagr) {
        if (fscanf(in, "%2hhx", &b) != 1) return C_KZG_BADARGS;
    }
    for (auto &b : g2) {
        if (fscanf(in, "%2hhx", &b) != 1) return C_KZG_BADARGS;
    }
    for (auto &b : mono) {
        if (fscanf(in, "%2hhx", &b) != 1) return C_KZG_BADARGS;
    }
    return load_trusted_setup(out, mono.data(), mono.size(), lagr.data(), lagr.size(), g2.data(),
                              g2.size(), precompute);
}

// ------------------------------------------------------------------------------------------
// byte <-> element helpers (src/common/bytes.c)
// ------------------------------------------------------------------------------------------

extern "C" C_KZG_RET bytes_to_bls_field(fr_t *out, const Bytes32 *b) {
    return fr_from_bytes_canonical(*as_fr(out), b->bytes) ? C_KZG_OK : C_KZG_BADARGS;
}

extern "C" void bytes_from_bls_field(Bytes32 *out, const fr_t *in) { fr_to_bytes(out->bytes, *as_fr(in)); }

extern "C" void bytes_from_g1(Bytes48 *out, const g1_t *in) {
    g1_compress_affine(out->bytes, jac_to_affine(*as_g1(in)));
}

extern "C" C_KZG_RET bytes_to_kzg_commitment(g1_t *out, const Bytes48 *b) {
    return validate_kzg_g1(*as_g1(out), b->bytes);
}

extern "C" C_KZG_RET bytes_to_kzg_proof(g1_t *out, const Bytes48 *b) {
    return validate_kzg_g1(*as_g1(out), b->bytes);
}

// ------------------------------------------------------------------------------------------
// blob_to_kzg_commitment (src/eip4844/eip4844.c:264-280) and its batch forms
// ------------------------------------------------------------------------------------------

// pageable -> pinned staging copy.  One core moves ~10 GB/s, which would make the copy (13 ms per
// 1024 blobs) longer than the kernels it is supposed to hide behind (10.4 ms): large chunks are split
// over four threads.
static void staged_copy(void *dst, const void *src, size_t bytes) {
    const size_t nt = 4;
    if (bytes < ((size_t)4 << 20) || std::thread::hardware_concurrency() < 8) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t part = (bytes / nt + 4095) & ~(size_t)4095;
    std::thread th[nt - 1];
    for (size_t t = 1; t < nt; t++) {
        size_t o = t * part, len = o >= bytes ? 0 : (bytes - o < part ? bytes - o : part);
        th[t - 1] = std::thread([=]() {
            if (len) memcpy((uint8_t *)dst + o, (const uint8_t *)src + o, len);
        });
    }
    memcpy(dst, src, part < bytes ? part : bytes);
    for (auto &x : th) x.join();
}

extern "C" C_KZG_RET ckzg_hip_blob_to_kzg_commitment_batch_device(void *d_out48, void *d_status,
                                                                  const void *d_blobs, uint64_t n,
                                                                  const KZGSettings *s) {
    dev::DeviceCtx *ctx = ctx_of(s);
    if (!ctx) return C_KZG_ERROR;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) return C_KZG_ERROR;
    return (C_KZG_RET)dev::commit_blobs_device(ctx, (uint8_t *)d_out48, (uint8_t *)d_status,
                                               (const uint8_t *)d_blobs, n);
}

extern "C" C_KZG_RET ckzg_hip_blob_to_kzg_commitment_batch(KZGCommitment *out, uint8_t *status,
                                                           const Blob *blobs, uint64_t n,
                                                           const KZGSettings *s) {
    dev::DeviceCtx *ctx = ctx_of(s);
    if (!ctx) return C_KZG_ERROR;
    if (n == 0) return C_KZG_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) return C_KZG_ERROR;
    // Host buffers are pageable, and a pageable hipMemcpyAsync serialises with everything.  So chunk
    // i+1 is memcpy'd by this thread into a pinned staging buffer and DMA'd on the copy stream while
    // the kernels of chunk i execute on the compute stream.  Two staging/device buffers, events
    // both ways.  Chunks grow geometrically (64, 192, 512, 512, ...): the first one is small so that
    // the GPU starts after ~0.5 ms, and since a chunk computes ~3x longer than the next one takes to
    // stage, the following ones can be large enough to run the kernels at full-batch efficiency.
    static const uint64_t CH = []() {
        const char *v = getenv("CKZG_HIP_COMMIT_CHUNK");
        long c = v && *v ? atol(v) : 512;
        return (uint64_t)(c < 16 ? 16 : (c > 1024 ? 1024 : c));
    }();
    const uint64_t FIRST = CH < 64 ? CH : 64;
    const uint64_t m = n < CH ? n : CH;
    Trace tr("commit_batch");
    // device temporaries from the context's arena, events kept in the context: a single-blob call
    // must not pay for hipMalloc/hipFree/hipEventCreate
    Arena &ar = ctx->api_arena;
    if (!ar.begin(2 * m * BYTES_PER_BLOB + n * 49)) return C_KZG_MALLOC;
    ArenaTrim trim(ar);
    ABuf<uint8_t> d_blobs[2] = {ABuf<uint8_t>(ar, m * BYTES_PER_BLOB), ABuf<uint8_t>(ar, n > FIRST ? m * BYTES_PER_BLOB : 1)};
    ABuf<uint8_t> d_out(ar, n * 48), d_status(ar, n);
    if (!d_blobs[0].p || !d_blobs[1].p || !d_out.p || !d_status.p) return C_KZG_MALLOC;
    hipEvent_t *copied = ctx->stage_ev, *consumed = ctx->stage_ev + 2;
    C_KZG_RET ret = C_KZG_OK;
    std::vector<uint8_t> st(n);
    bool pending[2] = {false, false};
    if (dev::scratch_reserve(ctx, dev::commit_scratch_bytes(ctx, m)) != 0) return C_KZG_MALLOC;
    for (int i = 0; i < 2; i++) {
        if (!ctx->h_stage[i] && hipHostMalloc(&ctx->h_stage[i], CH * BYTES_PER_BLOB, hipHostMallocDefault) != hipSuccess) {
            ctx->h_stage[i] = nullptr;
            return C_KZG_MALLOC;
        }
    }
    for (int i = 0; i < 4; i++) {
        if (!ctx->stage_ev[i] && hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming) != hipSuccess) {
            ctx->stage_ev[i] = nullptr;
            ret = C_KZG_ERROR;
        }
    }
    tr.mark("buffers + events");
    uint64_t chunk = 0, k = 0, want = FIRST;
    for (uint64_t off = 0; off < n && ret == C_KZG_OK; off += k, chunk++) {
        const int b = (int)(chunk & 1);
        k = n - off < want ? n - off : want;
        want = 3 * want < CH ? 3 * want : CH;
        bool ok = true;
        if (pending[b]) {
            // the pinned buffer is free once its DMA finished; the device buffer once its kernels did
            ok = ok && hipEventSynchronize(copied[b]) == hipSuccess;
            ok = ok && hipStreamWaitEvent(ctx->copy_stream, consumed[b], 0) == hipSuccess;
        }
        staged_copy(ctx->h_stage[b], blobs + off, k * BYTES_PER_BLOB);
        ok = ok && hipMemcpyAsync(d_blobs[b].p, ctx->h_stage[b], k * BYTES_PER_BLOB, hipMemcpyHostToDevice,
                                  ctx->copy_stream) == hipSuccess;
        ok = ok && hipEventRecord(copied[b], ctx->copy_stream) == hipSuccess;
        ok = ok && hipStreamWaitEvent(ctx->stream, copied[b], 0) == hipSuccess;
        if (!ok) {
            ret = C_KZG_ERROR;
            break;
        }
        int rc = dev::commit_blobs_enqueue(ctx, (uint8_t *)d_out.p + off * 48, (uint8_t *)d_status.p + off,
                                           (const uint8_t *)d_blobs[b].p, k);
        if (rc) {
            ret = (C_KZG_RET)rc;
            break;
        }
        if (hipEventRecord(consumed[b], ctx->stream) != hipSuccess) ret = C_KZG_ERROR;
        pending[b] = true;
    }
    tr.mark("staging loop (copies + enqueues)");
    if (hipStreamSynchronize(ctx->copy_stream) != hipSuccess) ret = ret == C_KZG_OK ? C_KZG_ERROR : ret;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) ret = ret == C_KZG_OK ? C_KZG_ERROR : ret;
    tr.mark("wait for the GPU");
    if (ret != C_KZG_OK) return ret;
    if (hipMemcpy(out, d_out.p, n * 48, hipMemcpyDeviceToHost) != hipSuccess) return C_KZG_ERROR;
    if (hipMemcpy(st.data(), d_status.p, n, hipMemcpyDeviceToHost) != hipSuccess) return C_KZG_ERROR;
    for (uint64_t i = 0; i < n; i++) {
        if (status) status[i] = st[i];
        if (st[i]) ret = C_KZG_BADARGS;
    }
    return ret;
}

extern "C" C_KZG_RET blob_to_kzg_commitment(KZGCommitment *out, const Blob *blob, const KZGSettings *s) {
    uint8_t st = 0;
    return ckzg_hip_blob_to_kzg_commitment_batch(out, &st, blob, 1, s);
}

// ------------------------------------------------------------------------------------------
// compute_cells_and_kzg_proofs (src/eip7594/eip7594.c:61-157) and its batch forms
// ------------------------------------------------------------------------------------------

extern "C" C_KZG_RET ckzg_hip_compute_cells_and_kzg_proofs_batch_device(void *d_cells, void *d_proofs,
                                                                        void *d_status,
                                                                        const void *d_blobs, uint64_t n,
                                                                        const KZGSettings *s) {
    if (d_cells == NULL && d_proofs == NULL) return C_KZG_BADARGS;
    dev::DeviceCtx *ctx = ctx_of(s);
    if (!ctx) return C_KZG_ERROR;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) return C_KZG_ERROR;
    return (C_KZG_RET)dev::cells_and_proofs_device(ctx, (uint8_t *)d_cells, (uint8_t *)d_proofs,
                                                   (uint8_t *)d_status, (const uint8_t *)d_blobs, n);
}

extern "C" C_KZG_RET ckzg_hip_compute_cells_and_kzg_proofs_batch(Cell *cells, KZGProof *proofs,
                                                                 uint8_t *status, const Blob *blobs,
                                                                 uint64_t n, const KZGSettings *s) {
    if (cells == NULL && proofs == NULL) return C_KZG_BADARGS;  // eip7594.c:72-74
    dev::DeviceCtx *ctx = ctx_of(s);
    if (!ctx) return C_KZG_ERROR;
    if (n == 0) return C_KZG_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) return C_KZG_ERROR;
    const uint64_t CH = 2048;  // >= 2 waves of G1-FFT butterflies per SIMD per stage launch
    uint64_t m = n < CH ? n : CH;
    const size_t cells_per = (size_t)CELLS_PER_EXT_BLOB * BYTES_PER_CELL, proofs_per = (size_t)CELLS_PER_EXT_BLOB * 48;
    Arena &ar = ctx->api_arena;
    if (!ar.begin(m * (BYTES_PER_BLOB + 1 + (cells ? cells_per : 0) + (proofs ? proofs_per : 0)))) return C_KZG_MALLOC;
    ArenaTrim trim(ar);
    ABuf<uint8_t> d_blobs(ar, m * BYTES_PER_BLOB), d_status(ar, m);
    ABuf<uint8_t> d_cells(ar, cells ? m * cells_per : 1), d_proofs(ar, proofs ? m * proofs_per : 1);
    if (!d_blobs.p || !d_status.p || !d_cells.p || !d_proofs.p) return C_KZG_MALLOC;
    if (!cells) d_cells.p = nullptr;
    if (!proofs) d_proofs.p = nullptr;
    std::vector<uint8_t> st(m);
    C_KZG_RET ret = C_KZG_OK;
    for (uint64_t off = 0; off < n; off += CH) {
        uint64_t k = n - off < CH ? n - off : CH;
        if (hipMemcpyAsync(d_blobs.p, blobs + off, k * BYTES_PER_BLOB, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return C_KZG_ERROR;
        int rc = dev::cells_and_proofs_device(ctx, (uint8_t *)d_cells.p, (uint8_t *)d_proofs.p,
                                              (uint8_t *)d_status.p, (const uint8_t *)d_blobs.p, k);
        if (rc) return (C_KZG_RET)rc;
        if (cells && hipMemcpy(cells + off * CELLS_PER_EXT_BLOB, d_cells.p, k * cells_per, hipMemcpyDeviceToHost) != hipSuccess)
            return C_KZG_ERROR;
        if (proofs && hipMemcpy(proofs + off * CELLS_PER_EXT_BLOB, d_proofs.p, k * proofs_per, hipMemcpyDeviceToHost) != hipSuccess)
            return C_KZG_ERROR;
        if (hipMemcpy(st.data(), d_status.p, k, hipMemcpyDeviceToHost) != hipSuccess) return C_KZG_ERROR;
        for (uint64_t i = 0; i < k; i++) {
            if (status) status[off + i] = st[i];
            if (st[i]) ret = C_KZG_BADARGS;
        }
    }
    return ret;
}

extern "C" C_KZG_RET compute_cells_and_kzg_proofs(Cell *cells, KZGProof *proofs, const Blob *blob,
                                                  const KZGSettings *s) {
    uint8_t st = 0;
    return ckzg_hip_compute_cells_and_kzg_proofs_batch(cells, proofs, &st, blob, 1, s);
}

extern "C" double ckzg_hip_last_kernel_ms(const KZGSettings *s, int which) {
    dev::DeviceCtx *ctx = ctx_of(s);
    if (!ctx || which < 0 || which > 3) return -1.0;
    return ctx->last_ms[which];
}

extern "C" uint64_t ckzg_hip_table_bytes(const KZGSettings *s) {
    dev::DeviceCtx *ctx = ctx_of(s);
    if (!ctx) return 0;
    return ctx->commit.bytes() + ctx->fk20.bytes() + ctx->mono.bytes();
}

extern "C" int ckzg_hip_table_wbits(const KZGSettings *s, int which) {
    dev::DeviceCtx *ctx = ctx_of(s);
    if (!ctx) return -1;
    switch (which) {
        case 0: return ctx->commit.wbits;
        case 1: return ctx->fk20.wbits;
        case 2: return ctx->mono.d_table ? ctx->mono.wbits : 0;
        default: return -1;
    }
}
