// ckzg_api.hip -- the extern "C" boundary: ckzg.h (reference ABI) + ckzg_hip.h (batch/device
// extensions).  Host-side protocol logic lives here; every MSM / NTT goes to the HIP kernels via
// device.hpp.  There is no CPU fallback for the hot path: if no GPU context is attached to the
// KZGSettings (or the HIP runtime fails) the call returns C_KZG_ERROR and says why on stderr.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <thread>

#include "api_common.hpp"
#include "combiner.hpp"

using namespace ckzg;
using namespace ckzg::host;
using namespace ckzg::api;

// ------------------------------------------------------------------------------------------
// options
// ------------------------------------------------------------------------------------------

namespace ckzg {
namespace api {
static std::mutex g_opts_mu;
static Options g_opts;
std::atomic<int> g_gpu_sha_min{0};
std::atomic<int> g_host_threads{0};
std::atomic<int> g_verify_pipe_min{1024}, g_verify_call_table{1}, g_verify_cu_partition{1};
static std::atomic<int> g_commit_graph{1};   // option "commit_graph": a lone one-blob commitment goes out as one (explicitly built) graph
static std::atomic<uint64_t> g_graph_stats[3];   // graphs built, builds that failed (plain launches instead), graph launches
Options options_snapshot() {
    std::lock_guard<std::mutex> lock(g_opts_mu);
    return g_opts;
}
}  // namespace api
}  // namespace ckzg

// ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams
// that share a queue serialise.  Every slot of a pool owns two streams, so the concurrent-caller
// design needs more queues than the default to overlap: ask for 16 unless the user chose a value.
// Only effective when this library is loaded before the process's first HIP call; a host that
// initialises HIP earlier (e.g. imports torch first) sets the variable itself.
__attribute__((constructor)) static void ckzg_hip_queue_hint() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

extern "C" C_KZG_RET ckzg_hip_set_option(const char *key, int64_t value) {
    if (!key) return C_KZG_BADARGS;
    std::lock_guard<std::mutex> lock(g_opts_mu);
    if (!strcmp(key, "device")) {
        if (value < -1 || value > 1023) return C_KZG_BADARGS;
        g_opts.device = (int)value;
    } else if (!strcmp(key, "devices")) {
        if (value < -1) return C_KZG_BADARGS;
        g_opts.devices = value;
    } else if (!strcmp(key, "replicas")) {
        if (value < 1 || value > 8) return C_KZG_BADARGS;
        g_opts.replicas = (int)value;
    } else if (!strcmp(key, "streams")) {
        if (value < 1 || value > 64) return C_KZG_BADARGS;
        g_opts.streams = (int)value;
    } else if (!strcmp(key, "commit_wbits")) {
        if (value < 4 || value > 16) return C_KZG_BADARGS;
        g_opts.commit_wbits = (int)value;
    } else if (!strcmp(key, "fk20_wbits")) {
        if (value != 0 && (value < 4 || value > 16)) return C_KZG_BADARGS;
        g_opts.fk20_wbits = (int)value;
    } else if (!strcmp(key, "proof_wbits")) {
        if (value != 0 && (value < 4 || value > 16)) return C_KZG_BADARGS;
        g_opts.proof_wbits = (int)value;
    } else if (!strcmp(key, "gpu_sha_min")) {
        if (value < 0 || value > (1 << 30)) return C_KZG_BADARGS;
        g_gpu_sha_min.store((int)value);  // read at call time, unlike the load-time options
    } else if (!strcmp(key, "verify_pipe_min")) {
        if (value < 2 || value > (1 << 30)) return C_KZG_BADARGS;
        g_verify_pipe_min.store((int)value);
    } else if (!strcmp(key, "verify_call_table")) {
        if (value != 0 && value != 1) return C_KZG_BADARGS;
        g_verify_call_table.store((int)value);
    } else if (!strcmp(key, "verify_cu_partition")) {
        if (value != 0 && value != 1) return C_KZG_BADARGS;
        g_verify_cu_partition.store((int)value);   // read at call time
    } else if (!strcmp(key, "commit_graph")) {
        if (value < 0 || value > 2) return C_KZG_BADARGS;   // (2: diagnostic -- build the graph anew on every call)
        g_commit_graph.store((int)value);  // read at call time
    } else if (!strcmp(key, "host_threads")) {
        if (value < 0 || value > 1024) return C_KZG_BADARGS;
        g_host_threads.store((int)value);  // read when the helper pools start and at call time
    } else if (!strcmp(key, "async_tables")) {
        if (value != 0 && value != 1) return C_KZG_BADARGS;
        g_opts.async_tables = (int)value;
    } else if (!strcmp(key, "coalesce")) {
        if (value != 0 && value != 1) return C_KZG_BADARGS;
        g_opts.coalesce = (int)value;
    } else if (!strcmp(key, "coalesce_active")) {
        if (value < 1 || value > 8) return C_KZG_BADARGS;
        g_opts.coalesce_active = (int)value;
    } else if (!strcmp(key, "wait_deadline_ms")) {
        if (value < 1 || value > 86400000) return C_KZG_BADARGS;
        dev::wait_deadline_ms_ref().store(value);   // read at wait time
    } else if (!strcmp(key, "direct_max")) {
        if (value < -1 || value > 4096) return C_KZG_BADARGS;
        g_opts.direct_max = (int)value;
    } else {
        return C_KZG_BADARGS;
    }
    return C_KZG_OK;
}

extern "C" int ckzg_hip_host_thread_budget(void) { return host_thread_budget(); }

extern "C" void ckzg_hip_commit_graph_stats(uint64_t out[3]) {
    if (!out) return;
    for (int i = 0; i < 3; i++) out[i] = g_graph_stats[i].load(std::memory_order_relaxed);
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
        destroy_settings_ctx(s);   // no-op for a struct this library did not finish loading
    }
    free(s->roots_of_unity);
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

static C_KZG_RET load_trusted_setup_impl(KZGSettings *out, const uint8_t *g1_monomial_bytes,
                                         uint64_t num_g1_monomial_bytes,
                                         const uint8_t *g1_lagrange_bytes,
                                         uint64_t num_g1_lagrange_bytes,
                                         const uint8_t *g2_monomial_bytes,
                                         uint64_t num_g2_monomial_bytes, uint64_t precompute) {
    // setup.c:392-505
    C_KZG_RET ret = C_KZG_OK;
    const auto t_host = std::chrono::steady_clock::now();
    std::vector<G1Affine> lagr_affine(NUM_G1_POINTS), mono_affine(NUM_G1_POINTS);
    memset(out, 0, sizeof *out);
    if (precompute > 15) return C_KZG_BADARGS;
    out->wbits = precompute;
    if (num_g1_monomial_bytes != NUM_G1_POINTS * 48 || num_g1_lagrange_bytes != NUM_G1_POINTS * 48 ||
        num_g2_monomial_bytes != NUM_G2_POINTS * 96) {
        return C_KZG_BADARGS;
    }
    // (roots_of_unity doubles as the key under which the GPU state of this struct is registered)
    out->roots_of_unity = (fr_t *)calloc(FIELD_ELEMENTS_PER_EXT_BLOB + 1, sizeof(fr_t));
    if (!out->roots_of_unity) return C_KZG_MALLOC;
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
        const int budget = host_thread_budget();
        const size_t nt = budget >= 16 ? 16 : (size_t)budget;
        std::atomic<int> bad(0);
        JoinThreads th;
        for (size_t t = 0; t < nt; t++) {
            th.spawn([&, t]() {
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
        th.join();
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
    pending_load_times().ms[LP_HOST_POINTS] =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count();
    // GPU state: commitment tables, NTT twiddles, FK20 columns and tables (setup.c:238-330)
    ret = create_settings_ctx(out, lagr_affine.data(), mono_affine.data());
    if (ret != C_KZG_OK) goto fail;
    if (!tables_ready(out)) {
        // "async_tables": one commitment and one cells+proofs call on the default tables leave the first slot's
        // arena, scratch and pinned staging allocated, so that the caller's first calls need no allocation while the
        // widener's large ones are in flight (results ignored: a failure here is a failure of the caller's call later)
        std::vector<uint8_t> zero(BYTES_PER_BLOB, 0), cells((size_t)CELLS_PER_EXT_BLOB * BYTES_PER_CELL), proofs((size_t)CELLS_PER_EXT_BLOB * 48);
        KZGCommitment c;
        (void)blob_to_kzg_commitment(&c, reinterpret_cast<const Blob *>(zero.data()), out);
        (void)compute_cells_and_kzg_proofs(reinterpret_cast<Cell *>(cells.data()), reinterpret_cast<KZGProof *>(proofs.data()),
                                           reinterpret_cast<const Blob *>(zero.data()), out);
        start_widening(out);
    }
    return C_KZG_OK;
fail:
    free_trusted_setup(out);
    return ret;
}

extern "C" C_KZG_RET load_trusted_setup(KZGSettings *out, const uint8_t *g1_monomial_bytes,
                                        uint64_t num_g1_monomial_bytes,
                                        const uint8_t *g1_lagrange_bytes,
                                        uint64_t num_g1_lagrange_bytes,
                                        const uint8_t *g2_monomial_bytes,
                                        uint64_t num_g2_monomial_bytes, uint64_t precompute) {
    if (out) memset(out, 0, sizeof *out);
    C_KZG_RET ret = guarded([&]() {
        return load_trusted_setup_impl(out, g1_monomial_bytes, num_g1_monomial_bytes, g1_lagrange_bytes,
                                       num_g1_lagrange_bytes, g2_monomial_bytes, num_g2_monomial_bytes, precompute);
    });
    if (ret != C_KZG_OK && out) free_trusted_setup(out);  // also after an exception half-way through
    return ret;
}

static C_KZG_RET load_trusted_setup_file_impl(KZGSettings *out, FILE *in, uint64_t precompute) {
    // setup.c:519-600: "<n_g1> <n_g2>" then hex of: G1 Lagrange, G2 monomial, G1 monomial
    uint64_t n1 = 0, n2 = 0;
    memset(out, 0, sizeof *out);
    const auto t_parse = std::chrono::steady_clock::now();
    pending_load_times() = LoadTimes();
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
    const double parse_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_parse).count();
    C_KZG_RET ret = load_trusted_setup(out, mono.data(), mono.size(), lagr.data(), lagr.size(), g2.data(), g2.size(), precompute);
    if (ret == C_KZG_OK) {
        if (SettingsCtx *sc = settings_of(out, false)) {
            std::lock_guard<std::mutex> lock(sc->widen_mu);
            sc->load.ms[LP_HOST_PARSE] = parse_ms;
        }
    }
    return ret;
}

extern "C" C_KZG_RET load_trusted_setup_file(KZGSettings *out, FILE *in, uint64_t precompute) {
    if (out) memset(out, 0, sizeof *out);
    if (!out || !in) return C_KZG_BADARGS;   // (the reference dereferences both)
    return guarded([&]() { return load_trusted_setup_file_impl(out, in, precompute); });
}

// ------------------------------------------------------------------------------------------
// byte <-> element helpers (src/common/bytes.c)
// ------------------------------------------------------------------------------------------

extern "C" C_KZG_RET bytes_to_bls_field(fr_t *out, const Bytes32 *b) {
    return fr_from_bytes_canonical(*as_fr(out), b->bytes) ? C_KZG_OK : C_KZG_BADARGS;
}

extern "C" void bytes_from_bls_field(Bytes32 *out, const fr_t *in) { fr_to_bytes(out->bytes, *as_fr(in)); }

extern "C" void bytes_from_g1(Bytes48 *out, const g1_t *in) {
    g1_compress_affine(out->bytes, jac_to_affine_fast(*as_g1(in)));
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


extern "C" C_KZG_RET ckzg_hip_blob_to_kzg_commitment_batch_device(void *d_out48, void *d_status,
                                                                  const void *d_blobs, uint64_t n,
                                                                  const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        if (n == 0) return C_KZG_OK;
        const void *ptrs[3] = {d_out48, d_status, d_blobs};
        const int pool = pool_of_pointers(sc, ptrs, 3);
        if (pool < 0 || !d_out48 || !d_blobs) return C_KZG_BADARGS;
        Lease lease(s, pool);
        dev::DeviceCtx *ctx = lease.ctx;
        if (!ctx) return C_KZG_ERROR;
        return (C_KZG_RET)dev::commit_blobs_device(ctx, (uint8_t *)d_out48, (uint8_t *)d_status,
                                                   (const uint8_t *)d_blobs, n);
    });
}

// One device's share of a host-pointer commitment batch.
// Host buffers are pageable, and a pageable hipMemcpyAsync serialises with everything.  So chunk
// i+1 is memcpy'd by this thread into a pinned staging buffer and DMA'd on the copy stream while
// the kernels of chunk i execute on the compute stream.  Two staging/device buffers, events
// both ways.  Chunks grow geometrically (64, 192, 256, 256, ...): the first one is small so that
// the GPU starts after ~0.5 ms, and since a chunk computes ~3x longer than the next one takes to
// stage, the following ones can be large enough to run the kernels at full-batch efficiency -- but not larger
// than 256 blobs: the staging copy of a 512-blob chunk (64 MB on four threads) outlasts the kernels of a 192-blob
// chunk before it and the GPU waits (profiles/r03_commit_chunk_ab.txt).
// pinned_io: the caller vouches that `blobs` is page-locked (the combiner's batch buffer): DMA'd from in place.
static C_KZG_RET commit_batch_on(dev::DeviceCtx *ctx, KZGCommitment *out, uint8_t *status, const Blob *blobs,
                                 uint64_t n, bool pinned_io = false) {
    if (n == 0) return C_KZG_OK;
    // measured: profiles/r03_commit_chunk_ab.txt (512: -11 % from pageable memory)
    static const uint64_t CH = (uint64_t)(dev::ab_knob("CKZG_HIP_COMMIT_CHUNK", 256) < 16 ? 16 : (dev::ab_knob("CKZG_HIP_COMMIT_CHUNK", 256) > 1024 ? 1024 : dev::ab_knob("CKZG_HIP_COMMIT_CHUNK", 256)));
    const uint64_t FIRST = CH < 64 ? CH : 64;
    const uint64_t m = n < CH ? n : CH;
    Trace tr("commit_batch");
    // device temporaries from the slot's arena, events kept in the slot: a single-blob call
    // must not pay for hipMalloc/hipFree/hipEventCreate
    Arena &ar = ctx->api_arena;
    // One finalize for the whole batch (partial sums of every chunk parked in d_part8) when every chunk of the
    // schedule splits a blob into at most 8 partial sums; otherwise each chunk finalizes itself.
    const bool one_chunk = n <= FIRST || (pinned_io && n <= CH);   // (a coalesced batch: nothing to stage, one copy + one launch)
    bool deferred = !one_chunk;
    {
        uint64_t want = FIRST;
        for (uint64_t off = 0, k = 0; off < n && deferred; off += k) {
            k = n - off < want ? n - off : want;
            want = 3 * want < CH ? 3 * want : CH;
            deferred = dev::commit_chunk_fits8(ctx, k);
        }
    }
    if (!ar.begin(2 * m * BYTES_PER_BLOB + n * 49 + (deferred ? n * (8 * sizeof(G1XYZZ) + 4) : 0) + 2048)) return C_KZG_MALLOC;
    ArenaTrim trim(ar);
    ABuf<uint8_t> d_blobs[2] = {ABuf<uint8_t>(ar, m * BYTES_PER_BLOB), ABuf<uint8_t>(ar, one_chunk ? 1 : m * BYTES_PER_BLOB)};
    ABuf<uint8_t> d_out(ar, n * 49);  // commitments, then the status bytes
    ABuf<G1XYZZ> d_part8(ar, deferred ? n * 8 : 1);
    ABuf<uint32_t> d_bad_all(ar, deferred ? n : 1);
    if (!d_blobs[0].p || !d_blobs[1].p || !d_out.p || !d_part8.p || !d_bad_all.p) return C_KZG_MALLOC;
    uint8_t *d_status = d_out.p + n * 48;
    if (deferred) {
        if (hipMemsetAsync(d_part8.p, 0, n * 8 * sizeof(G1XYZZ), ctx->stream) != hipSuccess ||
            hipMemsetAsync(d_bad_all.p, 0, n * 4, ctx->stream) != hipSuccess)
            return C_KZG_ERROR;
    }
    hipEvent_t *copied = ctx->stage_ev, *consumed = ctx->stage_ev + 2;
    C_KZG_RET ret = C_KZG_OK;
    bool pending[2] = {false, false};
    if (dev::scratch_reserve(ctx, dev::commit_scratch_bytes(ctx, m)) != 0) return C_KZG_MALLOC;
    // pinned staging: two input buffers of up to CH blobs; the results come back through the first 49 n
    // bytes of a third one (a pageable destination would make the final copy a blocking staged copy)
    if (!ensure_pinned(ctx->h_stage, ctx->h_stage_bytes, n == 1 || pinned_io ? (size_t)BYTES_PER_BLOB : CH * BYTES_PER_BLOB)) return C_KZG_MALLOC;
    for (int i = 0; i < 4; i++) {
        if (!ctx->stage_ev[i] && hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming) != hipSuccess) {
            ctx->stage_ev[i] = nullptr;
            ret = C_KZG_ERROR;
        }
    }
    tr.mark("buffers + events");
    if (one_chunk && ret == C_KZG_OK) {
        // one small chunk (the reference-shaped single-blob call among them): nothing to overlap, so the
        // copy in, the kernels and the copy out run on the one compute stream with a single wait
        const uint8_t *h_in = reinterpret_cast<const uint8_t *>(blobs);
        uint8_t *h_res = static_cast<uint8_t *>(ctx->h_stage[1]);
        if (!pinned_io) {
            memcpy(ctx->h_stage[0], blobs, n * BYTES_PER_BLOB);
            h_in = static_cast<const uint8_t *>(ctx->h_stage[0]);
        }
        auto enqueue_all = [&]() -> C_KZG_RET {
            if (hipMemcpyAsync(d_blobs[0].p, h_in, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                return C_KZG_ERROR;
            int rc = dev::commit_blobs_enqueue(ctx, d_out.p, d_status, (const uint8_t *)d_blobs[0].p, n);
            if (rc) return (C_KZG_RET)rc;
            if (hipMemcpyAsync(h_res, d_out.p, n * 49, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return C_KZG_ERROR;
            return C_KZG_OK;
        };
        bool launched = false;
        if (n == 1 && !pinned_io && !ctx->one_commit.unusable && g_commit_graph.load(std::memory_order_relaxed) != 0) {
            // the reference-shaped call: its six dependent nodes go to the device as ONE graph
            auto &g = ctx->one_commit;
            const void *key[7] = {ctx->commit.d_table, d_blobs[0].p, d_out.p, ctx->scratch.ptr, h_in, h_res,
                                  reinterpret_cast<const void *>((uintptr_t)ctx->commit.wbits)};
            bool have = g.exec && memcmp(g.key, key, sizeof key) == 0 && g_commit_graph.load(std::memory_order_relaxed) != 2;
            if (!have) {
                // built node by node (dev::commit_one_graph_build): no stream capture, so no quiet section and nothing a
                // HIP call of another thread -- of this library or of anything else in the process -- can invalidate.
                // Rare: once per slot and table set, again when a slot's buffers moved.
                g_graph_stats[0].fetch_add(1, std::memory_order_relaxed);
                if (g.exec) (void)hipGraphExecDestroy(g.exec);
                g.exec = nullptr;
                const int rc = dev::commit_one_graph_build(ctx, &g.exec, d_out.p, d_status, (const uint8_t *)d_blobs[0].p, h_in, h_res);
                if (rc != 0) {
                    (void)hipGetLastError();
                    g.exec = nullptr;
                    g.unusable = true;   // plain launches from now on (rc 4: this table geometry has no raw-partials form)
                    g_graph_stats[1].fetch_add(1, std::memory_order_relaxed);
                } else {
                    memcpy(g.key, key, sizeof key);
                    have = true;
                }
            }
            if (have) {
                if (hipGraphLaunch(g.exec, ctx->stream) != hipSuccess) return C_KZG_ERROR;
                g_graph_stats[2].fetch_add(1, std::memory_order_relaxed);
                launched = true;
            }
        }
        if (!launched) {
            const C_KZG_RET rc = enqueue_all();
            if (rc != C_KZG_OK) return rc;
        }
        if ((n == 1 ? wait_stream_low_latency(ctx->stream) : dev::sync_stream(ctx->stream)) != hipSuccess) return C_KZG_ERROR;
        if (launched) {
            for (int i = 0; i < 4; i++) ctx->last_ms[i] = -1.0f;   // (events recorded by graph nodes carry no readable time stamps)
        } else {
            dev::commit_collect_times(ctx);
        }
        tr.mark("copy in + kernels + copy out");
        memcpy(out, h_res, n * 48);
        for (uint64_t i = 0; i < n; i++) {
            if (status) status[i] = h_res[n * 48 + i];
            if (h_res[n * 48 + i]) ret = C_KZG_BADARGS;
        }
        return ret;
    }
    const bool src_pinned = pinned_io || host_pointer_is_pinned(blobs);  // page-locked caller memory: no staging copy
    uint64_t chunk = 0, k = 0, want = FIRST;
    for (uint64_t off = 0; off < n && ret == C_KZG_OK; off += k, chunk++) {
        const int b = (int)(chunk & 1);
        k = n - off < want ? n - off : want;
        want = 3 * want < CH ? 3 * want : CH;
        bool ok = true;
        if (pending[b]) {
            // the pinned buffer is free once its DMA finished; the device buffer once its kernels did
            ok = ok && dev::sync_event(copied[b]) == hipSuccess;
            ok = ok && hipStreamWaitEvent(ctx->copy_stream, consumed[b], 0) == hipSuccess;
        }
        const void *h_src = blobs + off;
        if (!src_pinned) {
            staged_copy(ctx->h_stage[b], blobs + off, k * BYTES_PER_BLOB);
            h_src = ctx->h_stage[b];
        }
        ok = ok && hipMemcpyAsync(d_blobs[b].p, h_src, k * BYTES_PER_BLOB, hipMemcpyHostToDevice,
                                  ctx->copy_stream) == hipSuccess;
        ok = ok && hipEventRecord(copied[b], ctx->copy_stream) == hipSuccess;
        ok = ok && hipStreamWaitEvent(ctx->stream, copied[b], 0) == hipSuccess;
        if (!ok) {
            ret = C_KZG_ERROR;
            break;
        }
        int rc = deferred ? dev::commit_accumulate8_enqueue(ctx, d_part8.p + off * 8, d_bad_all.p + off, (const uint8_t *)d_blobs[b].p, k)
                          : dev::commit_blobs_enqueue(ctx, d_out.p + off * 48, d_status + off, (const uint8_t *)d_blobs[b].p, k);
        if (rc) {
            ret = rc == 4 ? C_KZG_ERROR : (C_KZG_RET)rc;
            break;
        }
        if (hipEventRecord(consumed[b], ctx->stream) != hipSuccess) ret = C_KZG_ERROR;
        pending[b] = true;
    }
    if (deferred && ret == C_KZG_OK) {
        int rc = dev::commit_finalize8_enqueue(ctx, d_out.p, d_status, d_part8.p, d_bad_all.p, n);
        if (rc) ret = (C_KZG_RET)rc;
    }
    tr.mark("staging loop (copies + enqueues)");
    if (dev::sync_stream(ctx->copy_stream) != hipSuccess) ret = ret == C_KZG_OK ? C_KZG_ERROR : ret;
    // results: one async copy on the compute stream into pinned memory (free again: every input DMA is done)
    uint8_t *h_res = static_cast<uint8_t *>(ctx->h_stage[0]);
    const bool pinned_res = n * 49 <= ctx->h_stage_bytes;
    std::vector<uint8_t> pageable_res;
    if (!pinned_res) {
        pageable_res.resize(n * 49);
        h_res = pageable_res.data();
    }
    if (ret == C_KZG_OK && hipMemcpyAsync(h_res, d_out.p, n * 49, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        ret = C_KZG_ERROR;
    if (dev::sync_stream(ctx->stream) != hipSuccess) ret = ret == C_KZG_OK ? C_KZG_ERROR : ret;
    tr.mark("wait for the GPU");
    if (ret != C_KZG_OK) return ret;
    memcpy(out, h_res, n * 48);
    const uint8_t *st = h_res + n * 48;
    for (uint64_t i = 0; i < n; i++) {
        if (status) status[i] = st[i];
        if (st[i]) ret = C_KZG_BADARGS;
    }
    return ret;
}

extern "C" C_KZG_RET ckzg_hip_blob_to_kzg_commitment_batch(KZGCommitment *out, uint8_t *status,
                                                           const Blob *blobs, uint64_t n,
                                                           const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        if (!settings_of(s)) return C_KZG_ERROR;
        if (n == 0) return C_KZG_OK;
        // below 64 blobs per device the launch overheads outweigh a second GPU
        return for_each_device_shard(s, n, 64, [&](dev::DeviceCtx *ctx, uint64_t lo, uint64_t hi) {
            return commit_batch_on(ctx, out + lo, status ? status + lo : nullptr, blobs + lo, hi - lo);
        });
    });
}

// src/eip4844/eip4844.c:264-280.  One blob per call is all the reference API offers; threads that call it
// concurrently on one KZGSettings (bindings/go/main_test.go:953-971) are served by shared batch launches
// (combiner.hpp), a lone caller by its own launch as before.
extern "C" C_KZG_RET blob_to_kzg_commitment(KZGCommitment *out, const Blob *blob, const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        auto solo = [&]() -> C_KZG_RET {
            uint8_t st = 0;
            return ckzg_hip_blob_to_kzg_commitment_batch(out, &st, blob, 1, s);
        };
        Combiner *cb = sc->comb[CB_COMMIT];
        if (!cb) return solo();
        return cb->submit(
            nullptr, 0, solo,
            [&](uint8_t *h_in, size_t idx) { memcpy(h_in + idx * BYTES_PER_BLOB, blob, BYTES_PER_BLOB); },
            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *st, size_t n) -> C_KZG_RET {
                Lease lease(s);
                if (!lease.ctx) return C_KZG_ERROR;
                return commit_batch_on(lease.ctx, reinterpret_cast<KZGCommitment *>(h_out), st,
                                       reinterpret_cast<const Blob *>(h_in), n, true);
            },
            [&](const uint8_t *h_out, size_t idx, size_t) { memcpy(out, h_out + idx * 48, 48); });
    });
}

// ------------------------------------------------------------------------------------------
// compute_cells_and_kzg_proofs (src/eip7594/eip7594.c:61-157) and its batch forms
// ------------------------------------------------------------------------------------------

extern "C" C_KZG_RET ckzg_hip_compute_cells_and_kzg_proofs_batch_device(void *d_cells, void *d_proofs,
                                                                        void *d_status,
                                                                        const void *d_blobs, uint64_t n,
                                                                        const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        if (d_cells == NULL && d_proofs == NULL) return C_KZG_BADARGS;
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        if (n == 0) return C_KZG_OK;
        const void *ptrs[4] = {d_cells, d_proofs, d_status, d_blobs};
        const int pool = pool_of_pointers(sc, ptrs, 4);
        if (pool < 0 || !d_blobs) return C_KZG_BADARGS;
        Lease lease(s, pool);
        dev::DeviceCtx *ctx = lease.ctx;
        if (!ctx) return C_KZG_ERROR;
        return (C_KZG_RET)dev::cells_and_proofs_device(ctx, (uint8_t *)d_cells, (uint8_t *)d_proofs,
                                                       (uint8_t *)d_status, (const uint8_t *)d_blobs, n);
    });
}

static bool ensure_stage_events(dev::DeviceCtx *ctx) {
    for (int i = 0; i < 4; i++) {
        if (!ctx->stage_ev[i] && hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming) != hipSuccess) {
            ctx->stage_ev[i] = nullptr;
            return false;
        }
    }
    return true;
}

static size_t al256(size_t v) { return (v + 255) / 256 * 256; }

// One device's share of a host-pointer compute_cells_and_kzg_proofs batch.
// pinned_io (any n the caller's buffers hold; the combiner: <= 128): the caller vouches that blobs and the outputs are page-locked and that the outputs
// are laid out [cells of n blobs][proofs of n blobs][n spare bytes] (the forms wanted only): the combiner's batch
// buffers, DMA'd from and into in place.
static C_KZG_RET cells_and_proofs_batch_on(dev::DeviceCtx *ctx, Cell *cells, KZGProof *proofs, uint8_t *status,
                                           const Blob *blobs, uint64_t n, bool pinned_io = false) {
    if (n == 0) return C_KZG_OK;
    const size_t cells_per = (size_t)CELLS_PER_EXT_BLOB * BYTES_PER_CELL, proofs_per = (size_t)CELLS_PER_EXT_BLOB * 48;
    Arena &ar = ctx->api_arena;
    ArenaTrim trim(ar);
    C_KZG_RET ret = C_KZG_OK;
    if (n <= 64 || pinned_io) {
        // latency path (the reference-shaped one-blob call among them): pinned staging both ways, one stream
        const size_t out_per = (cells ? cells_per : 0) + (proofs ? proofs_per : 0) + 1;
        if (!ar.begin(n * (BYTES_PER_BLOB + out_per) + 1024)) return C_KZG_MALLOC;
        ABuf<uint8_t> d_blobs(ar, n * BYTES_PER_BLOB), d_out(ar, n * out_per);
        if (!d_blobs.p || !d_out.p) return C_KZG_MALLOC;
        uint8_t *d_cells = cells ? d_out.p : nullptr;
        uint8_t *d_proofs = proofs ? d_out.p + (cells ? n * cells_per : 0) : nullptr;
        uint8_t *d_status = d_out.p + n * (out_per - 1);
        const uint8_t *h_in = reinterpret_cast<const uint8_t *>(blobs);
        uint8_t *h = reinterpret_cast<uint8_t *>(cells ? (void *)cells : (void *)proofs);
        if (pinned_io) {
            if (cells && proofs && reinterpret_cast<uint8_t *>(proofs) != h + n * cells_per) return C_KZG_ERROR;   // not the promised layout
        } else {
            if (!ensure_pinned(ctx->h_stage, ctx->h_stage_bytes, n == 1 ? (size_t)BYTES_PER_BLOB : 64 * (size_t)BYTES_PER_BLOB))
                return C_KZG_MALLOC;
            if (!ensure_pinned(ctx->h_out, ctx->h_out_bytes, n == 1 ? cells_per + proofs_per + 64 : OutPipe::PIECE)) return C_KZG_MALLOC;
            memcpy(ctx->h_stage[0], blobs, n * BYTES_PER_BLOB);
            h_in = static_cast<const uint8_t *>(ctx->h_stage[0]);
            h = static_cast<uint8_t *>(ctx->h_out[0]);
        }
        if (hipMemcpyAsync(d_blobs.p, h_in, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return C_KZG_ERROR;
        bool cells_copied = false;   // (the device function may copy the cells out itself, underneath the proof kernels)
        int rc = dev::cells_and_proofs_device(ctx, d_cells, d_proofs, d_status, d_blobs.p, n, cells && proofs ? h : nullptr, &cells_copied);
        if (rc) return (C_KZG_RET)rc;
        const size_t done = cells_copied ? n * cells_per : 0;
        if (hipMemcpyAsync(h + done, d_out.p + done, n * out_per - done, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return C_KZG_ERROR;
        // (not the polling wait of the one-blob commitment: this call runs kernels on two streams, and a thread that polls
        // the runtime slows the runtime's own hand-over between them -- measured 1.63 -> 1.80 ms)
        if (dev::sync_stream(ctx->stream) != hipSuccess) return C_KZG_ERROR;
        if (!pinned_io) {
            if (cells) memcpy(cells, h, n * cells_per);
            if (proofs) memcpy(proofs, h + (cells ? n * cells_per : 0), n * proofs_per);
        }
        const uint8_t *st = h + n * (out_per - 1);
        for (uint64_t i = 0; i < n; i++) {
            if (status) status[i] = st[i];
            if (st[i]) ret = C_KZG_BADARGS;
        }
        return ret;
    }
    // Throughput path.  Per chunk of up to 2048 blobs (>= 2 waves of G1-FFT butterflies per SIMD per stage
    // launch): sub-chunks of 256 blobs are staged into pinned memory and DMA'd on the copy stream while the
    // previous sub-chunk's Fr kernels (bytes -> coefficients -> cells) run; the cells of each sub-chunk start
    // draining to the caller's memory (OutPipe) as soon as they exist, so that the bulk of the output
    // (256 KB per blob) crosses PCIe underneath the proof kernels that follow; proofs and status bytes last.
    const bool direct = proofs != nullptr && dev::proofs_use_direct(ctx, n);
    const uint64_t CH = direct ? 64 : 2048, SC = direct ? 64 : 256;
    const uint64_t m = n < CH ? n : CH;
    if (!ar.begin(2 * SC * BYTES_PER_BLOB + 2 * m * ((cells ? cells_per : 0) + (proofs ? proofs_per : 0) + 1) + 4096))
        return C_KZG_MALLOC;
    ABuf<uint8_t> d_in[2] = {ABuf<uint8_t>(ar, SC * BYTES_PER_BLOB), ABuf<uint8_t>(ar, SC * BYTES_PER_BLOB)};
    ABuf<uint8_t> d_cells[2] = {ABuf<uint8_t>(ar, cells ? m * cells_per : 1), ABuf<uint8_t>(ar, cells ? m * cells_per : 1)};
    ABuf<uint8_t> d_proofs[2] = {ABuf<uint8_t>(ar, proofs ? m * proofs_per : 1), ABuf<uint8_t>(ar, proofs ? m * proofs_per : 1)};
    ABuf<uint8_t> d_status[2] = {ABuf<uint8_t>(ar, m), ABuf<uint8_t>(ar, m)};
    for (int i = 0; i < 2; i++) {
        if (!d_in[i].p || !d_cells[i].p || !d_proofs[i].p || !d_status[i].p) return C_KZG_MALLOC;
    }
    const size_t poly_b = al256(m * FIELD_ELEMENTS_PER_BLOB * sizeof(Fr)), ext_b = al256(SC * FIELD_ELEMENTS_PER_EXT_BLOB * sizeof(Fr)),
                 bad_b = al256(m * 4);
    const size_t proof_b = proofs ? dev::proofs_scratch_bytes(ctx, m, direct) : 0;
    int rc = dev::scratch_reserve(ctx, poly_b + ext_b + bad_b + proof_b);
    if (rc) return (C_KZG_RET)rc;
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch.ptr);
    Fr *d_poly = reinterpret_cast<Fr *>(base), *d_ext = reinterpret_cast<Fr *>(base + poly_b);
    uint32_t *d_bad = reinterpret_cast<uint32_t *>(base + poly_b + ext_b);
    uint8_t *proof_scratch = base + poly_b + ext_b + bad_b;
    if (!ensure_pinned(ctx->h_stage, ctx->h_stage_bytes, SC * (size_t)BYTES_PER_BLOB)) return C_KZG_MALLOC;
    if (!ensure_stage_events(ctx)) return C_KZG_ERROR;
    hipEvent_t *copied = ctx->stage_ev, *consumed = ctx->stage_ev + 2;
    std::vector<uint8_t> st(n);
    std::vector<size_t> mark;
    OutPipe pipe(ctx);
    struct StreamDrain {  // nothing may still read the arena or the staging buffers when this function leaves
        dev::DeviceCtx *c;
        OutPipe &p;
        ~StreamDrain() {
            (void)p.finish();
            (void)dev::sync_stream(c->copy_stream);
            (void)dev::sync_stream(c->stream);
        }
    } drain{ctx, pipe};
    bool pending[2] = {false, false};
    const bool src_pinned = host_pointer_is_pinned(blobs);
    uint64_t sub_index = 0, chunk = 0;
    for (uint64_t off = 0; off < n; off += CH, chunk++) {
        const uint64_t k = n - off < CH ? n - off : CH;
        const int cb = (int)(chunk & 1);
        if (chunk >= 2) pipe.wait_for(mark[chunk - 2]);  // this chunk's output buffers have been drained
        if (hipMemsetAsync(d_bad, 0, k * 4, ctx->stream) != hipSuccess) return C_KZG_ERROR;
        for (uint64_t so = 0; so < k; so += SC, sub_index++) {
            const uint64_t ks = k - so < SC ? k - so : SC;
            const int b = (int)(sub_index & 1);
            bool ok = true;
            if (pending[b]) {
                ok = ok && dev::sync_event(copied[b]) == hipSuccess;
                ok = ok && hipStreamWaitEvent(ctx->copy_stream, consumed[b], 0) == hipSuccess;
            }
            const void *h_src = blobs + off + so;
            if (!src_pinned) {
                staged_copy(ctx->h_stage[b], blobs + off + so, ks * BYTES_PER_BLOB);
                h_src = ctx->h_stage[b];
            }
            ok = ok && hipMemcpyAsync(d_in[b].p, h_src, ks * BYTES_PER_BLOB, hipMemcpyHostToDevice,
                                      ctx->copy_stream) == hipSuccess;
            ok = ok && hipEventRecord(copied[b], ctx->copy_stream) == hipSuccess;
            ok = ok && hipStreamWaitEvent(ctx->stream, copied[b], 0) == hipSuccess;
            if (!ok) return C_KZG_ERROR;
            uint8_t *dc = cells ? d_cells[cb].p + so * cells_per : nullptr;
            rc = dev::cells_stage_enqueue(ctx, dc, d_poly + so * FIELD_ELEMENTS_PER_BLOB, d_ext, d_bad + so, d_in[b].p, ks);
            if (rc) return (C_KZG_RET)rc;
            if (hipEventRecord(consumed[b], ctx->stream) != hipSuccess) return C_KZG_ERROR;
            pending[b] = true;
            if (cells && !pipe.push(dc, cells + (off + so) * CELLS_PER_EXT_BLOB, ks * cells_per)) return C_KZG_ERROR;
        }
        if (proofs) {
            rc = dev::proofs_stage_enqueue(ctx, d_proofs[cb].p, d_poly, k, proof_scratch, direct);
            if (rc) return (C_KZG_RET)rc;
            if (!pipe.push(d_proofs[cb].p, proofs + off * CELLS_PER_EXT_BLOB, k * proofs_per)) return C_KZG_ERROR;
        }
        rc = dev::bad_to_status_enqueue(ctx, d_status[cb].p, d_bad, k);
        if (rc) return (C_KZG_RET)rc;
        if (!pipe.push(d_status[cb].p, st.data() + off, k)) return C_KZG_ERROR;
        mark.push_back(pipe.pushed_count());
    }
    if (pipe.finish() != C_KZG_OK) return C_KZG_ERROR;
    if (dev::sync_stream(ctx->stream) != hipSuccess) return C_KZG_ERROR;
    for (uint64_t i = 0; i < n; i++) {
        if (status) status[i] = st[i];
        if (st[i]) ret = C_KZG_BADARGS;
    }
    return ret;
}

extern "C" C_KZG_RET ckzg_hip_compute_cells_and_kzg_proofs_batch(Cell *cells, KZGProof *proofs,
                                                                 uint8_t *status, const Blob *blobs,
                                                                 uint64_t n, const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        if (cells == NULL && proofs == NULL) return C_KZG_BADARGS;  // eip7594.c:72-74
        if (!settings_of(s)) return C_KZG_ERROR;
        if (n == 0) return C_KZG_OK;
        return for_each_device_shard(s, n, 32, [&](dev::DeviceCtx *ctx, uint64_t lo, uint64_t hi) {
            return cells_and_proofs_batch_on(ctx, cells ? cells + lo * CELLS_PER_EXT_BLOB : nullptr,
                                             proofs ? proofs + lo * CELLS_PER_EXT_BLOB : nullptr,
                                             status ? status + lo : nullptr, blobs + lo, hi - lo);
        });
    });
}

// src/eip7594/eip7594.c:61-157.  Concurrent callers share launches of the batch path (combiner.hpp): from the
// hand-over point of the two proof algorithms upwards that is ONE FK20 pass over all of them instead of a
// chip-filling direct pass each.
extern "C" C_KZG_RET compute_cells_and_kzg_proofs(Cell *cells, KZGProof *proofs, const Blob *blob,
                                                  const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        if (cells == NULL && proofs == NULL) return C_KZG_BADARGS;  // eip7594.c:72-74
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        auto solo = [&]() -> C_KZG_RET {
            uint8_t st = 0;
            return ckzg_hip_compute_cells_and_kzg_proofs_batch(cells, proofs, &st, blob, 1, s);
        };
        Combiner *cb = sc->comb[cells && proofs ? CB_CELLS_PROOFS : (cells ? CB_CELLS : CB_PROOFS)];
        if (!cb) return solo();
        const size_t cells_per = (size_t)CELLS_PER_EXT_BLOB * BYTES_PER_CELL, proofs_per = (size_t)CELLS_PER_EXT_BLOB * 48;
        return cb->submit(
            nullptr, 0, solo,
            [&](uint8_t *h_in, size_t idx) { memcpy(h_in + idx * BYTES_PER_BLOB, blob, BYTES_PER_BLOB); },
            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *st, size_t n) -> C_KZG_RET {
                Lease lease(s);
                if (!lease.ctx) return C_KZG_ERROR;
                return cells_and_proofs_batch_on(lease.ctx, cells ? reinterpret_cast<Cell *>(h_out) : nullptr,
                                                 proofs ? reinterpret_cast<KZGProof *>(h_out + (cells ? n * cells_per : 0)) : nullptr,
                                                 st, reinterpret_cast<const Blob *>(h_in), n, true);
            },
            [&](const uint8_t *h_out, size_t idx, size_t n) {
                if (cells) memcpy(cells, h_out + idx * cells_per, cells_per);
                if (proofs) memcpy(proofs, h_out + (cells ? n * cells_per : 0) + idx * proofs_per, proofs_per);
            });
    });
}

// Timings of the most recent call on this KZGSettings (bench.py is single-threaded): every lease is numbered
// (SettingsCtx::lease_seq); a call that was fanned out over P pools took the last P leases, one per pool, so the
// answer is the maximum over the pools whose latest lease is among the last P -- for a one-pool call, or a
// single-unit call that went round robin to some pool, that is exactly the slot that served it.
extern "C" double ckzg_hip_last_kernel_ms(const KZGSettings *s, int which) {
    SettingsCtx *sc = settings_of(s);
    if (!sc || sc->pools.empty() || which < 0 || which > 5) return -1.0;
    const uint64_t now = sc->lease_seq.load();
    if (now == 0) return -1.0;
    uint64_t newest = 0;
    for (auto *p : sc->pools) newest = std::max(newest, p->last_seq.load());
    // pools that took part in the newest call: consecutive lease numbers ending at `newest`
    double best = -1.0;
    for (auto *p : sc->pools) {
        const uint64_t q = p->last_seq.load();
        dev::DeviceCtx *c = p->last.load();
        if (!c || q == 0 || newest - q >= sc->pools.size()) continue;
        if (newest - q > 0 && sc->pools.size() == 1) continue;
        best = std::max(best, (double)c->last_ms[which]);
    }
    return best;
}

extern "C" C_KZG_RET ckzg_hip_wait_tables(const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        if (!settings_of(s)) return C_KZG_ERROR;
        wait_for_tables(s);
        return C_KZG_OK;
    });
}

extern "C" int ckzg_hip_tables_ready(const KZGSettings *s) { return tables_ready(s) ? 1 : 0; }

extern "C" int ckzg_hip_load_times(const KZGSettings *s, double *ms, int n) {
    SettingsCtx *sc = settings_of(s, false);
    if (!sc || !ms) return 0;
    int k = n < (int)LP_COUNT ? n : (int)LP_COUNT;
    std::lock_guard<std::mutex> lock(sc->widen_mu);   // a running widener adds its phases under this lock
    for (int i = 0; i < k; i++) ms[i] = sc->load.ms[i];
    return k;
}

extern "C" uint64_t ckzg_hip_table_bytes(const KZGSettings *s) {
    SettingsCtx *sc = settings_of(s);
    if (!sc) return 0;
    uint64_t total = 0;
    for (auto *p : sc->pools) {
        std::lock_guard<std::mutex> lock(p->mu);
        total += p->pub.commit.bytes() + p->pub.fk20.bytes() + p->pub.mono.bytes();
    }
    return total;
}

extern "C" int ckzg_hip_table_wbits(const KZGSettings *s, int which) {
    SettingsCtx *sc = settings_of(s);
    if (!sc) return -1;
    DevicePool *p = sc->pools[0];
    std::lock_guard<std::mutex> lock(p->mu);
    switch (which) {
        case 0: return p->pub.commit.wbits;
        case 1: return p->pub.fk20.wbits;
        case 2: return p->pub.mono.d_table ? p->pub.mono.wbits : 0;
        default: return -1;
    }
}

extern "C" int ckzg_hip_coalesce_stats(const KZGSettings *s, int op, uint64_t *out, int n) {
    SettingsCtx *sc = settings_of(s, false);
    if (!sc || !out || op < 0 || op >= (int)CB_COUNT || !sc->comb[op]) return 0;
    const Combiner::Stats st = sc->comb[op]->stats();
    const uint64_t v[9] = {st.calls, st.solo, st.batches, st.batched, st.largest, st.run_us, st.retried, st.rescued, st.gave_up};
    int k = n < 9 ? n : 9;
    for (int i = 0; i < k; i++) out[i] = v[i];
    return k;
}

extern "C" void ckzg_hip_debug_dump(int fd) { debug_dump(fd); }

extern "C" int ckzg_hip_wait_stats(uint64_t *out, int n) {
    const uint64_t v[3] = {(uint64_t)dev::wait_deadline_ms(), dev::expired_waits_ref().load(), dev::wedged_devices_ref().load()};
    const int k = n < 3 ? (n < 0 ? 0 : n) : 3;
    for (int i = 0; out && i < k; i++) out[i] = v[i];
    return out ? k : 0;
}

extern "C" int ckzg_hip_num_devices(const KZGSettings *s) {
    SettingsCtx *sc = settings_of(s, false);
    return sc ? (int)sc->pools.size() : 0;
}
