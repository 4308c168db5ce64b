// device_ctx.hip -- creation / destruction of the GPU state of a loaded KZGSettings: device selection,
// per-device table construction, the slot pools (stream + scratch per concurrent call), the registry
// that maps a KZGSettings to its state, and slot leasing.
// This is the GPU half of load_trusted_setup (src/setup/setup.c:392-505): the reference builds
// x_ext_fft_columns and (optionally) blst fixed-base tables on the CPU (setup.c:238-330); here the
// 64 G1 FFTs and all tables are produced by kernels and stay resident in HBM.
#include <algorithm>
#include <chrono>
#include <shared_mutex>
#include <unordered_map>

#include "api_common.hpp"
#include "combiner.hpp"

namespace ckzg {
namespace api {

// ------------------------------------------------------------------------------------------
// registry: KZGSettings::roots_of_unity -> SettingsCtx
// ------------------------------------------------------------------------------------------

namespace {
std::shared_mutex g_reg_mu;
std::unordered_map<const void *, SettingsCtx *> &registry() {
    static std::unordered_map<const void *, SettingsCtx *> r;
    return r;
}
}  // namespace

SettingsCtx *settings_of(const KZGSettings *s, bool complain) {
    SettingsCtx *sc = nullptr;
    if (s && s->roots_of_unity) {
        std::shared_lock<std::shared_mutex> lock(g_reg_mu);
        auto it = registry().find(s->roots_of_unity);
        if (it != registry().end()) sc = it->second;
    }
    if (!sc && complain)
        fprintf(stderr, "[ckzg-hip] KZGSettings has no GPU context (not loaded by this library, or freed)\n");
    return sc;
}

// ------------------------------------------------------------------------------------------
// leases
// ------------------------------------------------------------------------------------------

void Lease::take(DevicePool *p) {
    std::unique_lock<std::mutex> lock(p->mu);
    // every slot busy: wait for one, but not for ever (a call that cannot get a stream slot by the deadline fails with
    // C_KZG_ERROR: every caller of the Lease checks ctx), and not at all on a device whose last wait expired
    if (dev::device_wedged(p->device) ||
        !cv_wait_bounded(p->cv, lock, [p]() { return !p->free_slots.empty() || dev::device_wedged(p->device); }, "stream slot", p) ||
        dev::device_wedged(p->device)) {
        static std::atomic<bool> said{false};
        if (dev::device_wedged(p->device) && !said.exchange(true))
            fprintf(stderr, "[ckzg-hip] device %d did not answer within the wait deadline earlier: calls on it fail with C_KZG_ERROR\n", p->device);
        return;
    }
    int idx = p->free_slots.back();
    p->free_slots.pop_back();
    ctx = p->slots[idx];
    if (ctx->tables_version != p->pub.version) {   // a wider table was published since this slot last ran
        ctx->commit = p->pub.commit;
        ctx->mono = p->pub.mono;
        ctx->fk20 = p->pub.fk20;
        ctx->direct_max = p->pub.direct_max;
        ctx->tables_version = p->pub.version;
    }
    lock.unlock();
    pool = p;
    p->last.store(ctx, std::memory_order_relaxed);
    if (p->owner) p->last_seq.store(p->owner->lease_seq.fetch_add(1, std::memory_order_relaxed) + 1, std::memory_order_relaxed);
    if (hipSetDevice(ctx->device) != hipSuccess) {
        fprintf(stderr, "[ckzg-hip] hipSetDevice(%d) failed\n", ctx->device);
        std::lock_guard<std::mutex> relock(p->mu);
        p->free_slots.push_back(idx);
        p->cv.notify_one();
        pool = nullptr;
        ctx = nullptr;
    }
}

Lease::Lease(DevicePool *p) { take(p); }

Lease::Lease(const KZGSettings *s, int pool_index) {
    SettingsCtx *sc = settings_of(s);
    if (!sc || sc->pools.empty()) return;
    const size_t np = sc->pools.size();
    if (pool_index >= 0) {
        take(sc->pools[(size_t)pool_index % np]);
        return;
    }
    // round robin over the pools, preferring one that has a free slot right now
    const unsigned start = sc->next.fetch_add(1, std::memory_order_relaxed);
    for (size_t k = 0; k < np; k++) {
        DevicePool *p = sc->pools[(start + k) % np];
        bool has_free;
        {
            std::lock_guard<std::mutex> lock(p->mu);
            has_free = !p->free_slots.empty();
        }
        if (has_free) {
            take(p);
            return;
        }
    }
    take(sc->pools[start % np]);
}

Lease::~Lease() {
    if (!pool || !ctx) return;
    // a slot whose device stopped answering is never handed out again: its kernels may still be running on its buffers
    if (dev::device_wedged(pool->device)) {
        pool->cv.notify_all();
        return;
    }
    {
        std::lock_guard<std::mutex> lock(pool->mu);
        pool->free_slots.push_back(ctx->slot);
    }
    pool->cv.notify_one();
}

// The pool that serves a *_device call: every non-null pointer must be device (or managed) memory of ONE device, and
// that device must hold tables of this KZGSettings.  -1 otherwise: a host pointer, a pointer the runtime does not
// know, buffers on different GPUs or on a GPU this KZGSettings was not loaded on would otherwise be dereferenced by
// kernels of some other device -- a fault, or silent peer access.
int pool_of_pointers(const SettingsCtx *sc, const void *const *ptrs, int count) {
    int device = -1;
    for (int i = 0; i < count; i++) {
        if (!ptrs[i]) continue;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ptrs[i]) != hipSuccess) {
            (void)hipGetLastError();   // an ordinary malloc'd pointer is "invalid value" to the runtime
            return -1;
        }
        if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) return -1;
        if (device >= 0 && at.device != device) return -1;
        device = at.device;
    }
    if (device < 0) return -1;
    for (size_t i = 0; i < sc->pools.size(); i++) {
        if (sc->pools[i]->device == device) return (int)i;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------------

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

static void destroy_slot(dev::DeviceCtx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)dev::sync_stream(ctx->stream);
    if (ctx->copy_stream) (void)dev::sync_stream(ctx->copy_stream);
    if (ctx->owns_tables) {   // (the fixed-base tables themselves belong to the pool: destroy_pool)
        if (ctx->d_lagr) (void)hipFree(ctx->d_lagr);
        if (ctx->d_xext) (void)hipFree(ctx->d_xext);
        if (ctx->d_roots) (void)hipFree(ctx->d_roots);
        if (ctx->d_brp_roots) (void)hipFree(ctx->d_brp_roots);
        if (ctx->d_brp_roots29) (void)hipFree(ctx->d_brp_roots29);
        if (ctx->d_roots_raw) (void)hipFree(ctx->d_roots_raw);
        if (ctx->d_mono) (void)hipFree(ctx->d_mono);
        if (ctx->d_shift) (void)hipFree(ctx->d_shift);
        if (ctx->d_unshift) (void)hipFree(ctx->d_unshift);
    }
    if (ctx->scratch.ptr) (void)hipFree(ctx->scratch.ptr);
    ctx->api_arena.release();
    ctx->lc_arena.release();
    for (auto &e : ctx->ev) {
        if (e) (void)hipEventDestroy(e);
    }
    for (auto &e : ctx->stage_ev) {
        if (e) (void)hipEventDestroy(e);
    }
    for (auto e : ctx->chunk_ev) (void)hipEventDestroy(e);
    if (ctx->table_ev) (void)hipEventDestroy(ctx->table_ev);
    if (ctx->one_commit.exec) (void)hipGraphExecDestroy(ctx->one_commit.exec);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->out_stream) (void)hipStreamDestroy(ctx->out_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->sha_stream) (void)hipStreamDestroy(ctx->sha_stream);
    for (auto &st : ctx->side_stream) {
        if (st) (void)hipStreamDestroy(st);
    }
    for (auto &h : ctx->h_stage) {
        if (h) (void)hipHostFree(h);
    }
    for (auto &h : ctx->h_out) {
        if (h) (void)hipHostFree(h);
    }
    delete ctx;
}

static void destroy_pool(DevicePool *p) {
    if (!p) return;
    // aliases first, the owner of the shared arrays last
    for (size_t i = p->slots.size(); i-- > 0;) destroy_slot(p->slots[i]);
    (void)hipSetDevice(p->device);
    // every slot is idle now: the published tables and the ones they replaced can go
    for (void *t : {(void *)p->pub.commit.d_table, (void *)p->pub.fk20.d_table, (void *)p->pub.mono.d_table}) {
        if (t) (void)hipFree(t);
    }
    for (void *t : p->retired) (void)hipFree(t);
    delete p;
}

static void destroy_settings(SettingsCtx *sc) {
    if (!sc) return;
    DeviceGuard guard;   // destroy_slot selects each slot's device; the caller's comes back afterwards
    sc->cancel_widening.store(true);   // a background table build stops at its next launch
    if (sc->widener.joinable()) sc->widener.join();
    for (auto *p : sc->pools) {
        if (p && dev::device_wedged(p->device)) {
            // kernels of this context may still be running (or stuck): hipFree / hipStreamDestroy would wait for them.
            // The device state is leaked; the process is expected to report the C_KZG_ERROR it got and restart.
            fprintf(stderr, "[ckzg-hip] free_trusted_setup: device %d stopped answering earlier, its GPU state is left in place\n", p->device);
            return;
        }
    }
    for (auto &c : sc->comb) {
        delete c;   // (page-locked batch buffers)
        c = nullptr;
    }
    for (auto *p : sc->pools) destroy_pool(p);
    delete sc;
}

#define CTX_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            fprintf(stderr, "[ckzg-hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e),   \
                    __FILE__, __LINE__);                                                         \
            return _e == hipErrorOutOfMemory ? C_KZG_MALLOC : C_KZG_ERROR;                       \
        }                                                                                        \
    } while (0)

static C_KZG_RET init_slot_runtime(dev::DeviceCtx *ctx) {
    CTX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CTX_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    ctx->api_arena.stream = ctx->lc_arena.stream = ctx->stream;
    for (auto &e : ctx->ev) CTX_TRY(hipEventCreate(&e));
    return C_KZG_OK;
}

// Tables of one pool, built on the calling thread's device into `ctx` (slot 0).  h_xext (optional):
// host copy of the x_ext_fft columns for the KZGSettings mirror.
namespace {
struct PhaseClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap() {
        auto t1 = std::chrono::steady_clock::now();
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
        return ms;
    }
};
}  // namespace

LoadTimes &pending_load_times() {
    static thread_local LoadTimes t;
    return t;
}

// widths asked for by the options / environment of a load (what build_owner builds, or -- "async_tables" -- what the
// widener is to reach after the load has returned with the default widths)
static void requested_widths(int out[3], const KZGSettings *s, const Options &opts) {
    int cw = env_int("CKZG_HIP_COMMIT_WBITS", opts.commit_wbits);
    if (cw < 4 || cw > 16) cw = 10;
    int fw = env_int("CKZG_HIP_FK20_WBITS", opts.fk20_wbits);
    if (fw == 0) fw = s->wbits > 8 ? (s->wbits > 13 ? 13 : (int)s->wbits) : 8;
    if (fw < 4 || fw > 16) fw = 8;
    int pw = env_int("CKZG_HIP_PROOF_WBITS", opts.proof_wbits);
    if (pw != 0 && (pw < 4 || pw > 16)) pw = 8;
    out[0] = cw;
    out[1] = fw;
    out[2] = pw;
}

// Largest batch that takes the FFT-free direct path (fk20.hip).  Round 5's small-batch FK20 (radix-8 steps, three-wave
// ladders) answers 2..16 blobs in 4.3-5.0 ms whatever the table, so the direct path keeps only what it still wins
// (same box, profiles/r05_fk20_small_ab.txt): one blob always (1.76 ms on a 16-bit table, 2.99 on 8 bits), two blobs on
// tables of >= 13 bits (3.08 ms at 16 bits against 4.3; 5.2 ms on 8 bits).
static int auto_direct_max(int proof_wbits) { return proof_wbits >= 13 ? 2 : 1; }

static C_KZG_RET build_owner(DevicePool *pool, const KZGSettings *s, const Options &opts,
                             const G1Affine *lagrange_brp_affine, const G1Affine *monomial_affine,
                             G1Affine *h_xext, LoadTimes *lt) {
    dev::DeviceCtx *ctx = pool->slots[0];
    int want[3];
    requested_widths(want, s, opts);
    const bool async = env_int("CKZG_HIP_ASYNC_TABLES", opts.async_tables) != 0;
    if (async) {   // start with the library's default footprint; the widener takes it from there
        want[0] = want[0] < 10 ? want[0] : 10;
        want[1] = want[1] < 8 ? want[1] : 8;
        want[2] = want[2] < 8 ? want[2] : 8;
    }
    LoadTimes scratch_times;
    if (!lt) lt = &scratch_times;
    PhaseClock clk;
    // The tables belong to the pool once they are published (destroy_pool frees them); a load that fails before
    // that -- the third table does not fit, say -- must not leave the first two behind.
    struct UnpublishedTables {
        dev::DeviceCtx *c;
        bool published = false;
        ~UnpublishedTables() {
            if (published) return;
            for (dev::FixedBaseTable *t : {&c->commit, &c->fk20, &c->mono}) {
                if (t->d_table) (void)hipFree(t->d_table);
                t->d_table = nullptr;
            }
        }
    } unpublished{ctx};
    CTX_TRY(hipSetDevice(ctx->device));
    C_KZG_RET r = init_slot_runtime(ctx);
    if (r != C_KZG_OK) return r;
    lt->ms[LP_HIP_INIT] += clk.lap();

    // Fr twiddles
    CTX_TRY(hipMalloc(&ctx->d_roots, (dev::N_EXT + 1) * sizeof(Fr)));
    CTX_TRY(hipMalloc(&ctx->d_brp_roots, dev::N_EXT * sizeof(Fr)));
    CTX_TRY(hipMemcpy(ctx->d_roots, s->roots_of_unity, (dev::N_EXT + 1) * sizeof(Fr), hipMemcpyHostToDevice));
    CTX_TRY(hipMemcpy(ctx->d_brp_roots, s->brp_roots_of_unity, dev::N_EXT * sizeof(Fr), hipMemcpyHostToDevice));
    CTX_TRY(hipMalloc(&ctx->d_brp_roots29, (size_t)dev::ROOTS29_ENTRIES * 9 * sizeof(uint32_t)));
    if (dev::roots29_build(ctx, ctx->d_brp_roots29) != 0) return C_KZG_ERROR;
    CTX_TRY(dev::sync_stream(ctx->stream));

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

    struct { void *p; } d_bases;
    CTX_TRY(hipMalloc(&ctx->d_lagr, NUM_G1_POINTS * sizeof(G1Affine)));
    d_bases.p = ctx->d_lagr;
    CTX_TRY(hipMemcpy(d_bases.p, lagrange_brp_affine, NUM_G1_POINTS * sizeof(G1Affine), hipMemcpyHostToDevice));
    CTX_TRY(hipMalloc(&ctx->d_mono, NUM_G1_POINTS * sizeof(G1Affine)));
    CTX_TRY(hipMemcpy(ctx->d_mono, monomial_affine, NUM_G1_POINTS * sizeof(G1Affine), hipMemcpyHostToDevice));

    // The fixed-base tables and the G1 FFT twiddles use the endomorphism phi = [lambda], which holds only
    // on the prime-order subgroup.  The reference checks setup points for curve membership only
    // (setup.c:447-477); a file with a point outside G1 is not a trusted setup, and is rejected here
    // rather than silently giving sums that differ from the reference's.
    {
        DeviceBuffer d_st;
        if (!d_st.alloc(2 * NUM_G1_POINTS)) return C_KZG_MALLOC;
        int rc = dev::subgroup_g1_batch_device(ctx, (uint8_t *)d_st.p, (const G1Affine *)d_bases.p, NUM_G1_POINTS);
        if (!rc) rc = dev::subgroup_g1_batch_device(ctx, (uint8_t *)d_st.p + NUM_G1_POINTS, ctx->d_mono, NUM_G1_POINTS);
        if (rc) return (C_KZG_RET)rc;
        std::vector<uint8_t> st(2 * NUM_G1_POINTS);
        CTX_TRY(dev::sync_stream(ctx->stream));
        CTX_TRY(hipMemcpy(st.data(), d_st.p, st.size(), hipMemcpyDeviceToHost));
        for (uint8_t b : st) {
            if (b) {
                fprintf(stderr, "[ckzg-hip] trusted setup holds a G1 point outside the prime-order subgroup\n");
                return C_KZG_BADARGS;
            }
        }
    }

    lt->ms[LP_SMALL_TABLES] += clk.lap();
    // commitment table over the bit-reversed Lagrange points
    {
        int wbits = fit_wbits("commit", want[0], 8, (int)NUM_G1_POINTS);
        int rc = dev::build_fixed_base_table(ctx, &ctx->commit, (const G1Affine *)d_bases.p, (int)NUM_G1_POINTS, wbits,
                                             &lt->ms[LP_COMMIT_MALLOC]);
        if (rc) return (C_KZG_RET)rc;
    }
    clk.lap();
    // FK20: x_ext_fft columns by 64 G1 FFTs on the GPU, then the fixed-base table over those 8192 points
    {
        int rc = dev::fk20_setup_device(ctx, ctx->d_mono, h_xext);
        if (rc) return (C_KZG_RET)rc;
        lt->ms[LP_FK20_SETUP] += clk.lap();
        int wbits = fit_wbits("fk20", want[1], 8, dev::N_CELLS_EXT * dev::N_CELL);
        rc = dev::build_fixed_base_table(ctx, &ctx->fk20, ctx->d_xext, dev::N_CELLS_EXT * dev::N_CELL, wbits,
                                         &lt->ms[LP_FK20_MALLOC]);
        if (rc) return (C_KZG_RET)rc;
    }
    // table over the monomial points for the low-latency (direct) cell-proof path
    {
        int wbits = want[2];
        ctx->direct_max = env_int("CKZG_HIP_DIRECT_MAX", opts.direct_max);
        if (wbits != 0 && ctx->direct_max != 0) {
            wbits = fit_wbits("proof", wbits, 8, (int)NUM_G1_POINTS);
            int rc = dev::build_fixed_base_table(ctx, &ctx->mono, ctx->d_mono, (int)NUM_G1_POINTS, wbits,
                                                 &lt->ms[LP_PROOF_MALLOC]);
            if (rc) return (C_KZG_RET)rc;
            // Automatic hand-over point (measured, tools/bench_direct_vs_fk20.py, profiles/r02_quad_ab.txt): with the
            // radix-4 / four-lane G1 FFT FK20 costs 6.8-7.1 ms for any batch of up to 8 blobs (it was ~28 ms with
            // twelve dependent one-lane ladder stages), the direct path 1.9 / 2.2 / 3.1 ms for one blob plus
            // ~1.3 / ~1.6 / ~2.3 ms per further blob with a 16 / 13 / 8-bit table.
            if (ctx->direct_max < 0) ctx->direct_max = auto_direct_max(wbits);
        }
        if (ctx->direct_max < 0) ctx->direct_max = 0;
    }
    // publication 1: what every slot of the pool serves calls from
    {
        std::lock_guard<std::mutex> lock(pool->mu);
        pool->pub.commit = ctx->commit;
        pool->pub.mono = ctx->mono;
        pool->pub.fk20 = ctx->fk20;
        pool->pub.direct_max = ctx->direct_max;
        pool->pub.version = 1;
        ctx->tables_version = 1;
        unpublished.published = true;
    }
    return C_KZG_OK;
}

// "async_tables": widen one pool's tables to the requested widths, one table at a time (commitment first), each
// published as soon as it is complete.  Runs on the widener thread with its own stream; a width that does not fit,
// an allocation failure or a cancelled build leaves the narrower table in service.
static void widen_pool(SettingsCtx *sc, DevicePool *pool) {
    if (hipSetDevice(pool->device) != hipSuccess) return;
    dev::DeviceCtx b;   // builder context: a stream and nothing else
    b.device = pool->device;
    if (hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking) != hipSuccess) return;
    const dev::DeviceCtx *owner = pool->slots[0];
    const int user_direct_max = env_int("CKZG_HIP_DIRECT_MAX", sc->opts.direct_max);
    struct Job {
        int which;
        const char *name;
        const G1Affine *bases;
        int npoints, phase;
    } jobs[3] = {{0, "commit", owner->d_lagr, (int)NUM_G1_POINTS, LP_COMMIT_MALLOC},
                 {2, "proof", owner->d_mono, (int)NUM_G1_POINTS, LP_PROOF_MALLOC},
                 {1, "fk20", owner->d_xext, dev::N_CELLS_EXT * dev::N_CELL, LP_FK20_MALLOC}};
    for (const Job &j : jobs) {
        if (sc->cancel_widening.load()) break;
        int have;
        {
            std::lock_guard<std::mutex> lock(pool->mu);
            const dev::FixedBaseTable &cur = j.which == 0 ? pool->pub.commit : (j.which == 1 ? pool->pub.fk20 : pool->pub.mono);
            have = cur.d_table ? cur.wbits : 0;
        }
        if (j.which == 2 && (have == 0 || user_direct_max == 0)) continue;   // the direct path is switched off
        int wbits = fit_wbits(j.name, sc->requested_wbits[j.which], have, j.npoints);
        if (wbits <= have) continue;
        dev::FixedBaseTable t;
        double times[2] = {0, 0};   // allocation, construction: merged into the load's phases under the lock
        int rc = dev::build_fixed_base_table(&b, &t, j.bases, j.npoints, wbits, times, &sc->cancel_widening);
        if (pool == sc->pools[0]) {
            std::lock_guard<std::mutex> lock(sc->widen_mu);
            sc->load.ms[j.phase] += times[0];
            sc->load.ms[j.phase + 1] += times[1];
        }
        if (rc) {
            if (rc != 5) fprintf(stderr, "[ckzg-hip] widening the %s table to %d bits failed (rc %d): the %d-bit table stays\n", j.name, wbits, rc, have);
            (void)hipGetLastError();
            continue;
        }
        std::lock_guard<std::mutex> lock(pool->mu);
        dev::FixedBaseTable &cur = j.which == 0 ? pool->pub.commit : (j.which == 1 ? pool->pub.fk20 : pool->pub.mono);
        if (cur.d_table) pool->retired.push_back(cur.d_table);
        cur = t;
        if (j.which == 2 && user_direct_max < 0) pool->pub.direct_max = auto_direct_max(wbits);
        pool->pub.version++;
    }
    (void)dev::sync_stream(b.stream);
    (void)hipStreamDestroy(b.stream);
}

static void widener_main(SettingsCtx *sc) {
    guarded([&]() -> C_KZG_RET {
        JoinThreads th;   // one builder per device; pools that share a device (replicas) one after the other
        std::vector<int> seen;
        for (DevicePool *first : sc->pools) {
            if (std::find(seen.begin(), seen.end(), first->device) != seen.end()) continue;
            seen.push_back(first->device);
            th.spawn([sc, first]() {
                pin_thread_to_device_numa(first->device);
                for (DevicePool *p : sc->pools) {
                    if (p->device == first->device) widen_pool(sc, p);
                }
            });
        }
        th.join();
        return C_KZG_OK;
    });
    {
        std::lock_guard<std::mutex> lock(sc->widen_mu);
        sc->widening_done = true;
    }
    sc->widen_cv.notify_all();
}

// a further slot of the same pool: own streams, events, scratch; the owner's tables
static C_KZG_RET clone_slot(dev::DeviceCtx **out, const dev::DeviceCtx *owner, int slot) {
    dev::DeviceCtx *c = new dev::DeviceCtx();
    *out = c;
    c->device = owner->device;
    c->slot = slot;
    c->owns_tables = false;
    c->commit = owner->commit;
    c->mono = owner->mono;
    c->fk20 = owner->fk20;
    c->direct_max = owner->direct_max;
    c->tables_version = owner->tables_version;
    c->d_lagr = owner->d_lagr;
    c->d_roots = owner->d_roots;
    c->d_brp_roots = owner->d_brp_roots;
    c->d_brp_roots29 = owner->d_brp_roots29;
    c->d_roots_raw = owner->d_roots_raw;
    c->d_xext = owner->d_xext;
    c->d_mono = owner->d_mono;
    c->d_shift = owner->d_shift;
    c->d_unshift = owner->d_unshift;
    c->host_prepared = owner->host_prepared;
    return init_slot_runtime(c);
}

C_KZG_RET create_settings_ctx(KZGSettings *s, const G1Affine *lagrange_brp_affine, const G1Affine *monomial_affine) {
    const Options opts = options_snapshot();
    DeviceGuard guard;   // the load selects devices on this thread (clone_slot loop); the caller's comes back
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "[ckzg-hip] no HIP device available: this build has no CPU fallback for the MSM/FFT hot path\n");
        return C_KZG_ERROR;
    }
    // device list: the "devices" mask if given, else the single `device`
    std::vector<int> devs;
    int64_t mask = opts.devices;
    if (const char *v = getenv("CKZG_HIP_DEVICES")) {
        if (*v) mask = !strcmp(v, "all") ? -1 : (int64_t)strtoll(v, nullptr, 0);
    }
    if (mask != 0) {
        for (int d = 0; d < ndev && d < 63; d++) {
            if (mask < 0 || ((mask >> d) & 1)) devs.push_back(d);
        }
        if (mask > 0 && (mask >> (ndev < 63 ? ndev : 63)) != 0) {
            fprintf(stderr, "[ckzg-hip] devices mask 0x%llx names a device that is not visible (%d visible)\n",
                    (unsigned long long)mask, ndev);
            return C_KZG_ERROR;
        }
    } else {
        int device = opts.device;
        if (device < 0) device = env_int("CKZG_HIP_DEVICE", -1);
        if (device < 0) device = env_int("LOCAL_RANK", 0) % ndev;
        if (device >= ndev) {
            fprintf(stderr, "[ckzg-hip] device %d out of range (%d visible)\n", device, ndev);
            return C_KZG_ERROR;
        }
        devs.push_back(device);
    }
    if (devs.empty()) return C_KZG_ERROR;
    int replicas = env_int("CKZG_HIP_REPLICAS", opts.replicas);
    if (replicas < 1) replicas = 1;
    if (replicas > 8) replicas = 8;
    int nslots = env_int("CKZG_HIP_STREAMS", opts.streams);
    if (nslots < 1) nslots = 1;
    if (nslots > 64) nslots = 64;

    SettingsCtx *sc = new SettingsCtx();
    sc->opts = opts;
    sc->load = pending_load_times();       // host phases measured by load_trusted_setup(_file) on this thread
    pending_load_times() = LoadTimes();
    PhaseClock slots_clk;
    host::g2_prepare(sc->prepared.gen, host::g2_to_affine(host::g2_generator()));
    host::g2_prepare(sc->prepared.s1, host::g2_to_affine(*as_g2(&s->g2_values_monomial[1])));
    host::g2_prepare(sc->prepared.s64, host::g2_to_affine(*as_g2(&s->g2_values_monomial[dev::N_CELL])));
    (void)host::g1_gen_table();   // built once per process, here rather than inside the first verification
    for (int d : devs) {
        for (int r = 0; r < replicas; r++) {
            DevicePool *p = new DevicePool();
            p->device = d;
            p->owner = sc;
            dev::DeviceCtx *owner = new dev::DeviceCtx();
            owner->device = d;
            owner->host_prepared = &sc->prepared;
            p->slots.push_back(owner);
            sc->pools.push_back(p);
        }
    }
    // the pools are independent: build them concurrently, one host thread per device (pools that share a
    // device -- replicas -- build one after the other so that fit_wbits sees what is really free)
    std::vector<G1Affine> h_xext((size_t)dev::N_CELLS_EXT * dev::N_CELL);
    std::vector<C_KZG_RET> rets(sc->pools.size(), C_KZG_OK);
    {
        JoinThreads th;
        for (size_t di = 0; di < devs.size(); di++) {
            th.spawn([&, di]() {
                if (devs.size() > 1) pin_thread_to_device_numa(devs[di]);
                for (int r = 0; r < replicas; r++) {
                    const size_t pi = di * replicas + r;
                    rets[pi] = guarded([&]() {
                        return build_owner(sc->pools[pi], s, opts, lagrange_brp_affine, monomial_affine,
                                           pi == 0 ? h_xext.data() : nullptr, pi == 0 ? &sc->load : nullptr);
                    });
                    if (rets[pi] != C_KZG_OK) break;
                }
            });
        }
        th.join();
    }
    C_KZG_RET ret = C_KZG_OK;
    for (auto r : rets) ret = worse(ret, r);
    slots_clk.lap();
    for (size_t pi = 0; pi < sc->pools.size() && ret == C_KZG_OK; pi++) {
        DevicePool *p = sc->pools[pi];
        if (hipSetDevice(p->device) != hipSuccess) ret = C_KZG_ERROR;
        for (int k = 1; k < nslots && ret == C_KZG_OK; k++) {
            dev::DeviceCtx *c = nullptr;
            ret = clone_slot(&c, p->slots[0], k);
            p->slots.push_back(c);
        }
        for (int k = (int)p->slots.size(); k-- > 0;) p->free_slots.push_back(k);  // slot 0 is handed out first
    }
    if (ret == C_KZG_OK) {
        // mirror of x_ext_fft_columns in the host struct (setup.c:238-330), Jacobian as in the reference
        s->x_ext_fft_columns = (g1_t **)calloc(dev::N_CELLS_EXT, sizeof(g1_t *));
        if (!s->x_ext_fft_columns) ret = C_KZG_MALLOC;
        for (int j = 0; j < dev::N_CELLS_EXT && ret == C_KZG_OK; j++) {
            s->x_ext_fft_columns[j] = (g1_t *)calloc(dev::N_CELL, sizeof(g1_t));
            if (!s->x_ext_fft_columns[j]) {
                ret = C_KZG_MALLOC;
                break;
            }
            for (int i = 0; i < dev::N_CELL; i++) {
                *as_g1(&s->x_ext_fft_columns[j][i]) = jac_from_affine(h_xext[(size_t)j * dev::N_CELL + i]);
            }
        }
    }
    if (ret != C_KZG_OK) {
        destroy_settings(sc);
        return ret;
    }
    if (opts.coalesce != 0) {
        // Coalescing of concurrent one-unit callers (combiner.hpp).  Units per launch: commitments and blob proofs
        // 256 (one staging chunk of their batch paths), cells / proofs 128 and recovery 64 (the batch paths' latency form:
        // page-locked both ways on one stream; 34 / 17 MB of results per launch), blob verifications 128.  Buffers are allocated by the first
        // batch, never by a caller that finds the device idle.
        int act = opts.coalesce_active;
        act = (act < 1 ? 1 : (act > 8 ? 8 : act)) * (int)sc->pools.size();
        const size_t blob = (size_t)dev::N_BLOB * 32, cells = (size_t)dev::N_CELLS_EXT * 2048, proofs = (size_t)dev::N_CELLS_EXT * 48;
        struct Shape {
            size_t units, in_per, out_per;
        } shape[CB_COUNT] = {{256, blob, 48}, {128, blob, cells + 1}, {128, blob, proofs + 1}, {128, blob, cells + proofs + 1},
                             {256, blob + 48, 48}, {64, cells, cells + proofs}, {128, blob + 96, 1}};
        for (int i = 0; i < CB_COUNT; i++) {
            // a single verification is mostly host work (transcript, pairing) that scales over the callers' own cores:
            // up to one caller per stream slot they stay on their own, beyond that four batches rotate (measured:
            // 8 threads 6.4 k calls/s alone vs 2.6 k coalesced; 128 threads 6.2 k alone vs 35 k coalesced)
            // Up to three concurrent commitment / blob-proof callers also stay on their own: a one-blob launch leaves most
            // of the device idle, and three of them overlap better than batches of one or two taking turns (measured,
            // default tables: 3 threads 8.9 k commitments/s alone vs 5.7 k coalesced; from 6 threads the batches win).
            // Cells / proofs / recovery launches take 4-5 ms for 1..16 units (the G1 transforms of FK20 are latency), so
            // callers that come back together share ONE launch: combiner.hpp "gathering", up to 150 us
            // (profiles/r05_callers_gather_ab.txt).
            const bool verify = i == CB_VERIFY_BLOB, light = i == CB_COMMIT || i == CB_BLOB_PROOF;
            const bool fk20 = !verify && !light;
            sc->comb[i] = new Combiner(shape[i].units, shape[i].units * shape[i].in_per, shape[i].units * shape[i].out_per,
                                       verify ? 2 * act : act, verify ? nslots : (light ? 3 : 0),
                                       fk20 ? (int)dev::ab_knob("CKZG_HIP_COALESCE_GATHER_US", 150) : 0);
        }
    }
    sc->load.ms[LP_SLOTS] += slots_clk.lap();
    {
        std::unique_lock<std::shared_mutex> lock(g_reg_mu);
        registry()[s->roots_of_unity] = sc;
    }
    if (env_int("CKZG_HIP_ASYNC_TABLES", opts.async_tables) != 0) {
        requested_widths(sc->requested_wbits, s, opts);
        sc->widening_done = true;
        for (DevicePool *p : sc->pools) {
            if (sc->requested_wbits[0] > p->pub.commit.wbits || sc->requested_wbits[1] > p->pub.fk20.wbits ||
                (p->pub.mono.d_table && sc->requested_wbits[2] > p->pub.mono.wbits))
                sc->widening_done = false;   // start_widening() has work to do
        }
    }
    return C_KZG_OK;
}

// Second half of an "async_tables" load, called by load_trusted_setup once the new KZGSettings has served its warm-up
// calls: from here on the widener's 100 GB allocations may hold the runtime's allocator for seconds (the kernel driver
// scrubs VRAM another process has just released), and a first call that still had to allocate its own arena or
// pinned staging would wait behind them.
void start_widening(const KZGSettings *s) {
    SettingsCtx *sc = settings_of(s, false);
    if (!sc) return;
    {
        std::lock_guard<std::mutex> lock(sc->widen_mu);
        if (sc->widening_done || sc->widener.joinable()) return;
    }
    try {
        // (the thread gets the SettingsCtx, never the caller's KZGSettings: bindings move that struct around)
        sc->widener = std::thread(widener_main, sc);
    } catch (...) {   // no thread, no widening: the default-width tables stay in service
        std::lock_guard<std::mutex> lock(sc->widen_mu);
        sc->widening_done = true;
    }
}

void wait_for_tables(const KZGSettings *s) {
    SettingsCtx *sc = settings_of(s, false);
    if (!sc) return;
    // (the widener ends by itself: its launches poll `cancel_widening`, its device waits are bounded)
    std::unique_lock<std::mutex> lock(sc->widen_mu);
    while (!sc->widening_done) (void)sc->widen_cv.wait_for(lock, std::chrono::milliseconds(50));
}

bool tables_ready(const KZGSettings *s) {
    SettingsCtx *sc = settings_of(s, false);
    if (!sc) return true;
    std::lock_guard<std::mutex> lock(sc->widen_mu);
    return sc->widening_done;
}

// What the library is waiting for, for a process that has stopped making progress (tests/watchdog.py calls it from
// a signal handler of the stalled child; an application may call it from a watchdog thread).  Takes no lock it could
// wait for: everything is try-locked, and what is held is reported as held.
void debug_dump(int fd) {
    dprintf(fd, "== ckzg_hip_debug_dump: wait deadline %lld ms, waits expired so far %llu, devices that stopped answering: mask 0x%llx\n",
            (long long)dev::wait_deadline_ms(), (unsigned long long)dev::expired_waits_ref().load(),
            (unsigned long long)dev::wedged_devices_ref().load());
    const int64_t now = dev::monotonic_us();
    dev::WaitNoteSlot *notes = dev::wait_notes();
    int waits = 0;
    for (int i = 0; i < dev::WAIT_NOTES; i++) {
        const char *what = notes[i].what.load(std::memory_order_acquire);
        if (!what) continue;
        waits++;
        dprintf(fd, "  thread %ld waits for: %s (%p), %.1f ms so far\n", notes[i].tid.load(std::memory_order_relaxed), what,
                notes[i].obj.load(std::memory_order_relaxed), (double)(now - notes[i].since_us.load(std::memory_order_relaxed)) / 1000.0);
    }
    if (!waits) dprintf(fd, "  no thread is inside a wait of this library\n");
    if (!g_reg_mu.try_lock_shared()) {
        dprintf(fd, "  registry of loaded KZGSettings: locked exclusively (a load or a free is in progress)\n");
        return;
    }
    static const char *const op_names[CB_COUNT] = {"commit", "cells", "proofs", "cells+proofs", "blob_proof", "recover", "verify_blob"};
    for (auto &kv : registry()) {
        SettingsCtx *sc = kv.second;
        dprintf(fd, " KZGSettings %p: %zu table set(s), leases so far %llu\n", kv.first, sc->pools.size(),
                (unsigned long long)sc->lease_seq.load(std::memory_order_relaxed));
        for (DevicePool *p : sc->pools) {
            if (p->mu.try_lock()) {
                dprintf(fd, "  pool on device %d: %zu of %zu stream slots free\n", p->device, p->free_slots.size(), p->slots.size());
                p->mu.unlock();
            } else {
                dprintf(fd, "  pool on device %d: mutex held\n", p->device);
            }
        }
        for (int op = 0; op < (int)CB_COUNT; op++) {
            if (sc->comb[op]) sc->comb[op]->dump(fd, op_names[op]);
        }
    }
    g_reg_mu.unlock_shared();
    dprintf(fd, "== end of ckzg_hip_debug_dump\n");
}

void destroy_settings_ctx(const KZGSettings *s) {
    SettingsCtx *sc = nullptr;
    {
        std::unique_lock<std::shared_mutex> lock(g_reg_mu);
        auto it = registry().find(s->roots_of_unity);
        if (it == registry().end()) return;
        sc = it->second;
        registry().erase(it);
    }
    destroy_settings(sc);
}

}  // namespace api
}  // namespace ckzg
