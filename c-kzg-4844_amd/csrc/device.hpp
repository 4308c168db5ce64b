// device.hpp -- the GPU state that hangs off a loaded KZGSettings, and the launchers the C-ABI layer
// (ckzg_api.hip) calls.  Per load_trusted_setup and per selected device there is one set of immutable
// tables in HBM and a small pool of DeviceCtx "slots": each slot has its own streams, events, scratch
// and arenas and only aliases the tables, so concurrent callers of one KZGSettings (legal in the
// reference: bindings/rust/src/bindings/mod.rs:910-913, bindings/go/main_test.go:953-971) each lease a
// slot and overlap on the GPU instead of serialising (api_common.hpp: DevicePool, Lease).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <sys/syscall.h>
#include <unistd.h>
#include "g1.hpp"

namespace ckzg {
namespace dev {

constexpr int N_BLOB = 4096;       // FIELD_ELEMENTS_PER_BLOB
constexpr int N_EXT = 8192;        // FIELD_ELEMENTS_PER_EXT_BLOB
constexpr int N_CELL = 64;         // FIELD_ELEMENTS_PER_CELL
constexpr int N_CELLS_EXT = 128;   // CELLS_PER_EXT_BLOB

// Fixed-base table over `npoints` bases: entry (w, i, e) = (e+1) * 2^(wbits*w) * P_i, affine,
// Montgomery form, laid out [w][i][e] with e < half = 2^(wbits-1), w < twin.  The table covers only
// 128-bit half-scalars: a scalar k is split as k = k1 + lambda*k2 (balanced GLV, g1_28.hpp:
// glv_split_signed, |k1|, |k2| < 0.68 * 2^127) and phi(P) = (beta*x, y) = [lambda]P turns the k2 half
// into the same table entries with x multiplied by beta -- applied ONCE to the partial sum of all k2
// terms (phi is a homomorphism), never per addition.  So twin = floor(127/wbits)+1 windows are stored
// and each scalar yields nwin = 2*twin signed digits in [-half, half]: windows 0..twin-1 are the k2
// (phi) half, twin..nwin-1 the k1 half.  The MSM is a plain sum of npoints*nwin table entries -- no
// buckets, no doublings, no data-dependent scatter -- at half the memory of a 255-bit table.

// The tuning constants of the product are constants.  The A/B build (-DCKZG_AB: tools/build_variant.sh) reads them
// from the environment instead, once per process, so that tools/ab_*.sh can sweep one constant on one box; the
// measured sweeps that chose the defaults are under profiles/.  No losing variant lives in the product.
#ifdef CKZG_AB
inline long ab_knob(const char *env, long dflt) {
    const char *e = getenv(env);
    return e && *e ? atol(e) : dflt;
}
#else
constexpr long ab_knob(const char *, long dflt) { return dflt; }
#endif

// ---------------------------------------------------------------------------------------------------------------
// Bounded waits.  The reference never waits on anything (src/eip4844/eip4844.c:264-280 is straight-line code; an
// internal failure is C_KZG_ERROR, returned: src/common/ret.h:24-29).  This library waits for the GPU, for pool
// threads and for other callers, and NONE of those waits may be unbounded: a caller of a consensus client must get an
// answer or an error.  Every wait goes through one of the forms below (tests/test_wait_sites.py greps the product for
// the raw ones) and gives up after `wait_deadline_ms` -- option "wait_deadline_ms" / env CKZG_HIP_WAIT_DEADLINE_MS,
// default 30 s: three orders of magnitude above the longest wait of any call the bench or the tests make, so that
// profilers, sanitizers and oversubscribed hosts do not trip it -- with one line on stderr that names what was waited for.
//   * device waits (sync_stream / sync_event): hipStreamSynchronize has no timed form, so the stream is polled
//     (hipStreamQuery): spinning at first, then asleep for 1/32 of the time waited so far (at most 1 ms) per step -- a
//     wait ends within ~3 % of the moment the work did.  A device wait that expires marks the device WEDGED: its
//     kernels may still be running and writing, so the slot is never handed out again, later calls on that device fail
//     at once with C_KZG_ERROR instead of queueing behind it, and free_trusted_setup leaks the device state instead
//     of calling into a runtime that would block.
//   * host waits (futexes, condition variables): api_common.hpp.
// Waits that are in progress are noted in a small table for ckzg_hip_debug_dump.
// ---------------------------------------------------------------------------------------------------------------
inline std::atomic<int64_t> &wait_deadline_ms_ref() {
    static std::atomic<int64_t> v{[]() -> int64_t {
        const char *e = getenv("CKZG_HIP_WAIT_DEADLINE_MS");
        const long x = e && *e ? atol(e) : 0;
        return x >= 1 ? (int64_t)x : (int64_t)30000;
    }()};
    return v;
}
inline int64_t wait_deadline_ms() { return wait_deadline_ms_ref().load(std::memory_order_relaxed); }
inline std::atomic<uint64_t> &wedged_devices_ref() {
    static std::atomic<uint64_t> m{0};
    return m;
}
inline bool device_wedged(int device) {
    return device >= 0 && device < 64 && ((wedged_devices_ref().load(std::memory_order_relaxed) >> device) & 1) != 0;
}
inline std::atomic<uint64_t> &expired_waits_ref() {   // waits that hit their deadline since the library was loaded
    static std::atomic<uint64_t> n{0};
    return n;
}
inline int64_t monotonic_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// what a thread of this library is waiting for right now (diagnostics only: ckzg_hip_debug_dump)
struct WaitNoteSlot {
    std::atomic<const char *> what{nullptr};
    std::atomic<const void *> obj{nullptr};
    std::atomic<int64_t> since_us{0};
    std::atomic<long> tid{0};
};
constexpr int WAIT_NOTES = 256;
inline WaitNoteSlot *wait_notes() {
    static WaitNoteSlot t[WAIT_NOTES];
    return t;
}
struct WaitNote {
    WaitNoteSlot *slot = nullptr;
    int64_t t0;
    const char *what;
    const void *obj;
    WaitNote(const char *w, const void *o) : t0(monotonic_us()), what(w), obj(o) {
        static thread_local long tid = (long)syscall(SYS_gettid);
        WaitNoteSlot *t = wait_notes();
        for (int i = 0, at = (int)(tid % WAIT_NOTES); i < WAIT_NOTES; i++, at = (at + 1) % WAIT_NOTES) {
            const char *none = nullptr;
            if (t[at].what.compare_exchange_strong(none, w, std::memory_order_acq_rel)) {
                t[at].obj.store(o, std::memory_order_relaxed);
                t[at].since_us.store(t0, std::memory_order_relaxed);
                t[at].tid.store(tid, std::memory_order_relaxed);
                slot = &t[at];
                break;
            }
        }
    }
    ~WaitNote() {
        if (slot) slot->what.store(nullptr, std::memory_order_release);
    }
    WaitNote(const WaitNote &) = delete;
    WaitNote &operator=(const WaitNote &) = delete;
    int64_t waited_us() const { return monotonic_us() - t0; }
    // true once the deadline has passed (says so on stderr, once per wait)
    bool expired() {
        if (waited_us() <= wait_deadline_ms() * 1000) return false;
        if (!said) {
            said = true;
            expired_waits_ref().fetch_add(1, std::memory_order_relaxed);
            fprintf(stderr, "[ckzg-hip] wait deadline exceeded (%lld ms): %s (%p) -- giving up with C_KZG_ERROR\n",
                    (long long)wait_deadline_ms(), what, obj);
        }
        return true;
    }
    bool said = false;
};
inline void sleep_us(int64_t us) {
    struct timespec ts = {(time_t)(us / 1000000), (long)(us % 1000000) * 1000L};
    (void)nanosleep(&ts, nullptr);
}

// query() -> hipSuccess / hipErrorNotReady / an error.  spin_us: how long to poll without sleeping (a one-unit call
// that lasts 250 us must not pay a sleep's granularity; a 10-ms batch may).
// Sleeping in steps of 1/32 of the time waited so far ends a wait ~1.6 % late on average (a 1024-blob commitment
// step: +0.16 ms of 10.0, profiles/r06_sync_poll_ab.txt).  Most waits of a thread repeat -- a server thread makes the
// same call over and over -- so the thread remembers how long its last wait of this kind took and polls WITHOUT
// sleeping from 7/8 of that time on, for at most another 3/16 of it: a steady caller's wait ends when the work does, for
// a fifth of a core while it waits; a wait that is nothing like the last one falls back to the steps.
template <class Query>
inline hipError_t bounded_device_wait(Query &&query, const char *what, const void *obj, int64_t spin_us) {
    hipError_t e = query();
    if (e != hipErrorNotReady) return e;
    static thread_local int64_t last_us[2] = {0, 0};   // [0] batch waits, [1] the one-unit latency waits
    int64_t &last = last_us[spin_us > 1000 ? 1 : 0];
    static const int64_t div = ab_knob("CKZG_HIP_SYNC_STEP_DIV", 32);          // (A/B: 0 = never sleep, the floor)
    static const bool predict = ab_knob("CKZG_HIP_SYNC_PREDICT", 1) != 0;
    const int64_t spin_from = predict && last > 400 ? last - last / 8 : INT64_MAX, spin_to = last + last / 16 + 50;
    WaitNote note(what, obj);
    for (;;) {
        e = query();
        if (e != hipErrorNotReady) break;
        const int64_t waited = note.waited_us();
        if (waited < spin_us || (waited >= spin_from && waited <= spin_to)) continue;
        if (note.expired()) {
            int d = -1;
            if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) wedged_devices_ref().fetch_or((uint64_t)1 << d, std::memory_order_relaxed);
            (void)hipGetLastError();
            return hipErrorLaunchTimeOut;
        }
        if (div <= 0) continue;   // A/B only
        int64_t step = waited / div;
        step = step < 20 ? 20 : (step > 1000 ? 1000 : step);
        if (waited < spin_from && waited + step > spin_from) step = spin_from - waited;   // do not sleep into the polling window
        sleep_us(step < 5 ? 5 : step);
    }
    last = note.waited_us();
    (void)hipGetLastError();   // hipErrorNotReady is not an error, but the runtime remembers it as the thread's last one
    return e;
}
inline hipError_t sync_stream(hipStream_t s, int64_t spin_us = 100) {
    return bounded_device_wait([s]() { return hipStreamQuery(s); }, "stream", (const void *)s, spin_us);
}
inline hipError_t sync_event(hipEvent_t ev, int64_t spin_us = 100) {
    return bounded_device_wait([ev]() { return hipEventQuery(ev); }, "event", (const void *)ev, spin_us);
}

struct FixedBaseTable {
    G1Affine *d_table = nullptr;
    int npoints = 0;
    int wbits = 0;
    int nwin = 0;   // digit windows per scalar (both GLV halves)
    int twin = 0;   // table windows = nwin / 2
    size_t half = 0;
    size_t bytes() const { return (size_t)twin * npoints * half * sizeof(G1Affine); }
    static int twin_for(int wbits) { return 127 / wbits + 1; }
};

struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};

// Bump allocator over one persistent HBM block: the verification entry points need a dozen small
// temporaries per call, and hipMalloc/hipFree (which synchronises) cost more than their kernels.
// Reservation hint of the calling thread: "this call serves n units, calls of up to max_units will follow on the same
// slot" (combiner.hpp sets it around a batch launch).  An arena / the scratch area that has to GROW during such a call
// grows to what the largest batch will need -- sizes are linear in the units -- so that a slot pays the synchronising
// hipFree + hipMalloc once, in its first batch, instead of again whenever the batches have crept up (the 35-80 ms worst
// calls of 256 concurrent callers against means of 5-7 ms).  Capped at 1 GB on top of the need; 1.0 outside batches.
inline double &reserve_scale() {
    static thread_local double s = 1.0;
    return s;
}
struct ReserveScale {
    double old;
    ReserveScale(size_t max_units, size_t n) : old(reserve_scale()) {
        const double f = n ? (double)max_units / (double)n : 1.0;
        reserve_scale() = f < 1.0 ? 1.0 : f;
    }
    ~ReserveScale() { reserve_scale() = old; }
    ReserveScale(const ReserveScale &) = delete;
    ReserveScale &operator=(const ReserveScale &) = delete;
};
inline size_t reserve_scaled(size_t bytes) {
    const double f = reserve_scale();
    if (f <= 1.0) return bytes;
    const double want = (double)bytes * f, cap = (double)bytes + (double)((size_t)1 << 30);
    return (size_t)(want < cap ? want : cap);
}
// Room taken ahead of need (the reservation above, the geometric regrowth of arenas and scratch) is a bet on later
// calls: it may only be placed with memory nobody is short of.  With the tables sized to the free HBM at load time
// (238 GB for the wide set) the extra gigabytes of eight slots could otherwise be what a LATER allocation -- another
// slot's scratch, the call-time table of a verification -- fails for (round-5 advisor finding).  So: nothing is taken
// ahead unless it is at most an eighth of the HBM that is free right now; the need itself is never reduced.
inline size_t speculative_bytes(size_t need, size_t ahead) {
    if (ahead <= need) return need;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return need;
    }
    return ahead <= free_b / 8 ? ahead : need;
}

struct Arena {
    void *base = nullptr;
    size_t cap = 0, used = 0;
    hipStream_t stream = nullptr;   // the owning slot's compute stream: host copies of ABuf are ordered on it
    // start a call: make room for `bytes` (plus alignment slack) and forget earlier contents
    bool begin(size_t bytes) {
        bytes += 64 * 256;
        used = 0;
        if (cap >= bytes) return true;
        const size_t old_cap = cap;
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = 0;
        // growth slack: half again for small arenas, at most 256 MB for large ones -- a 4096-blob verification with its
        // call-time table needs 1.65 GB, and half again on top of that put it over the limit above which ArenaTrim
        // gives the block back after every call (a hipMalloc + a synchronising hipFree per call: ~0.5-0.8 ms)
        const size_t slack = (bytes >> 1) < ((size_t)256 << 20) ? (bytes >> 1) : ((size_t)256 << 20);
        size_t want = bytes + slack;
        if (reserve_scaled(bytes) > want && reserve_scaled(bytes) <= ((size_t)3 << 30) &&
            speculative_bytes(want, reserve_scaled(bytes)) > want) {
            // a batch of a coalescing operation: room for the largest batch at once (fall back to the need if it fails)
            if (hipMalloc(&base, reserve_scaled(bytes)) == hipSuccess) {
                cap = reserve_scaled(bytes);
                return true;
            }
            (void)hipGetLastError();
            base = nullptr;
        }
        // an arena that grows AGAIN doubles (by at most 1 GB): see scratch_reserve (msm.hip) -- every regrowth stalls the
        // callers of a coalesced batch for a synchronising hipFree + hipMalloc
        size_t twice = old_cap + (old_cap < ((size_t)1 << 30) ? old_cap : ((size_t)1 << 30));
        if (twice > ((size_t)3 << 30)) twice = (size_t)3 << 30;   // (never past the size above which ArenaTrim gives the block back)
        if (old_cap && want < twice && speculative_bytes(want, twice) > want && hipMalloc(&base, twice) == hipSuccess) {
            cap = twice;
            return true;
        }
        (void)hipGetLastError();
        base = nullptr;
        if (hipMalloc(&base, want) != hipSuccess) {
            // Callers may treat this as "does not fit, do without" (the call-time table of a verification): the
            // runtime keeps the failure as the thread's last error until somebody reads it, and the next
            // hipGetLastError() after a kernel launch must not mistake it for a launch failure.
            (void)hipGetLastError();
            base = nullptr;
            return false;
        }
        cap = want;
        return true;
    }
    template <class T>
    T *get(size_t count) {
        size_t off = (used + 255) & ~(size_t)255, bytes = (count ? count : 1) * sizeof(T);
        if (!base || off + bytes > cap) return nullptr;
        used = off + bytes;
        return reinterpret_cast<T *>(static_cast<uint8_t *>(base) + off);
    }
    void release() {
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = used = 0;
    }
};

struct DeviceCtx {
    int device = 0;
    int slot = 0;                 // index in its pool; slot 0 owns the tables, the others alias them
    bool owns_tables = true;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // host-to-device staging, overlapped with `stream`
    hipStream_t out_stream = nullptr;   // device-to-host draining of results (OutPipe), created on first use
    hipStream_t aux_stream = nullptr;   // call-time table construction under a long copy (verification), created on first use
    // Compute-unit partition for the resident verification (round 6, ckzg_api2.hip: verify_blobs_core): the SHA-256
    // chain is ONE wave per 64 blobs whose speed is its own issue rate -- any wave that shares its SIMD takes issue slots
    // from it -- so it runs on a stream confined to a quarter of the compute units and the work that runs underneath
    // it (point validation, call-time table) on streams confined to the rest.  Created on first use; nullptr where the
    // runtime refuses a masked stream (the call then uses the plain streams).
    hipStream_t sha_stream = nullptr, side_stream[2] = {nullptr, nullptr};
    bool cu_partition_tried = false;
    void *h_stage[2] = {nullptr, nullptr};  // pinned host staging buffers (allocated on first use)
    size_t h_stage_bytes = 0;
    void *h_out[2] = {nullptr, nullptr};    // pinned device-to-host staging (cells+proofs / recover pipelines)
    size_t h_out_bytes = 0;
    FixedBaseTable commit;        // over g1_values_lagrange_brp (4096 points)
    FixedBaseTable mono;          // over g1_values_monomial (4096 points): low-latency cell proofs
    int direct_max = 10;          // batches up to this many blobs use the direct proof path (set at load)
    uint64_t tables_version = 0;  // which publication of the pool's tables the copies above are (api_common.hpp: PublishedTables)
    G1Affine *d_lagr = nullptr;   // g1_values_lagrange_brp, affine, 4096 (bases of the commitment table; kept for widening)
    Scratch scratch;              // per slot: reused by every call that leases the slot
    Arena api_arena, lc_arena;    // temporaries of the host-pointer entry points / of gpu_lincomb_multi
    hipEvent_t stage_ev[4] = {};  // copied[2], consumed[2] of the staging pipeline (created on first use)
    hipEvent_t table_ev = nullptr;      // "the call-time table is complete" (verification; created on first use)
    std::vector<hipEvent_t> chunk_ev;   // per-chunk events of the pipelined verification (grown on demand, kept)
    hipEvent_t ev[12] = {};       // timing events
    // The one-blob blob_to_kzg_commitment as a graph built node by node (copy in, flag reset, recoding, accumulate, fold +
    // finalize, copy out: six dependent nodes launched with ONE submission; msm.hip: commit_one_graph_build), valid while
    // everything its nodes point at stays where it was; rebuilt otherwise (ckzg_api.hip: commit_batch_on).
    struct OneCommitGraph {
        hipGraphExec_t exec = nullptr;
        const void *key[7] = {};
        bool unusable = false;    // construction or instantiation failed once on this slot: plain launches from then on
    } one_commit;
    float last_ms[6] = {-1, -1, -1, -1, -1, -1};  // see ckzg_hip_last_kernel_ms
    // Fr tables for NTTs and evaluation (Montgomery form, 8 x u32)
    Fr *d_roots = nullptr;        // w^i, 8193 entries
    Fr *d_brp_roots = nullptr;    // 8192
    uint32_t *d_brp_roots29 = nullptr;   // the first 4096 of them as fr29.hpp's nine 29-bit limbs, then 1/brp_roots[2m], m < 2048
    uint32_t *d_roots_raw = nullptr;  // GLV halves {k1, k2} of w^(64 i), i <= 128 (twiddles of the G1 FFT)
    // FK20
    FixedBaseTable fk20;          // over x_ext_fft columns: point index = col*64 + row
    G1Affine *d_xext = nullptr;   // [128][64] affine
    G1Affine *d_mono = nullptr;   // g1_values_monomial, affine, 4096
    Fr *d_shift = nullptr;        // 7^i, i < 8192   (coset_fft, fft.c:257-279)
    Fr *d_unshift = nullptr;      // 7^-i, i < 8192  (coset_ifft, fft.c:290-301)
    void *host_prepared = nullptr; // api::PreparedG2 (host-side line tables for the pairing checks; owned by SettingsCtx)
};

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            fprintf(stderr, "[ckzg-hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), \
                    __FILE__, __LINE__);                                                       \
            return _e == hipErrorOutOfMemory ? 3 : 2;                                          \
        }                                                                                      \
    } while (0)

// a device allocation that is freed on every exit path of the function holding it
struct DevTmp {
    void *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
};

// returns 0 ok / 2 error / 3 out of memory (C_KZG_RET values)
int scratch_reserve(DeviceCtx *ctx, size_t bytes);

// Build a fixed-base table for `npoints` affine bases already in HBM.
// times_ms (optional): [0] += allocation, [1] += construction kernels.  cancel (optional): polled between the
// construction launches; a set flag abandons the build (return value 5, nothing left allocated).
int build_fixed_base_table(DeviceCtx *ctx, FixedBaseTable *t, const G1Affine *d_bases, int npoints,
                           int wbits, double *times_ms = nullptr, const std::atomic<bool> *cancel = nullptr);

// Commit n blobs resident in HBM: d_out48[n][48], d_status[n] (0 ok, 1 non-canonical element).
int commit_blobs_device(DeviceCtx *ctx, uint8_t *d_out48, uint8_t *d_status, const uint8_t *d_blobs,
                        size_t n);

// the same without the final wait, for callers that pipeline host-to-device copies against it
size_t commit_scratch_bytes(const DeviceCtx *ctx, size_t n);
int commit_blobs_enqueue(DeviceCtx *ctx, uint8_t *d_out48, uint8_t *d_status, const uint8_t *d_blobs,
                         size_t n);

// the pipelined host-pointer form: accumulate per chunk into d_part8[blob][8], one finalize for the batch
int commit_accumulate8_enqueue(DeviceCtx *ctx, G1XYZZ *d_part8, uint32_t *d_bad, const uint8_t *d_blobs, size_t k);
bool commit_chunk_fits8(const DeviceCtx *ctx, size_t k);
int commit_finalize8_enqueue(DeviceCtx *ctx, uint8_t *d_out48, uint8_t *d_status, const G1XYZZ *d_part8,
                             const uint32_t *d_bad, size_t n);

// Generic: sum_i scalar_i * P_i over the ctx->commit table for n independent scalar vectors that
// are already canonical little-endian 8xu32 integers in HBM ([n][4096][8]); writes n compressed
// points.  Used by compute_kzg_proof (quotient polynomial) and friends.
int msm_commit_table_raw_device(DeviceCtx *ctx, uint8_t *d_out48, const uint32_t *d_scalars, size_t n);

// call-time tables (msm.hip): a fixed-base table over points that arrive with the call, and sums over it
void call_table_geometry(FixedBaseTable *t, int npoints, int wbits);
size_t call_table_bytes(const FixedBaseTable &t);       // (accumulator-form entries: not FixedBaseTable::bytes())
size_t call_table_tmp_bytes(const FixedBaseTable &t);
int call_table_enqueue(hipStream_t stream, FixedBaseTable *t, void *d_table, uint8_t *d_tmp, const G1Affine *d_bases);
size_t table_sums_scratch_bytes(const FixedBaseTable &t, size_t nvec);
int table_sums_enqueue(hipStream_t stream, const FixedBaseTable &t, G1XYZZ *d_sums, const uint32_t *d_scalars, size_t nvec,
                       uint8_t *scratch);

size_t msm_partials_needed(const FixedBaseTable &t, size_t nvec);
int msm_from_digits_device(DeviceCtx *ctx, const FixedBaseTable &t, uint8_t *d_out48,
                           const int16_t *d_digits, G1XYZZ *d_partials, size_t nvec);
int msm_small_vectors_device(DeviceCtx *ctx, const FixedBaseTable &t, G1XYZZ *d_out,
                             const int16_t *d_digits, size_t nvec, uint32_t ppv,
                             uint32_t vecs_per_group);

// ntt.hip
int fr_ntt_batch(DeviceCtx *ctx, Fr *d_data, size_t count, int logn, bool dif, bool inverse,
                 bool scale_by_inv_n);
int bytes_to_fr_batch(DeviceCtx *ctx, Fr *d_out, uint32_t *d_bad, const uint8_t *d_in, size_t total,
                      uint32_t elems_per_unit);
int fr_to_bytes_batch(DeviceCtx *ctx, uint8_t *d_out, const Fr *d_in, size_t total);
int zero_extend_batch(DeviceCtx *ctx, Fr *d_dst, const Fr *d_src, size_t count, uint32_t n_src,
                      uint32_t n_dst);

// fk20.hip
// x_ext_fft columns (setup.c:238-330) from the 4096 monomial points; fills ctx->d_xext
// ([128][64] affine) and, if h_xext != nullptr, copies them to the host.
int fk20_setup_device(DeviceCtx *ctx, const G1Affine *d_monomial, G1Affine *h_xext);
// cells (n*128*2048 B) and/or proofs (n*128*48 B) for n blobs in HBM; either output may be null.
// h_cells (page-locked, or null): in the latency form (n <= 64, both outputs) the cells are also copied there, on the
// slot's second stream right behind the kernels that made them -- underneath the proof kernels instead of after them
// (16.8 MB for 64 blobs: 0.33 ms of the call); the copy has completed when the function returns.
int cells_and_proofs_device(DeviceCtx *ctx, uint8_t *d_cells, uint8_t *d_proofs, uint8_t *d_status,
                            const uint8_t *d_blobs, size_t n, uint8_t *h_cells = nullptr, bool *cells_copied = nullptr);
// the same in enqueue-only stages, for callers that pipeline host copies against them (fk20.hip)
int cells_stage_enqueue(DeviceCtx *ctx, uint8_t *d_cells, Fr *d_poly, Fr *d_ext, uint32_t *d_bad,
                        const uint8_t *d_blobs, size_t k);
bool proofs_use_direct(const DeviceCtx *ctx, size_t n);
size_t proofs_scratch_bytes(const DeviceCtx *ctx, size_t k, bool direct);
int proofs_stage_enqueue(DeviceCtx *ctx, uint8_t *d_proofs, const Fr *d_poly, size_t k, uint8_t *scratch, bool direct);
int bad_to_status_enqueue(DeviceCtx *ctx, uint8_t *d_status, const uint32_t *d_bad, size_t k);
// FK20 proofs from monomial coefficients already on the device ([n][4096] Fr, Montgomery)
int fk20_proofs_device(DeviceCtx *ctx, uint8_t *d_proofs, const Fr *d_poly_monomial, size_t n);
void fk20_collect_times(DeviceCtx *ctx);
// the six nodes of commit_blobs_enqueue(n = 1) between a host input and a host result buffer as an explicitly built,
// instantiated graph (msm.hip says why not a capture); 0 ok, 4 geometry does not apply, 2 HIP error
int commit_one_graph_build(DeviceCtx *ctx, hipGraphExec_t *exec_out, uint8_t *d_out48, uint8_t *d_status,
                           const uint8_t *d_blob, const void *h_in, void *h_res);
// after a synchronised commit_blobs_enqueue: its event pairs -> ctx->last_ms[0..3] (digits, accumulate, finalize, all)
void commit_collect_times(DeviceCtx *ctx);
// verify.hip
// d_out[ROOTS29_ENTRIES][9] <- the evaluation domain in fr29.hpp's form (4096 roots, then the 2048 inverses of the
// even-indexed ones), from ctx->d_brp_roots; on ctx->stream.  Enqueue-only.
constexpr int ROOTS29_ENTRIES = 4096 + 2048;
int roots29_build(DeviceCtx *ctx, uint32_t *d_out);
// d_y[i] <- blob i's polynomial at d_z[i] (evaluate_polynomial_in_evaluation_form, eip4844.c:192-240), from the blobs' bytes
// (n x 131,072, in HBM), conversion and range check folded in: d_bad[i] |= 1 where blob i holds a field element >= r
// (its y is then meaningless).  Enqueue-only, on ctx->stream.
int eval_blob_bytes_batch_device(DeviceCtx *ctx, Fr *d_y, uint32_t *d_bad, const uint8_t *d_blob_bytes, const Fr *d_z, size_t n);
// d_sc[3][2n][8] <- the scalar vectors of a blob batch's three sums over a call-time table of (commitments, proofs),
// from the batch challenge r and the blobs' challenges d_z (verify.hip: k_rlc_scalars).  Enqueue-only.
// the same for a cell batch (verify.hip: k_cell_rlc_scalars, k_commit_weights): d_rp[n] <- r^i (Montgomery),
// d_vec_rp[n][8] <- r^i, d_vec_wrp[n][8] <- r^i * h_k^64, d_vec_w[nc][8] <- per-commitment sums of r^i over the member
// lists (d_grp_start[nc + 1], d_members[n]); on ctx->stream.  Enqueue-only.
int cell_rlc_scalars_enqueue(DeviceCtx *ctx, Fr *d_rp, uint32_t *d_vec_rp, uint32_t *d_vec_wrp, uint32_t *d_vec_w,
                             const uint32_t *d_cell_idx, const uint32_t *d_grp_start, const uint32_t *d_members,
                             const Fr &r, size_t n, size_t nc);
int rlc_scalars_enqueue(hipStream_t stream, uint32_t *d_sc, const Fr *d_z, const Fr &r, size_t n);
int eval_quotient_batch_device(DeviceCtx *ctx, Fr *d_y, uint32_t *d_q_raw, int *d_hit, const Fr *d_poly,
                               const Fr *d_z, size_t n);
// stream: nullptr = the context's compute stream
int validate_g1_batch_device(DeviceCtx *ctx, G1Affine *d_out, uint8_t *d_status, const uint8_t *d_in48,
                             size_t n, hipStream_t stream = nullptr);
// the two halves of the validation as separate launches: curve membership + decompression, then the
// subgroup test on the decompressed points (status 1 = finite point outside G1), so that the test can
// run on a second stream next to the kernels that already consume the points
int decompress_g1_batch_device(DeviceCtx *ctx, G1Affine *d_out, uint8_t *d_status, const uint8_t *d_in48, size_t n,
                               hipStream_t stream = nullptr);
int subgroup_g1_batch_device(DeviceCtx *ctx, uint8_t *d_status, const G1Affine *d_pts, size_t n,
                             hipStream_t stream = nullptr);
// d_off: njobs + 1 words of device scratch
int lincomb_multi_device(DeviceCtx *ctx, G1Affine *d_out, G1XYZZ *d_partials, uint32_t *d_off, const G1Affine *d_pts,
                         const uint32_t *d_scalars, size_t total, const uint32_t *h_part_off, int njobs, bool quad);
// pippenger.hip: the same sums by bucket accumulation (enqueue-only; see bucket_msm_enqueue).  In the product since round 6,
// used on request only (ckzg_hip_g1_lincomb, algo = 2): the library's own sums take the ladders (pippenger.hip says why)
bool bucket_msm_available();
size_t bucket_msm_scratch_bytes(size_t total, int njobs, int wbits);
int bucket_msm_wbits(size_t max_job_terms);
int bucket_msm_enqueue(DeviceCtx *ctx, G1Affine *d_out, const G1Affine *d_pts, const uint32_t *d_scalars, size_t total,
                       const uint32_t *h_job_off, int njobs, int wbits, uint8_t *scratch);
// z_i = hash_to_bls_field(SHA-256(domain | degree | blob_i | commitment_i)) for n blobs in HBM
int sha256_challenges_device(DeviceCtx *ctx, Fr *d_z, const uint8_t *d_blobs, const uint8_t *d_commit48, size_t n,
                             hipStream_t stream = nullptr);   // stream: ctx->stream if null
// d_rows[n][160] <- commitment_i | z_i | y_i | proof_i, the per-blob rows of the batch transcript (eip4844.c:597-680);
// d_pts48: commitments [0, n), proofs [n, 2n), 16-byte aligned.  Enqueue-only, on ctx->stream.
int batch_transcript_rows_device(DeviceCtx *ctx, uint8_t *d_rows, const uint8_t *d_pts48, const Fr *d_z, const Fr *d_y, size_t n);
// cells[b][j] (2048 B each) -> image[b][idx[j]] (128 x 2048 B per row, zero-filled by the caller)
int scatter_cells_device(DeviceCtx *ctx, uint8_t *d_image, const uint8_t *d_cells, const uint32_t *d_idx,
                         uint32_t num_cells, size_t num_rows);
// agg[c][j] = sum over the cells i of column c (CSR: col_start[129], order[n]) of rp[i] * cell_fr[i][j]
int cell_aggregate_device(DeviceCtx *ctx, Fr *d_agg, const Fr *d_cell_fr, const Fr *d_rp, const uint32_t *d_col_start,
                          const uint32_t *d_order, size_t n_cells);
// interpolation-polynomial coefficients summed over the 128 columns, as canonical MSM scalars
int interp_sum_device(DeviceCtx *ctx, Fr *d_interp, const Fr *d_cols);
int fr_mul_inplace_device(DeviceCtx *ctx, Fr *d_a, const Fr *d_b, size_t n, size_t period);
int fr_div_inplace_device(DeviceCtx *ctx, Fr *d_a, const Fr *d_b, size_t n);
// generic helpers
int batch_to_affine_device(DeviceCtx *ctx, G1Affine *d_out, const G1XYZZ *d_in, Fp *d_prefix, size_t n);

}  // namespace dev
}  // namespace ckzg
