// api_common.hpp -- glue shared by the C-ABI translation units: casts between the public ckzg.h
// structs and the internal 32-bit-limb types (identical bytes), the hidden header that ties a
// KZGSettings to its GPU context, small host helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <climits>
#include <thread>
#include <vector>

#include "../../include/ckzg_hip.h"
#include "device.hpp"
#include "host_pairing.hpp"

namespace ckzg {
namespace api {

constexpr size_t NUM_G1_POINTS = 4096;  // src/setup/setup.c:34-43
constexpr size_t NUM_G2_POINTS = 65;

static_assert(sizeof(fr_t) == sizeof(Fr), "fr_t layout");
static_assert(sizeof(g1_t) == sizeof(G1Jac), "g1_t layout");
static_assert(sizeof(g2_t) == sizeof(host::G2Jac), "g2_t layout");
static_assert(sizeof(KZGSettings) == 80, "KZGSettings is ABI (src/setup/settings.h:27-79)");

inline Fr *as_fr(fr_t *p) { return reinterpret_cast<Fr *>(p); }
inline const Fr *as_fr(const fr_t *p) { return reinterpret_cast<const Fr *>(p); }
inline G1Jac *as_g1(g1_t *p) { return reinterpret_cast<G1Jac *>(p); }
inline const G1Jac *as_g1(const g1_t *p) { return reinterpret_cast<const G1Jac *>(p); }
inline const host::G2Jac *as_g2(const g2_t *p) { return reinterpret_cast<const host::G2Jac *>(p); }

// CKZG_HIP_TRACE=1 prints a per-phase wall-clock breakdown of the host-pointer entry points to stderr
struct Trace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    const char *what;
    explicit Trace(const char *w) : on(getenv("CKZG_HIP_TRACE") != nullptr), t0(std::chrono::steady_clock::now()), what(w) {}
    void mark(const char *phase) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ckzg-hip trace] %s: %s %.3f ms\n", what, phase,
                std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

struct Options {
    int device = -1;        // -1: env CKZG_HIP_DEVICE, else LOCAL_RANK, else 0
    int64_t devices = 0;    // bit mask of devices to load on (bit i = device i; -1 = every visible one); 0: `device` only
    int replicas = 1;       // independent pools (own tables) per selected device: exercises the multi-device
                            // fan-out on a one-GPU box; 1 in production
    int streams = 8;        // slots (stream + scratch) per pool = concurrent calls per device
    int commit_wbits = 10;  // env CKZG_HIP_COMMIT_WBITS overrides the default
    int fk20_wbits = 0;     // 0: max(8, precompute); env CKZG_HIP_FK20_WBITS
    int proof_wbits = 8;    // monomial-point table for the direct proof path; 0 disables it
    int direct_max = -1;    // largest batch that takes the direct path (0 disables it; -1: by table width)
    int async_tables = 0;   // 1: load with the default-width tables, widen to the requested widths in the background
    int coalesce = 1;       // 1: concurrent one-unit callers of the ckzg.h functions share batch launches (combiner.hpp)
    int coalesce_active = 2;  // launches of one operation in flight per device before callers start to queue
};
// options of the NEXT load_trusted_setup; snapshotted under a lock when a load starts, so concurrent loads
// with different options do not see each other's half-written state
Options options_snapshot();
// read at call time (ckzg_hip_set_option("gpu_sha_min", n)); 0 = decide by host CPU
extern std::atomic<int> g_gpu_sha_min;
// read at call time (ckzg_hip_set_option("host_threads", n)); 0 = automatic (host_thread_budget)
extern std::atomic<int> g_host_threads;
// read at call time: smallest verify_blob_kzg_proof_batch that takes the pipelined (chunked copy) form; whether the
// verifications may build their call-time table (0: ladder sums, the path a device too full for the table takes)
extern std::atomic<int> g_verify_pipe_min, g_verify_call_table, g_verify_cu_partition;

// How many host threads ONE process of this library may keep busy for a call (challenge hashing, staging copies,
// point decompression at load): the CPUs of the process's affinity mask divided by the processes that share the host.
// Under a one-process-per-GPU launcher every rank runs the same code at the same time on the same cores -- eight
// ranks that each size their pools by the machine would run 8 x 32 hashing threads on whatever the container
// grants -- so the share is cpus / (ranks on this node).  A
// one-process fan-out over several GPUs ("devices") shares ONE process-wide pool between its shards already.
inline int host_thread_budget() {
    const int forced = g_host_threads.load(std::memory_order_relaxed);
    if (forced > 0) return forced;
    static const int automatic = []() {
        int cpus = 0;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = (int)std::thread::hardware_concurrency();
        if (cpus <= 0) cpus = 4;
        // (A cgroup CPU quota is deliberately NOT applied: it is a budget of CPU time per 100 ms period, and a call that
        // hashes 512 MB in a 12 ms burst on 32 threads stays inside a 16-core quota -- capping the pool at 16 threads
        // made the 4096-blob verification 18.1 instead of 12.5 ms on exactly such a box.)
        // ranks on THIS node: only node-local variables are trusted as they are (torchrun, Open MPI, MPICH / Intel MPI /
        // PMI, MVAPICH, Slurm -- whose per-node counts may read "8(x2)": the leading number is taken).  WORLD_SIZE counts
        // the ranks of the whole job -- 64 on an 8-node launch by mpirun / srun, which export no LOCAL_WORLD_SIZE -- so it
        // is clamped to 8: no node of this platform holds more GPUs, and one process per GPU is the shape.  (Round 5
        // clamped it to the devices VISIBLE to the process: a launcher that binds one GPU per rank --
        // ROCR_VISIBLE_DEVICES, srun --gpu-bind -- then made every rank take the whole machine, and a host-only query
        // initialised the HIP runtime from a static initialiser.  No HIP call here.)
        auto positive = [](const char *name) -> int {
            const char *v = getenv(name);
            if (!v || !*v) return 0;
            char *end = nullptr;
            const long x = strtol(v, &end, 10);
            // ("8(x2)", "8,7": Slurm's compressed lists -- the first count; anything else after the digits is garbage)
            if (end == v || (*end != '\0' && *end != '(' && *end != ',')) return 0;
            return (x < 1 || x > 1 << 20) ? 0 : (int)x;
        };
        int ranks = 0;
        for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "PMI_LOCAL_SIZE",
                                 "MV2_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE", "SLURM_STEP_TASKS_PER_NODE",
                                 "SLURM_TASKS_PER_NODE"}) {
            if ((ranks = positive(name)) > 0) break;
        }
        if (ranks == 0 && (ranks = positive("WORLD_SIZE")) > 8) ranks = 8;
        if (ranks < 1) ranks = 1;
        const int share = cpus / ranks;
        return share < 1 ? 1 : share;
    }();
    return automatic;
}

// line tables of the three G2 constants that appear in verification equations
struct PreparedG2 {
    host::G2Prepared gen;     // [1]_2
    host::G2Prepared s1;      // [s]_2      = g2_values_monomial[1]
    host::G2Prepared s64;     // [s^64]_2   = g2_values_monomial[64]
};
inline const PreparedG2 *prepared_of(const dev::DeviceCtx *ctx) {
    return static_cast<const PreparedG2 *>(ctx->host_prepared);
}

// The tables a pool serves calls from.  Immutable once published; a wider table replaces a narrower one by
// publishing a new set (version + 1) under the pool's mutex, and every slot refreshes its copies when it is leased
// -- a lease is exclusive, so a call sees one consistent set from start to end.  Replaced tables are retired, not
// freed: a call that started before the publication may still be reading them.
struct PublishedTables {
    dev::FixedBaseTable commit, mono, fk20;
    int direct_max = 0;
    uint64_t version = 0;
};

// One device's share of a loaded KZGSettings: the tables and the slots that alias them.
struct DevicePool {
    int device = 0;
    PublishedTables pub;            // guarded by mu
    std::vector<void *> retired;    // device allocations of replaced tables, freed with the pool
    std::vector<dev::DeviceCtx *> slots;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> free_slots;
    std::atomic<dev::DeviceCtx *> last{nullptr};  // slot of this pool's most recent lease
    std::atomic<uint64_t> last_seq{0};            // ... and its number in SettingsCtx::lease_seq
    struct SettingsCtx *owner = nullptr;
};

// Everything this library hangs off one KZGSettings.  Found through a registry keyed by the struct's
// roots_of_unity pointer: that pointer survives the by-value copies/moves bindings make of KZGSettings
// (Go embeds it, Rust moves it), and a struct that was not loaded by this library is simply not in the
// registry -- nothing is read through a foreign pointer.
// wall-clock phases of load_trusted_setup (ckzg_hip_load_times), milliseconds, first pool
enum LoadPhase {
    LP_HOST_PARSE = 0,     // hex text -> bytes (load_trusted_setup_file only)
    LP_HOST_POINTS,        // decompression of the 8192 G1 + 65 G2 points, the Lagrange/monomial pairing check, roots
    LP_HIP_INIT,           // device selection, first runtime calls: HIP initialisation + code-object load, streams
    LP_SMALL_TABLES,       // Fr twiddles, coset factors, setup points to HBM, subgroup check of the setup points
    LP_COMMIT_MALLOC,      // hipMalloc of the commitment table (+ its construction scratch)
    LP_COMMIT_BUILD,       // k_window_bases / k_table_chain / k_batch_to_affine of the commitment table
    LP_FK20_SETUP,         // the 64 G1 FFTs of x_ext_fft_columns (setup.c:238-330)
    LP_FK20_MALLOC,
    LP_FK20_BUILD,
    LP_PROOF_MALLOC,
    LP_PROOF_BUILD,
    LP_SLOTS,              // the other slots of the pool (streams, events) + host mirror of x_ext_fft_columns
    LP_COUNT
};
struct LoadTimes {
    double ms[LP_COUNT] = {};
};
// the phases measured before the SettingsCtx exists (load_trusted_setup_file / _impl), per calling thread
LoadTimes &pending_load_times();

// the ckzg.h entry points whose concurrent callers are coalesced (combiner.hpp); the three output forms of
// compute_cells_and_kzg_proofs are separate operations because they run different batch paths
enum CombinedOp { CB_COMMIT = 0, CB_CELLS, CB_PROOFS, CB_CELLS_PROOFS, CB_BLOB_PROOF, CB_RECOVER, CB_VERIFY_BLOB, CB_COUNT };
class Combiner;

struct SettingsCtx {
    LoadTimes load;
    Combiner *comb[CB_COUNT] = {};    // nullptr: "coalesce" = 0
    std::vector<DevicePool *> pools;
    PreparedG2 prepared;
    Options opts;
    std::atomic<unsigned> next{0};
    std::atomic<uint64_t> lease_seq{0};   // leases handed out so far, over all pools (ckzg_hip_last_kernel_ms)
    // "async_tables": the thread that widens the tables after load_trusted_setup has returned
    std::thread widener;
    std::atomic<bool> cancel_widening{false};
    std::mutex widen_mu;              // guards widening_done and, while the widener runs, `load`
    std::condition_variable widen_cv;
    bool widening_done = true;
    int requested_wbits[3] = {0, 0, 0};   // commitment, FK20, proof: what the widener is to reach
};
SettingsCtx *settings_of(const KZGSettings *s, bool complain = true);

// The library never leaves the calling thread's current HIP device changed: whatever selects a device on a
// caller's thread (a lease, a load, a free) holds one of these, which puts the caller's device back on the way
// out -- a host application that shares the process (torch, another HIP library) keeps allocating and launching
// where it was.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
        if (hipGetDevice(&saved) != hipSuccess) {
            (void)hipGetLastError();
            saved = -1;
        }
    }
    ~DeviceGuard() {
        int now = -1;
        if (saved >= 0 && hipGetDevice(&now) == hipSuccess && now != saved) (void)hipSetDevice(saved);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Exclusive use of one slot for the duration of a call.  Default: the first pool with a free slot (round
// robin), waiting if every slot is busy.  Selects the slot's HIP device on the calling thread for the lifetime
// of the lease and restores the caller's device when it ends (`guard` is destroyed after the slot is returned).
struct Lease {
    DeviceGuard guard;
    DevicePool *pool = nullptr;
    dev::DeviceCtx *ctx = nullptr;
    explicit Lease(const KZGSettings *s, int pool_index = -1);
    explicit Lease(DevicePool *p);
    ~Lease();
    Lease(const Lease &) = delete;
    Lease &operator=(const Lease &) = delete;

   private:
    void take(DevicePool *p);
};
// the pool whose device owns ALL of these device pointers (null entries skipped), for the *_device entry points;
// -1 if one of them is not device memory, they live on different devices, or that device holds no tables of `sc`
int pool_of_pointers(const SettingsCtx *sc, const void *const *ptrs, int count);

// implemented in device_ctx.hip
C_KZG_RET create_settings_ctx(KZGSettings *s, const G1Affine *lagrange_brp_affine,
                              const G1Affine *monomial_affine);
void destroy_settings_ctx(const KZGSettings *s);
void debug_dump(int fd);                     // ckzg_hip_debug_dump
void start_widening(const KZGSettings *s);   // "async_tables": after the warm-up calls of the load
// blocks until the background widening of an "async_tables" load has finished (returns at once otherwise)
void wait_for_tables(const KZGSettings *s);
bool tables_ready(const KZGSettings *s);

// (Round 4 kept a process-wide shared lock here -- every entry held it, a stream capture needed it exclusively --
// because a capture on this runtime is invalidated by other threads' HIP calls.  The one graph the library uses is now
// built node by node (msm.hip: commit_one_graph_build), which involves no capture: the lock and its quiet section are gone.)

// No C++ exception may cross the C ABI (the reference returns C_KZG_MALLOC where these would throw).
template <class F>
C_KZG_RET guarded(F &&f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return C_KZG_MALLOC;
    } catch (...) {
        return C_KZG_ERROR;
    }
}

inline C_KZG_RET worse(C_KZG_RET a, C_KZG_RET b) { return (int)a > (int)b ? a : b; }

// Threads that are joined on every exit path: if constructing a later std::thread throws (EAGAIN, rlimit), the
// ones already running are joined by the destructor instead of std::terminate() firing on a joinable thread, and
// the exception reaches guarded() like any other.
struct JoinThreads {
    std::vector<std::thread> th;
    template <class F>
    void spawn(F &&f) {
        th.emplace_back(std::forward<F>(f));
    }
    void join() {
        for (auto &t : th) {
            if (t.joinable()) t.join();
        }
        th.clear();
    }
    ~JoinThreads() { join(); }
};

// A fan-out thread (one per device: staging copies, transcript hashing, its slot's pinned buffers on first touch)
// runs on the NUMA node its GPU hangs off: sysfs gives the node of the PCI function and the node's CPU list.
// Only threads the library creates itself are pinned, never the caller's.  CKZG_HIP_NUMA_PIN=0 switches it off.
inline void pin_thread_to_device_numa(int device) {
    static const bool enabled = []() {
        const char *e = getenv("CKZG_HIP_NUMA_PIN");
        return !(e && *e == '0');
    }();
    if (!enabled) return;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, device) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (char *c = bdf; *c; c++) {
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    }
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    if (node < 0) return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return;
    // never leave the CPUs the process was given (taskset, cgroup cpusets, a launcher's per-rank binding)
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) {
        fclose(f);
        return;
    }
    cpu_set_t set;
    CPU_ZERO(&set);
    int lo, hi, count = 0;
    while (fscanf(f, "%d", &lo) == 1) {
        hi = lo;
        int ch = fgetc(f);
        if (ch == '-') {
            if (fscanf(f, "%d", &hi) != 1) break;
            ch = fgetc(f);
        }
        for (int c = lo; c <= hi && c < CPU_SETSIZE; c++) {
            if (!CPU_ISSET(c, &allowed)) continue;
            CPU_SET(c, &set);
            count++;
        }
        if (ch != ',') break;
    }
    fclose(f);
    if (count > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

// Host-pointer batch entry points: contiguous ranges of the n units over the pools (one host thread and
// one leased slot per device), results written in place by each shard; the reference's equivalent is the
// goroutine fan-out of bindings/go/main_test.go:953-971.  body(ctx, lo, hi) -> C_KZG_RET.
template <class F>
C_KZG_RET for_each_device_shard(const KZGSettings *s, uint64_t n, uint64_t min_shard, F &&body) {
    SettingsCtx *sc = settings_of(s);
    if (!sc) return C_KZG_ERROR;
    size_t np = sc->pools.size();
    if (min_shard == 0) min_shard = 1;
    if (np > 1 && n / min_shard < np) np = (size_t)(n / min_shard);
    if (np <= 1) {
        Lease lease(s);
        if (!lease.ctx) return C_KZG_ERROR;
        return body(lease.ctx, (uint64_t)0, n);
    }
    std::vector<C_KZG_RET> rets(np, C_KZG_OK);
    JoinThreads th;   // joined on every exit path, also when a later thread cannot be created
    const uint64_t base = n / np, extra = n % np;
    uint64_t lo = 0;
    for (size_t d = 0; d < np; d++) {
        const uint64_t hi = lo + base + (d < extra ? 1 : 0);
        th.spawn([&, d, lo, hi]() {
            rets[d] = guarded([&]() -> C_KZG_RET {
                pin_thread_to_device_numa(sc->pools[d]->device);
                Lease lease(sc->pools[d]);
                if (!lease.ctx) return C_KZG_ERROR;
                return body(lease.ctx, lo, hi);
            });
        });
        lo = hi;
    }
    th.join();
    C_KZG_RET ret = C_KZG_OK;
    for (auto r : rets) ret = worse(ret, r);
    return ret;
}

// Sleeping on a 32-bit word instead of spinning on it (the combiner's batches, the transcript hasher of a pipelined
// verification, the waits for pool jobs): a futex wait returns on a wake, a changed value or a signal -- callers
// re-check in a loop; the happens-before edge is the acquire load / release store of the word itself.
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");
// relative timeout: returns on a wake, a changed value, a signal or after `ns` nanoseconds -- there is no form
// without one (device.hpp: bounded waits); the callers re-check their word in a loop
inline void futex_wait_for(std::atomic<uint32_t> *w, uint32_t expected, long ns) {
    struct timespec ts = {ns / 1000000000L, ns % 1000000000L};
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t> *w, int count) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, count, nullptr, nullptr, 0);
}
// Two kinds of host waits:
//  (1) for ANOTHER CALLER's progress or for the device (a batch of the combiner, a stream slot, a published chunk):
//      wait_word_until -- sleeps in slices of at most 50 ms (a lost wake-up costs a slice, not the call), gives up at
//      the deadline and returns false; the caller returns C_KZG_ERROR.
//  (2) for HOST WORK OF THIS CALL that running threads of this library are doing with pointers into the caller's
//      buffers or this frame (staging-copy parts, hashing jobs): the call cannot return before they have let go, and the
//      work is finite by construction (memcpy, SHA-256 over bytes that are there), so wait_host_work_done keeps waiting
//      -- in slices as well, and it says so on stderr once the deadline has passed, so that a stall is never silent.
constexpr long WAIT_SLICE_NS = 50L * 1000 * 1000;
// until pred(value of *w) holds; false: deadline (the word is left as it is)
template <class Pred>
inline bool wait_word_until(std::atomic<uint32_t> *w, Pred &&pred, const char *what) {
    uint32_t v = w->load(std::memory_order_acquire);
    if (pred(v)) return true;
    dev::WaitNote note(what, w);
    for (;;) {
        futex_wait_for(w, v, WAIT_SLICE_NS);
        v = w->load(std::memory_order_acquire);
        if (pred(v)) return true;
        if (note.expired()) return false;
    }
}
// block until *w == 0 (the countdown of outstanding pool jobs of a call); the job that takes it to 0 calls futex_wake
inline void wait_host_work_done(std::atomic<uint32_t> *w, const char *what = "host jobs of this call") {
    uint32_t v = w->load(std::memory_order_acquire);
    if (v == 0) return;
    dev::WaitNote note(what, w);
    bool said = false;
    for (; (v = w->load(std::memory_order_acquire)) != 0;) {
        futex_wait_for(w, v, WAIT_SLICE_NS);
        if (!said && note.waited_us() > dev::wait_deadline_ms() * 1000) {
            said = true;
            fprintf(stderr, "[ckzg-hip] %s: still %u outstanding after %lld ms (host work holding this call's buffers; "
                            "the call cannot return before it has finished)\n", what, v, (long long)dev::wait_deadline_ms());
        }
    }
}
// The idle wait of a SERVICE thread (pool workers parked until there is a job, a result drain parked until its owner
// pushes or closes): nobody's call is waiting on it, it is meant to last as long as nothing is asked of the thread.
template <class CV, class Lock, class Pred>
inline void park_until(CV &cv, Lock &lock, Pred &&pred) {
    while (!pred()) (void)cv.wait_for(lock, std::chrono::seconds(1));
}
// condition variable, same rules as wait_word_until; false: deadline
template <class CV, class Lock, class Pred>
inline bool cv_wait_bounded(CV &cv, Lock &lock, Pred &&pred, const char *what, const void *obj = nullptr) {
    if (pred()) return true;
    dev::WaitNote note(what, obj);
    for (;;) {
        (void)cv.wait_for(lock, std::chrono::nanoseconds(WAIT_SLICE_NS));
        if (pred()) return true;
        if (note.expired()) return false;
    }
}

// pageable <-> pinned staging copy.  One core moves ~10 GB/s, which would make a copy (13 ms per
// 1024 blobs) longer than the kernels it is supposed to hide behind: large chunks are split eight ways, seven
// parts going to a small process-wide set of persistent helper threads (started on first use, never per chunk;
// if they cannot be started the caller copies everything itself).
class CopyHelpers {
   public:
    static CopyHelpers &get() {
        static CopyHelpers *h = new CopyHelpers();   // leaked on purpose: helpers may outlive static destruction
        return *h;
    }
    struct Job {
        void *dst;
        const void *src;
        size_t len;
        std::atomic<uint32_t> *pending;
    };
    // false: no helper available (or no memory for the queue entry), the caller does this part itself.  Never throws:
    // staged_copy has counted the part as pending and has parts in flight that point at its stack.
    bool submit(const Job &j) noexcept {
        try {
            std::lock_guard<std::mutex> lock(mu);
            if (!ensure_started()) return false;
            q.push_back(j);
        } catch (...) {
            return false;
        }
        cv.notify_one();
        return true;
    }

   private:
    static constexpr int NHELP = 14;  // two concurrent callers are served at full width (staged_copy: up to 8 ways)
    bool ensure_started() {
        if (started) return nworkers > 0;
        started = true;
        const int budget = host_thread_budget() - 1;   // (the caller copies a part itself)
        for (int i = 0; i < NHELP && i < budget; i++) {
            try {
                std::thread([this]() { run(); }).detach();
                nworkers++;
            } catch (...) {
                break;
            }
        }
        return nworkers > 0;
    }
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(mu);
                park_until(cv, lock, [&]() { return !q.empty(); });
                j = q.front();
                q.pop_front();
            }
            if (j.len) memcpy(j.dst, j.src, j.len);
            std::atomic<uint32_t> *p = j.pending;   // (the submitter's stack: not touched after the count reaches zero)
            if (p->fetch_sub(1, std::memory_order_acq_rel) == 1) futex_wake(p, INT_MAX);
        }
    }
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool started = false;
    int nworkers = 0;
};

// Persistent worker threads for the per-call host work that used to start its own (the challenge hashers of a
// pipelined verification: 32 thread creations before the first byte moved and 32 joins after the last -- 0.3-0.5 ms
// and 0.15 ms of a 12 ms call).  Same life cycle as CopyHelpers: started on first use, parked on a condition variable,
// leaked at exit.  A job is any callable; completion is the submitter's business (an atomic it counts down).
class WorkerPool {
   public:
    static WorkerPool &get() {
        static WorkerPool *p = new WorkerPool();
        return *p;
    }
    // false: the job was NOT taken (no worker could be started, or no memory for the queue entry) and the caller
    // runs it some other way.  Never throws: the submitters keep counters of outstanding jobs that an exception
    // between the increment and the hand-over would leave wrong for ever.
    bool submit(std::function<void()> job) noexcept {
        try {
            std::lock_guard<std::mutex> lock(mu);
            if (!ensure_started()) return false;
            q.push_back(std::move(job));
        } catch (...) {
            return false;
        }
        cv.notify_one();
        return true;
    }
    int workers() {
        std::lock_guard<std::mutex> lock(mu);
        return ensure_started() ? nworkers : 0;
    }

   private:
    bool ensure_started() {
        if (started) return nworkers > 0;
        started = true;
        int want = host_thread_budget();   // this process's share of the host, not the machine
        if (want > 256) want = 256;        // eight concurrent verifications at full width (a fan-out over 8 GPUs in one process)
        for (int i = 0; i < want; i++) {
            try {
                std::thread([this]() { run(); }).detach();
                nworkers++;
            } catch (...) {
                break;
            }
        }
        return nworkers > 0;
    }
    void run() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lock(mu);
                park_until(cv, lock, [&]() { return !q.empty(); });
                job = std::move(q.front());
                q.pop_front();
            }
            job();
        }
    }
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool started = false;
    int nworkers = 0;
};

inline void staged_copy(void *dst, const void *src, size_t bytes) {
    static const size_t nt = []() {
        // ways a staging copy is split (the caller is one of them).  Measured (profiles/r03_copy_threads_ab.txt): 4 ways make
        // the pageable source of a 4096-blob verification memcpy-bound (median 15.8 ms), 8 ways DMA-bound (12.5 ms)
        long v = dev::ab_knob("CKZG_HIP_COPY_THREADS", 8);
        const long budget = host_thread_budget();
        if (v > budget) v = budget;
        return (size_t)(v < 1 ? 1 : (v > 8 ? 8 : v));
    }();
    if (nt == 1 || bytes < ((size_t)4 << 20)) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t part = (bytes / nt + 4095) & ~(size_t)4095;
    std::atomic<uint32_t> pending{0};
    for (size_t t = 1; t < nt; t++) {
        size_t o = t * part, len = o >= bytes ? 0 : (bytes - o < part ? bytes - o : part);
        if (!len) continue;
        pending.fetch_add(1, std::memory_order_relaxed);
        if (!CopyHelpers::get().submit({(uint8_t *)dst + o, (const uint8_t *)src + o, len, &pending})) {
            pending.fetch_sub(1, std::memory_order_relaxed);
            memcpy((uint8_t *)dst + o, (const uint8_t *)src + o, len);
        }
    }
    memcpy(dst, src, part < bytes ? part : bytes);
    wait_host_work_done(&pending, "staging-copy parts");
}

// The wait that ends a ONE-unit call.  hipStreamSynchronize parks the thread on an interrupt, and being woken costs
// 15-30 us of a 250 us call; polling the stream for the few hundred microseconds such a call lasts costs a fraction
// of one core (the combiner lets at most `coalesce_active` callers per operation get here at a time).  After 2 ms the
// call is not a latency call any more and the thread sleeps between polls (device.hpp: bounded_device_wait).
inline hipError_t wait_stream_low_latency(hipStream_t stream) { return dev::sync_stream(stream, 2000); }

// true if the caller's host buffer is page-locked (hipHostMalloc / hipHostRegister): such a buffer is DMA'd
// from directly, chunk by chunk, without the staging copy a pageable one needs
inline bool host_pointer_is_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'd pointer is "invalid value" to the runtime
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// ThreadSanitizer builds (make tsan): the HIP runtime maps pinned host memory behind the tool's back, often at the
// address of a finished helper thread's stack; telling the tool that the range is new memory keeps it from pairing our
// first accesses with that dead thread's writes.
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
extern "C" void AnnotateNewMemory(const char *file, int line, const volatile void *mem, long size);
#define CKZG_TSAN_NEW_MEMORY(p, n) AnnotateNewMemory(__FILE__, __LINE__, (p), (long)(n))
#endif
#endif
#ifndef CKZG_TSAN_NEW_MEMORY
#define CKZG_TSAN_NEW_MEMORY(p, n) ((void)0)
#endif

// pinned staging of a slot, grown on demand: h_stage[2] (towards the device), h_out[2] (back)
inline bool ensure_pinned(void **bufs, size_t &have, size_t want) {
    if (have >= want) return true;
    for (int i = 0; i < 2; i++) {
        if (bufs[i]) (void)hipHostFree(bufs[i]);
        bufs[i] = nullptr;
    }
    have = 0;
    for (int i = 0; i < 2; i++) {
        if (hipHostMalloc(&bufs[i], want, hipHostMallocDefault) != hipSuccess) {
            bufs[i] = nullptr;
            (void)hipGetLastError();
            return false;
        }
        CKZG_TSAN_NEW_MEMORY(bufs[i], want);
    }
    have = want;
    return true;
}

// Drains results into the caller's (pageable) memory while the compute stream keeps working: the batch
// entry points push (device source, host destination) pairs as soon as the producing kernels are enqueued;
// a helper thread waits for each producer, DMAs the bytes into a pinned double buffer on the slot's
// out_stream and copies them out.  A pageable hipMemcpy would block the enqueueing thread instead, and the
// 268 KB per blob of compute_cells_and_kzg_proofs would serialise behind -- not under -- the proof kernels.
class OutPipe {
   public:
    static constexpr size_t PIECE = (size_t)32 << 20;
    explicit OutPipe(dev::DeviceCtx *c) : ctx(c) {}
    OutPipe(const OutPipe &) = delete;
    OutPipe &operator=(const OutPipe &) = delete;
    ~OutPipe() { (void)finish(); }
    // the bytes at d_src are final once everything enqueued on ctx->stream so far has run
    bool push(const void *d_src, void *h_dst, size_t bytes) {
        if (!bytes) return true;
        if (!started) {
            if (!ctx->out_stream && hipStreamCreateWithFlags(&ctx->out_stream, hipStreamNonBlocking) != hipSuccess) return false;
            if (!ensure_pinned(ctx->h_out, ctx->h_out_bytes, PIECE)) return false;
            for (auto &e : piece_ev) {
                if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                    e = nullptr;
                    return false;
                }
            }
            worker = std::thread([this]() { run(); });
            started = true;
        }
        Item it{nullptr, d_src, h_dst, bytes};
        if (hipEventCreateWithFlags(&it.ready, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventRecord(it.ready, ctx->stream) != hipSuccess) {
            (void)hipEventDestroy(it.ready);
            return false;
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            q.push_back(it);
            pushed++;
        }
        cv.notify_all();
        return true;
    }
    size_t pushed_count() const { return pushed; }
    // blocks until the first `count` pushed items have reached host memory (or the drain has failed: finish() says so;
    // the worker's own waits for the device are bounded, so a drain that cannot go on fails instead of standing still)
    void wait_for(size_t count) {
        std::unique_lock<std::mutex> lock(mu);
        if (!cv_wait_bounded(cv, lock, [&]() { return drained >= count || failed; }, "result drain", this)) failed = true;
    }
    C_KZG_RET finish() {
        if (started) {
            {
                std::lock_guard<std::mutex> lock(mu);
                closing = true;
            }
            cv.notify_all();
            worker.join();
            started = false;
        }
        // also after a push() that failed half-way through its set-up (events created, worker never started)
        for (auto &e : piece_ev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        return failed ? C_KZG_ERROR : C_KZG_OK;
    }

   private:
    struct Item {
        hipEvent_t ready;
        const void *src;
        void *dst;
        size_t bytes;
    };
    void run() {
        bool ok = hipSetDevice(ctx->device) == hipSuccess;
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lock(mu);
                park_until(cv, lock, [&]() { return !q.empty() || closing; });
                if (q.empty()) break;
                it = q.front();
                q.pop_front();
            }
            ok = ok && dev::sync_event(it.ready) == hipSuccess;
            (void)hipEventDestroy(it.ready);
            // pieces through the pinned double buffer: the DMA of piece i+1 runs under the host copy of piece i
            const size_t np = (it.bytes + PIECE - 1) / PIECE;
            auto issue = [&](size_t i) {
                const size_t off = i * PIECE, len = it.bytes - off < PIECE ? it.bytes - off : PIECE;
                ok = ok && hipMemcpyAsync(ctx->h_out[i & 1], (const uint8_t *)it.src + off, len, hipMemcpyDeviceToHost,
                                          ctx->out_stream) == hipSuccess;
                ok = ok && hipEventRecord(piece_ev[i & 1], ctx->out_stream) == hipSuccess;
            };
            if (ok) issue(0);
            for (size_t i = 0; i < np && ok; i++) {
                if (i + 1 < np) issue(i + 1);
                ok = ok && dev::sync_event(piece_ev[i & 1]) == hipSuccess;
                const size_t off = i * PIECE, len = it.bytes - off < PIECE ? it.bytes - off : PIECE;
                if (ok) staged_copy((uint8_t *)it.dst + off, ctx->h_out[i & 1], len);
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                drained++;
                if (!ok) failed = true;
            }
            cv.notify_all();
        }
        if (!ok) {
            std::lock_guard<std::mutex> lock(mu);
            failed = true;
            cv.notify_all();
        }
    }
    dev::DeviceCtx *ctx;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Item> q;
    hipEvent_t piece_ev[2] = {nullptr, nullptr};
    size_t pushed = 0, drained = 0;
    bool started = false, closing = false, failed = false;
};

struct DeviceBuffer {
    void *p = nullptr;
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
};

// the same interface over a slice of a persistent arena (device.hpp: Arena): no hipMalloc/hipFree
// up() / down() are ordered on the slot's compute stream (Arena::stream) and return when the copy is complete:
// every slot stream is hipStreamNonBlocking, so a legacy-stream hipMemcpy would NOT wait for kernels enqueued
// there -- a down() placed after an enqueue-only stage would read stale bytes without any error.
template <class T>
struct ABuf {
    T *p;
    hipStream_t stream;
    ABuf(dev::Arena &a, size_t count) : p(a.get<T>(count)), stream(a.stream) {}
    bool up(const T *h, size_t count) {
        return hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, stream) == hipSuccess &&
               dev::sync_stream(stream) == hipSuccess;
    }
    bool down(T *h, size_t count) const {
        return hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, stream) == hipSuccess &&
               dev::sync_stream(stream) == hipSuccess;
    }
};
using dev::Arena;

// a very large batch must not pin its temporaries in HBM for ever: blocks above 2 GiB are returned
// when the call ends (one hipFree per such call), smaller ones stay for the next call
struct ArenaTrim {
    Arena &a;
    explicit ArenaTrim(Arena &ar) : a(ar) {}
    ~ArenaTrim() {
        if (a.cap > ((size_t)3 << 30)) a.release();
    }
};

// src/common/utils.c:103-140 (n must be a power of two)
inline void bit_reversal_permutation(void *values, size_t size, size_t n) {
    if (n < 2) return;
    unsigned bits = 0;
    while (((size_t)1 << bits) < n) bits++;
    uint8_t *v = static_cast<uint8_t *>(values);
    std::vector<uint8_t> tmp(size);
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (unsigned b = 0; b < bits; b++) j |= ((i >> b) & 1) << (bits - 1 - b);
        if (j > i) {
            memcpy(tmp.data(), v + i * size, size);
            memcpy(v + i * size, v + j * size, size);
            memcpy(v + j * size, tmp.data(), size);
        }
    }
}

inline size_t reverse_bits_limited(size_t n, size_t v) {
    unsigned bits = 0;
    while (((size_t)1 << bits) < n) bits++;
    size_t j = 0;
    for (unsigned b = 0; b < bits; b++) j |= ((v >> b) & 1) << (bits - 1 - b);
    return j;
}

inline void be_to_raw8(uint32_t raw[8], const uint8_t *b) {
    for (int i = 0; i < 8; i++) {
        const uint8_t *p = b + 4 * (7 - i);
        raw[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
}

// src/common/bytes.c:64-70
inline bool fr_from_bytes_canonical(Fr &out, const uint8_t *b) {
    uint32_t raw[8], m[8];
    be_to_raw8(raw, b);
    mod_limbs<FrParams>(m);
    if (limbs_geq<8>(raw, m)) return false;
    out = from_raw<FrParams>(raw);
    return true;
}

// src/common/bytes.c:123-127 (hash_to_bls_field: reduce, never reject)
inline Fr fr_from_bytes_reduce(const uint8_t *b) {
    uint32_t raw[8];
    be_to_raw8(raw, b);
    return from_raw<FrParams>(raw);
}

// src/common/bytes.c:52-56
inline void fr_to_bytes(uint8_t *out, const Fr &a) {
    uint32_t raw[8];
    to_raw<FrParams>(raw, a);
    for (int i = 0; i < 8; i++) {
        uint32_t v = raw[7 - i];
        out[4 * i] = (uint8_t)(v >> 24);
        out[4 * i + 1] = (uint8_t)(v >> 16);
        out[4 * i + 2] = (uint8_t)(v >> 8);
        out[4 * i + 3] = (uint8_t)v;
    }
}

inline void be64(uint8_t out[8], uint64_t v) {
    for (int i = 7; i >= 0; i--) {
        out[i] = (uint8_t)v;
        v >>= 8;
    }
}

// src/common/bytes.c:81-95: decompress, accept infinity, otherwise require the prime-order subgroup
inline C_KZG_RET validate_kzg_g1(G1Jac &out, const uint8_t *b) {
    G1Affine a;
    if (g1_uncompress(a, b) != 0) return C_KZG_BADARGS;
    out = jac_from_affine(a);
    return host::g1_in_subgroup_host(a) ? C_KZG_OK : C_KZG_BADARGS;
}

}  // namespace api
}  // namespace ckzg
