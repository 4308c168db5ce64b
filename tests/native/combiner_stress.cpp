// combiner_stress.cpp -- the combiner's state machine (csrc/combiner.hpp) WITHOUT a GPU: N threads submit units whose
// "batch path" is a host function, with plain malloc for the batch buffers.  What is checked is the protocol: every
// caller gets the result of ITS OWN input, a flagged unit fails its own caller only, a RETRY_SOLO unit runs alone
// afterwards, keys never mix in one launch, nothing hangs.  Built by tests/test_combiner_cpu.py with the host compiler
// of hipcc, once plain and once under -fsanitize=thread (the lock-free parts: batch state words, reference counts,
// the copied counter).  TEST INFRASTRUCTURE; exit code 0 = ok.
#include <cstdlib>
#include <atomic>
#include <cstdint>
#include <unistd.h>
static std::atomic<long> g_allocs_left{1L << 40};   // the scenario "the page-locked pool runs dry" counts this down
static std::atomic<int> g_alloc_us{0};              // a page-locked allocation takes milliseconds (hipHostMalloc: 10-30 ms for 33 MB)
static bool test_alloc(uint8_t **pp, size_t bytes) {
    if (g_alloc_us.load()) usleep(g_alloc_us.load());
    if (g_allocs_left.fetch_sub(1) <= 0) return false;
    return (*pp = static_cast<uint8_t *>(malloc(bytes))) != nullptr;
}
#define CKZG_COMBINER_PINNED_ALLOC(pp, bytes) test_alloc((pp), (bytes))
#define CKZG_COMBINER_PINNED_FREE(p) free(p)
#include "../../c-kzg-4844_amd/csrc/combiner.hpp"

#include <cstdio>
#include <random>
#include <unistd.h>

using namespace ckzg::api;
namespace dev = ckzg::dev;

namespace ckzg {
namespace api {
std::atomic<int> g_gpu_sha_min{0}, g_host_threads{0}, g_verify_pipe_min{1024}, g_verify_call_table{1};
}  // namespace api
}  // namespace ckzg

static std::atomic<int> in_library{0};
struct InLibrary {   // counted strictly inside the shared hold of guarded()
    std::atomic<int> &c;
    explicit InLibrary(std::atomic<int> &c_) : c(c_) { c.fetch_add(1, std::memory_order_acq_rel); }
    ~InLibrary() { c.fetch_sub(1, std::memory_order_acq_rel); }
};

static uint64_t mix(uint64_t x, uint64_t key) {
    x ^= key * 0x9e3779b97f4a7c15ull;
    x ^= x >> 29;
    x *= 0xbf58476d1ce4e5b9ull;
    return x ^ (x >> 32);
}

// After a burst of callers has gone, few callers must get the go-alone regime back (`peak` has to decay all the way:
// an integer eighth of a peak below 8 is zero, and the round-4 combiner stayed in the batching regime for good).
static int burst_then_few() {
    Combiner cb(/*max_batch=*/32, /*in=*/32 * 8, /*out=*/32 * 8, /*max_active=*/2, /*solo_below=*/3);
    auto phase = [&](int threads, int calls) {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) {
            th.emplace_back([&, t]() {
                for (int c = 0; c < calls; c++) {
                    uint64_t in = (uint64_t)t * 100000 + c, out = 0;
                    (void)guarded([&]() -> C_KZG_RET {
                        return cb.submit(
                            nullptr, 0,
                            [&]() -> C_KZG_RET {
                                usleep(50);
                                out = mix(in, 1);
                                return C_KZG_OK;
                            },
                            [&](uint8_t *h_in, size_t idx) { memcpy(h_in + idx * 8, &in, 8); },
                            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *, size_t n) -> C_KZG_RET {
                                for (size_t i = 0; i < n; i++) {
                                    uint64_t v;
                                    memcpy(&v, h_in + i * 8, 8);
                                    v = mix(v, 1);
                                    memcpy(h_out + i * 8, &v, 8);
                                }
                                usleep(60);
                                return C_KZG_OK;
                            },
                            [&](const uint8_t *h_out, size_t idx, size_t) { memcpy(&out, h_out + idx * 8, 8); });
                    });
                    if (out != mix(in, 1)) abort();
                }
            });
        }
        for (auto &x : th) x.join();
    };
    phase(3, 300);
    const Combiner::Stats s0 = cb.stats();
    phase(6, 300);    // the burst: more callers than solo_below -> batches
    const Combiner::Stats s1 = cb.stats();
    phase(3, 600);    // few callers again
    const Combiner::Stats s2 = cb.stats();
    const uint64_t solo_fresh = s0.solo, solo_after = s2.solo - s1.solo;
    printf("burst_then_few: fresh 3 threads %llu/900 alone; burst of 6: %llu batches; 3 threads afterwards %llu/1800 alone\n",
           (unsigned long long)solo_fresh, (unsigned long long)(s1.batches - s0.batches), (unsigned long long)solo_after);
    if (solo_fresh != 900) return 5;
    if (s1.batches == s0.batches) return 5;
    return solo_after >= 1700 ? 0 : 5;   // (the first few calls after the burst may still join a batch)
}

// Queueing: with more callers than a launch takes, how long a call waits is measured in LAUNCH GENERATIONS (launches
// started between its submit() and its own launch).  Batches are first in, first out, a caller joins the oldest open
// batch it fits, and callers that had to wait for a buffer are admitted oldest first (tickets), so the mean is a few
// generations; this gate keeps it there and bounds the worst case.  (On a test box the worst case also contains the
// operating system: 96 runnable threads on a handful of cores can keep a thread off the CPU for a few dozen 100-us
// "launches" before it even reaches submit(), which is why the bound is not tighter.  The tail the round-4 review saw
// on the GPU -- worst call 10-14x the mean at 256 callers -- came from buffers growing inside launches: scratch_reserve.)
static int starvation() {
    const int threads = 96, calls = 80;
    std::atomic<long> generation{0};
    std::atomic<long> worst_wait{0}, total_wait{0}, served{0};
    for (int gather = 0; gather < 2; gather++) {
        Combiner cb(/*max_batch=*/16, /*in=*/16 * 8, /*out=*/16 * 8, /*max_active=*/2, /*solo_below=*/0, /*gather_us=*/gather ? 100 : 0);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) {
            th.emplace_back([&, t]() {
                for (int c = 0; c < calls; c++) {
                    const long g0 = generation.load(std::memory_order_acquire);
                    uint64_t in = (uint64_t)t << 32 | (uint64_t)c, out = 0;
                    long my_gen = -1;
                    (void)guarded([&]() -> C_KZG_RET {
                        return cb.submit(
                            nullptr, 0,
                            [&]() -> C_KZG_RET {
                                my_gen = generation.fetch_add(1, std::memory_order_acq_rel);
                                usleep(80);
                                out = mix(in, 3);
                                return C_KZG_OK;
                            },
                            [&](uint8_t *h_in, size_t idx) { memcpy(h_in + idx * 8, &in, 8); },
                            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *, size_t n) -> C_KZG_RET {
                                const uint64_t gen = (uint64_t)generation.fetch_add(1, std::memory_order_acq_rel);
                                for (size_t i = 0; i < n; i++) {
                                    uint64_t v;
                                    memcpy(&v, h_in + i * 8, 8);
                                    v = mix(v, 3) ^ (gen << 48);   // the launch's generation rides in the top bits
                                    memcpy(h_out + i * 8, &v, 8);
                                }
                                usleep(100);
                                return C_KZG_OK;
                            },
                            [&](const uint8_t *h_out, size_t idx, size_t) { memcpy(&out, h_out + idx * 8, 8); });
                    });
                    if (my_gen < 0) {   // served by a batch launch: recover its generation from the top bits
                        my_gen = (long)((out ^ mix(in, 3)) >> 48);
                        out ^= (uint64_t)my_gen << 48;
                    }
                    if (out != mix(in, 3)) abort();
                    const long waited = ((my_gen - g0) & 0xffff);
                    total_wait.fetch_add(waited);
                    served.fetch_add(1);
                    long w = worst_wait.load();
                    while (waited > w && !worst_wait.compare_exchange_weak(w, waited)) {
                    }
                }
            });
        }
        for (auto &x : th) x.join();
        const Combiner::Stats st = cb.stats();
        printf("starvation (gather %d): solo %llu batches %llu batched %llu largest %llu\n", gather, (unsigned long long)st.solo,
               (unsigned long long)st.batches, (unsigned long long)st.batched, (unsigned long long)st.largest);
    }
    const double mean = (double)total_wait.load() / (double)served.load();
    printf("starvation: %ld calls, launches between submit and own launch: mean %.2f, worst %ld\n", served.load(), mean, worst_wait.load());
    // 96 callers over batches of <= 16 and two places: ~ threads / 16 generations on average
    return mean <= 2.0 * (threads / 16 + 2) && worst_wait.load() <= 150 ? 0 : 6;
}

// THE STALL OF THE ROUND-5 DRIVER RUN.  A caller that needs a new batch buffer allocates it with the mutex released
// (milliseconds of hipHostMalloc).  When the allocation FAILED, round 5's submit() carried on where it was instead of
// looking again: it took a batch that had come back to the free list meanwhile and opened it -- although the launch that
// had been in flight when it decided to allocate had ended meanwhile too.  A batch opened on an idle device is released
// by nobody; its opener slept for good (futex_wait on the OPEN state word, no timeout: the one native thread left in
// profiles/r06_stall_coalesced_callers.txt).  Short runs of the commitment combiner's shape (24 callers x 6 calls, up to
// three callers on their own) with slow allocations that start failing after 0..7 of them: every call must come back,
// and WITHOUT the net under the protocol (Stats::rescued) or the wait deadline (Stats::gave_up) having been needed.
static int failed_allocation_on_a_device_gone_idle() {
    const int rounds = 120, threads = 24, calls = 6;
    uint64_t rescued = 0, gave_up = 0;
    std::atomic<long> wrong{0};
    for (int r = 0; r < rounds; r++) {
        g_allocs_left.store(r % 9 == 8 ? (1L << 40) : (r % 9));
        g_alloc_us.store(2000 + (r % 5) * 4000);
        Combiner cb(/*max_batch=*/256, /*in=*/256 * 8, /*out=*/256 * 8, /*max_active=*/2, /*solo_below=*/3);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) {
            th.emplace_back([&, t]() {
                std::mt19937_64 rng((uint64_t)r * 100 + t);
                for (int c = 0; c < calls; c++) {
                    const uint64_t in = rng();
                    uint64_t out = 0;
                    const int solo_us = 200 + (int)(rng() % 800), run_us = 500 + (int)(rng() % 3000);
                    const C_KZG_RET rc = guarded([&]() -> C_KZG_RET {
                        return cb.submit(
                            nullptr, 0,
                            [&]() -> C_KZG_RET {
                                usleep(solo_us);
                                out = mix(in, 5);
                                return C_KZG_OK;
                            },
                            [&](uint8_t *h_in, size_t idx) { memcpy(h_in + idx * 8, &in, 8); },
                            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *, size_t n) -> C_KZG_RET {
                                usleep(run_us);
                                for (size_t i = 0; i < n; i++) {
                                    uint64_t v;
                                    memcpy(&v, h_in + i * 8, 8);
                                    v = mix(v, 5);
                                    memcpy(h_out + i * 8, &v, 8);
                                }
                                return C_KZG_OK;
                            },
                            [&](const uint8_t *h_out, size_t idx, size_t) { memcpy(&out, h_out + idx * 8, 8); });
                    });
                    if (rc != C_KZG_OK || out != mix(in, 5)) wrong.fetch_add(1);
                }
            });
        }
        for (auto &x : th) x.join();
        const Combiner::Stats st = cb.stats();
        rescued += st.rescued;
        gave_up += st.gave_up;
    }
    g_allocs_left.store(1L << 40);
    g_alloc_us.store(0);
    printf("failed_allocation_on_a_device_gone_idle: %d runs, wrong %ld, open batches rescued %llu, calls that gave up %llu\n", rounds,
           wrong.load(), (unsigned long long)rescued, (unsigned long long)gave_up);
    return wrong.load() || rescued || gave_up ? 8 : 0;
}

// The member that runs a batch does not come back (a device wait that never ends, a thread that was descheduled for
// good): every OTHER member must come back with C_KZG_ERROR once the wait deadline has passed -- the reference's
// convention for an internal failure (src/common/ret.h:24-29), and what a caller of a library that cannot block expects
// (src/eip4844/eip4844.c:264-280 is straight-line code) -- the stuck launch's own caller gets its result when (if) it
// ends, and the combiner serves later callers as before: nothing leaks a launch place or a batch buffer.
static int runner_never_comes_back() {
    dev::wait_deadline_ms_ref().store(300);
    const int threads = 12;
    std::atomic<int> stall_left{1};           // the first batch launch stalls
    std::atomic<bool> stalling{false};
    std::atomic<long> errors{0}, wrong{0}, slow_errors{0};
    Combiner cb(/*max_batch=*/16, /*in=*/16 * 8, /*out=*/16 * 8, /*max_active=*/1);
    auto call = [&](uint64_t in, uint64_t &out) -> C_KZG_RET {
        return guarded([&]() -> C_KZG_RET {
            return cb.submit(
                nullptr, 0,
                [&]() -> C_KZG_RET {
                    usleep(20000);   // long enough for everybody else to queue up behind it
                    out = mix(in, 9);
                    return C_KZG_OK;
                },
                [&](uint8_t *h_in, size_t idx) { memcpy(h_in + idx * 8, &in, 8); },
                [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *, size_t n) -> C_KZG_RET {
                    if (n >= 2 && stall_left.fetch_sub(1) == 1) {
                        stalling.store(true);
                        usleep(1500000);   // five deadlines
                        stalling.store(false);
                    }
                    for (size_t i = 0; i < n; i++) {
                        uint64_t v;
                        memcpy(&v, h_in + i * 8, 8);
                        v = mix(v, 9);
                        memcpy(h_out + i * 8, &v, 8);
                    }
                    return C_KZG_OK;
                },
                [&](const uint8_t *h_out, size_t idx, size_t) { memcpy(&out, h_out + idx * 8, 8); });
        });
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) {
        th.emplace_back([&, t]() {
            for (int c = 0; c < 3; c++) {
                const uint64_t in = (uint64_t)t * 1000 + c;
                uint64_t out = 0;
                const auto t0 = std::chrono::steady_clock::now();
                const C_KZG_RET r = call(in, out);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (r == C_KZG_ERROR) {
                    errors.fetch_add(1);
                    if (ms > 1200.0) slow_errors.fetch_add(1);   // an error must arrive AT the deadline, not when the stall ends
                } else if (r != C_KZG_OK || out != mix(in, 9)) {
                    wrong.fetch_add(1);
                }
            }
        });
    }
    for (auto &x : th) x.join();
    // afterwards: the combiner works as before
    long after_wrong = 0;
    std::vector<std::thread> th2;
    std::atomic<long> after_bad{0};
    for (int t = 0; t < threads; t++) {
        th2.emplace_back([&, t]() {
            for (int c = 0; c < 20; c++) {
                const uint64_t in = 777000 + (uint64_t)t * 100 + c;
                uint64_t out = 0;
                if (call(in, out) != C_KZG_OK || out != mix(in, 9)) after_bad.fetch_add(1);
            }
        });
    }
    for (auto &x : th2) x.join();
    after_wrong = after_bad.load();
    dev::wait_deadline_ms_ref().store(30000);
    printf("runner_never_comes_back: %ld callers got C_KZG_ERROR at the deadline (%ld late), wrong %ld, wrong afterwards %ld, "
           "expired waits %llu\n", errors.load(), slow_errors.load(), wrong.load(), after_wrong,
           (unsigned long long)dev::expired_waits_ref().load());
    if (errors.load() < 1 || slow_errors.load() || wrong.load() || after_wrong) return 7;
    return 0;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 48, calls = argc > 2 ? atoi(argv[2]) : 400;
    std::atomic<long> wrong{0}, solos{0};
    // scenarios: plain; few-callers-go-alone threshold; only ONE batch buffer could be allocated (the others fail: callers
    // wait for the buffer or go alone); every fifth launch fails as a whole (its members all see the error, later
    // launches are unaffected); gathering
    for (int scenario = 0; scenario < 5; scenario++) {
        const int solo_below = scenario == 1 ? 4 : 0;
        g_allocs_left.store(scenario == 2 ? 3 : 1L << 40);    // 2 buffers of the first batch + h_in of the second
        std::atomic<long> launches{0}, failed_calls{0};
        // scenario 4: an operation that gathers (idle-path callers with recent company wait up to 200 us for each other)
        Combiner cb(/*max_batch=*/32, /*in=*/32 * 8, /*out=*/32 * 8, /*max_active=*/2, solo_below, /*gather_us=*/scenario == 4 ? 200 : 0);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) {
            th.emplace_back([&, t]() {
                std::mt19937_64 rng(1234 + t);
                for (int c = 0; c < calls; c++) {
                    const uint64_t key = rng() % 3;                 // three kinds of requests that must not share a launch
                    const uint64_t in = rng();
                    const int kind = (int)(mix(in, 77) % 16);       // 0: flagged (BADARGS), 1: RETRY_SOLO, else fine
                    uint64_t out = 0;
                    auto solo = [&]() -> C_KZG_RET {
                        solos.fetch_add(1, std::memory_order_relaxed);
                        usleep(120);                                // (a launch takes time: that is what lets callers queue)
                        if (kind == 0) return C_KZG_BADARGS;
                        out = mix(in, key);
                        return C_KZG_OK;
                    };
                    C_KZG_RET r = guarded([&]() -> C_KZG_RET {   // as every entry point of the library does
                      InLibrary here(in_library);
                      return cb.submit(
                        &key, sizeof key, solo,
                        [&](uint8_t *h_in, size_t idx) {
                            memcpy(h_in + idx * 8, &in, 8);
                        },
                        [&, key](const uint8_t *h_in, uint8_t *h_out, uint8_t *st, size_t n) -> C_KZG_RET {
                            for (size_t i = 0; i < n; i++) {
                                uint64_t v;
                                memcpy(&v, h_in + i * 8, 8);
                                const int k2 = (int)(mix(v, 77) % 16);   // the unit's kind as the BATCH path sees it
                                const uint64_t res = mix(v, key);
                                memcpy(h_out + i * 8, &res, 8);
                                st[i] = k2 == 0 ? (uint8_t)C_KZG_BADARGS : (k2 == 1 ? Combiner::RETRY_SOLO : 0);
                            }
                            usleep(150);
                            if (scenario == 3 && launches.fetch_add(1) % 5 == 4) return C_KZG_ERROR;
                            return C_KZG_OK;
                        },
                        [&](const uint8_t *h_out, size_t idx, size_t) { memcpy(&out, h_out + idx * 8, 8); });
                    });
                    // what must have happened: a flagged input fails (its own call only), every other one holds
                    // mix(in, key) -- a launch that mixed keys or a swapped slot would give another value; a RETRY_SOLO
                    // unit got its value from the solo path afterwards
                    if (scenario == 3 && r == C_KZG_ERROR) {
                        failed_calls.fetch_add(1);   // (a member of a failed launch; flagged units keep their own code)
                        continue;
                    }
                    if (kind == 0 ? r != C_KZG_BADARGS : (r != C_KZG_OK || out != mix(in, key))) wrong.fetch_add(1);
                }
            });
        }
        for (auto &x : th) x.join();
        const Combiner::Stats st = cb.stats();
        printf("scenario %d: failed-launch calls %ld; ", scenario, failed_calls.load());
        if (scenario == 3 && failed_calls.load() == 0) return 3;
        if (scenario != 3 && failed_calls.load() != 0) return 3;
        printf("solo_below=%d: calls %llu solo %llu batches %llu batched %llu largest %llu retried %llu wrong %ld\n", solo_below,
               (unsigned long long)st.calls, (unsigned long long)st.solo, (unsigned long long)st.batches,
               (unsigned long long)st.batched, (unsigned long long)st.largest, (unsigned long long)st.retried, wrong.load());
        if (st.calls != (uint64_t)threads * calls) return 2;
        if (threads <= solo_below) {
            if (st.batches != 0) return 2;   // so few callers that each goes alone
        } else if (st.batches == 0 || (threads >= 16 && st.largest < 2) || st.retried == 0) {
            return 2;
        }
    }
    if (wrong.load()) return 1;
    if (int rc = burst_then_few()) return rc;
    if (int rc = runner_never_comes_back()) return rc;
    if (int rc = failed_allocation_on_a_device_gone_idle()) return rc;
    return threads >= 48 ? starvation() : 0;
}
