// combiner.hpp -- turns concurrent one-unit callers of the ckzg.h entry points into batch launches.
//
// The reference API is one blob per call (src/eip4844/eip4844.h:43-81, src/eip7594/eip7594.h:35-57); its only
// parallel shape is N threads each making one-blob calls on a shared KZGSettings (bindings/go/main_test.go:953-971,
// bindings/rust/src/bindings/mod.rs:910-913).  A GPU serves N such callers best with ONE launch over N blobs, so a
// Combiner sits between the unchanged C-ABI functions and the batch paths that back the additive *_batch symbols:
//
//   * A caller that finds no launch of its operation in flight runs its own one-unit call at once, exactly as
//     without a combiner: no queue, no timer, no staging -- the idle path keeps its latency.
//   * Otherwise the caller JOINS the open batch for its key (or opens one): it copies its own inputs into the
//     batch's page-locked buffer -- every caller moves its own bytes, in parallel -- and sleeps.  When a launch
//     finishes, the oldest open batch is closed and ONE of its members woken to run it (any member will do: the
//     first that sees the batch released claims it; while fewer than `max_active` launches are in flight a batch
//     does not wait for a launch to finish but goes as soon as it is about as large as the previous one): it leases a slot, runs the batch path once over everything
//     that queued up (inputs DMA'd from the page-locked buffer in place, results DMA'd into the batch's
//     page-locked output buffer), hands its launch place on, and wakes the members, who copy their own results
//     out and leave.  Whoever leaves a batch last returns its buffers to the free list.
//   * Members sleep on the batch's state word (futex), not on the combiner's mutex: a call takes that mutex once,
//     to join; 127 members woken at once do not queue up behind each other on their way out.
//   * A unit the batch path flags in its per-unit status (a non-canonical field element, an invalid commitment)
//     fails ITS caller only; a failure of the launch itself (HIP error, out of memory) fails every member.
//
// Batches are per key: requests may only share a launch when the batch path treats them alike (same operation,
// same outputs wanted; for recover_cells_and_kzg_proofs the same set of cell indices).
#pragma once
#include <pthread.h>

#include <atomic>
#include <climits>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <optional>
#include <thread>
#include <vector>

#include "api_common.hpp"

// page-locked batch buffers (tests/native/combiner_stress.cpp, which runs the state machine without a GPU, supplies
// plain malloc / free instead)
#ifndef CKZG_COMBINER_PINNED_ALLOC
#define CKZG_COMBINER_PINNED_ALLOC(pp, bytes) (hipHostMalloc((void **)(pp), (bytes), hipHostMallocPortable) == hipSuccess)
#define CKZG_COMBINER_PINNED_FREE(p) ((void)hipHostFree(p))
#endif
#ifndef CKZG_COMBINER_RESERVE_SCALE
#define CKZG_COMBINER_RESERVE_SCALE(max_units, n) ::ckzg::dev::ReserveScale reserve_hint_((max_units), (n))
#endif

namespace ckzg {
namespace api {

namespace detail {
// Hundreds of callers take the combiner's mutex for a few dozen nanoseconds each, in bursts (a batch's members
// return to their callers together and come back together): spin briefly before sleeping.
class AdaptiveMutex {
   public:
    AdaptiveMutex() {
        pthread_mutexattr_t a;
        pthread_mutexattr_init(&a);
        pthread_mutexattr_settype(&a, PTHREAD_MUTEX_ADAPTIVE_NP);
        pthread_mutex_init(&m, &a);
        pthread_mutexattr_destroy(&a);
    }
    ~AdaptiveMutex() { pthread_mutex_destroy(&m); }
    AdaptiveMutex(const AdaptiveMutex &) = delete;
    AdaptiveMutex &operator=(const AdaptiveMutex &) = delete;
    void lock() { pthread_mutex_lock(&m); }
    void unlock() { pthread_mutex_unlock(&m); }
    bool try_lock() { return pthread_mutex_trylock(&m) == 0; }

   private:
    pthread_mutex_t m;
};
}  // namespace detail

class Combiner {
   public:
    // in_bytes / out_bytes: page-locked bytes a batch of max_batch units needs; layout is the call site's business
    // solo_below: while no more than this many threads have been calling concurrently of late, every caller runs its own
    // one-unit call (as many at a time as there are callers): right for operations whose single call is mostly HOST
    // work that scales over the callers' own cores (a verification: transcript hash + pairing), wrong for the ones a
    // single launch already fills the device with (0: only a caller that finds nothing in flight goes alone).
    // gather_us: for operations whose launch takes the same few milliseconds for 1 or 16 units (FK20 proofs: the G1
    // transforms are latency) a caller that finds the device idle while OTHER callers have been about (peak >= 2) does
    // not go alone -- its one-unit launch would make everybody who arrives a moment later wait for it and then share a
    // second launch -- but opens a batch that goes when everyone seen lately has joined, or after this many
    // microseconds.  Callers that come back together (the members of the previous batch) then share ONE launch.
    // 0: no gathering (operations whose launch is short against such a wait).
    Combiner(size_t max_batch_, size_t in_bytes_, size_t out_bytes_, int max_active_, int solo_below_ = 0, int gather_us_ = 0)
        : max_batch(max_batch_), in_bytes(in_bytes_), out_bytes(out_bytes_), max_active(max_active_ < 1 ? 1 : max_active_),
          solo_below(solo_below_), gather_ns((long)gather_us_ * 1000L) {
        all.reserve((size_t)max_active + 2);        // so that the bookkeeping of a call cannot throw
        free_list.reserve((size_t)max_active + 2);
    }
    Combiner(const Combiner &) = delete;
    Combiner &operator=(const Combiner &) = delete;
    ~Combiner() {
        // free_trusted_setup must not race with calls (src/setup/setup.c:162-190): nobody is inside
        for (Batch *b : all) {
            if (b->h_in) CKZG_COMBINER_PINNED_FREE(b->h_in);
            if (b->h_out) CKZG_COMBINER_PINNED_FREE(b->h_out);
            delete b;
        }
    }

    struct Stats {
        uint64_t calls = 0;        // submit() calls
        uint64_t solo = 0;         // ... that ran their own one-unit call (idle path)
        uint64_t batches = 0;      // batch launches
        uint64_t batched = 0;      // calls served by those launches
        uint64_t largest = 0;      // units in the largest launch
        uint64_t run_us = 0;       // wall time the batch launches took (lease + copies + kernels), microseconds
        uint64_t retried = 0;      // calls a batch launch could not answer and that ran alone afterwards
        uint64_t rescued = 0;      // open batches of an idle device released by a member's periodic look (must stay 0 for an
                                   // operation that does not gather: the net under the protocol, never a path of it)
        uint64_t gave_up = 0;      // calls that left with C_KZG_ERROR at the wait deadline
    };
    Stats stats() {
        std::lock_guard<detail::AdaptiveMutex> lock(mu);
        return st;
    }

    // diagnostics (ckzg_hip_debug_dump): the state of the queue, without waiting for the mutex -- the dump may be asked
    // for BECAUSE somebody holds it for good
    void dump(int fd, const char *name) {
        if (!mu.try_lock()) {
            dprintf(fd, "  combiner %s: mutex held -- inside=%d\n", name, inside.load(std::memory_order_relaxed));
            return;
        }
        dprintf(fd, "  combiner %s: inside=%d active=%d peak=%d pending=%zu batches=%zu free=%zu allocating=%d alloc_failed=%d "
                    "tickets next=%llu serving=%llu | calls=%llu solo=%llu launches=%llu batched=%llu\n",
                name, inside.load(std::memory_order_relaxed), active, peak, pending.size(), all.size(), free_list.size(), (int)allocating,
                (int)alloc_failed, (unsigned long long)next_ticket, (unsigned long long)serving, (unsigned long long)st.calls,
                (unsigned long long)st.solo, (unsigned long long)st.batches, (unsigned long long)st.batched);
        static const char *const names[] = {"OPEN", "RELEASED", "RUNNING", "DONE"};
        for (Batch *b : all) {
            bool queued = false;
            for (Batch *q : pending) queued = queued || q == b;
            const uint32_t stt = b->state.load(std::memory_order_relaxed);
            dprintf(fd, "    batch %p: state=%s n=%zu copied=%zu refs=%u queued=%d\n", (void *)b, stt < 4 ? names[stt] : "?", b->n,
                    b->copied.load(std::memory_order_relaxed), b->refs.load(std::memory_order_relaxed), (int)queued);
        }
        mu.unlock();
    }

    // status value a batch path gives a unit it cannot answer for inside the batch (a verification batch that did not
    // come out true says nothing about its single members): that caller runs its own one-unit call afterwards
    static constexpr uint8_t RETRY_SOLO = 0xff;

    // solo():                               the caller's own one-unit call -> C_KZG_RET
    // copy_in(h_in, idx):                   place this caller's inputs as unit idx of the batch buffer
    // run(h_in, h_out, status, n):          the batch path over units 0..n-1; status[i] != 0 flags unit i.  Runs on the
    //                                       thread of ANY member of the batch: it may only depend on the key
    // copy_out(h_out, idx, n):              fetch unit idx's results (called only if the unit succeeded)
    template <class Solo, class CopyIn, class Run, class CopyOut>
    C_KZG_RET submit(const void *key, size_t key_len, Solo &&solo, CopyIn &&copy_in, Run &&run, CopyOut &&copy_out) {
        std::unique_lock<detail::AdaptiveMutex> lock(mu);
        st.calls++;
        const int in = inside.fetch_add(1, std::memory_order_relaxed) + 1;
        if (in > peak) peak = in;
        struct Leave {
            std::atomic<int> &c;
            ~Leave() { c.fetch_sub(1, std::memory_order_relaxed); }
        } leave{inside};
        Batch *b = nullptr;
        bool release_now = false, gathering = false;
        bool no_memory_now = false;   // THIS caller has just seen an allocation fail (its own state: nobody else's business)
        // Callers that had to wait for a batch buffer are served roughly in the order they came: a waiter takes a ticket,
        // every batch buffer that comes back admits the oldest max_batch tickets (`serving` advances by that much), and
        // a newcomer queues behind the tickets that are out.  (Without it the woken waiters raced newcomers for the
        // mutex: in the CPU stress test -- 96 callers, batches of 16 -- a call could be passed over by 60 launches while
        // the mean waited 4.  Admitting one ticket at a time instead made the line itself the bottleneck.)
        bool have_ticket = false;
        uint64_t ticket = 0;
        auto admitted = [&]() { have_ticket = false; };
        // No wait in here is unbounded (device.hpp: bounded waits): the queueing phase as a whole -- for a batch buffer, a
        // launch place, a turn in the line, an allocation another caller is making -- gives up at the deadline with
        // C_KZG_ERROR for THIS caller, who has joined nothing yet.
        std::optional<dev::WaitNote> queueing;
        for (;;) {
            if (queueing && queueing->expired()) return C_KZG_ERROR;
            // (the bounded wait below is the safety valve: a ticket that no returning buffer admits -- its group found
            // room elsewhere -- proceeds after a millisecond)
            if (have_ticket ? ticket >= serving : serving < next_ticket) {
                if (!have_ticket) {
                    have_ticket = true;
                    ticket = next_ticket++;
                }
                if (!queueing) queueing.emplace("combiner: a turn in the line for a batch buffer", this);
                if (cv_pool.wait_for(lock, std::chrono::milliseconds(1)) == std::cv_status::timeout && ticket >= serving)
                    serving = ticket + 1;
                continue;
            }
            // (gathering: the device is idle, but this caller has had company lately -- or a gathering batch is open)
            gathering = gather_ns > 0 && active == 0 && (peak >= 2 || !pending.empty()) && !(peak <= solo_below && pending.empty());
            if ((active == 0 && !gathering) || (peak <= solo_below && pending.empty())) {
                // the idle path: nothing of this operation is in flight (so nothing is queued either), or so few callers
                // are about that each is better off with a launch of its own
                active++;
                st.solo++;
                admitted();
                lock.unlock();
                C_KZG_RET r = guarded([&]() -> C_KZG_RET { return solo(); });
                lock.lock();
                decay_peak();   // the go-alone regime has to forget a past burst too (it runs no batch that would)
                Batch *next = launch_done();
                lock.unlock();
                release(next);
                return r;
            }
            for (Batch *p : pending) {
                if (p->n < (gathering ? max_batch : batch_cap()) && p->key.size() == key_len && (key_len == 0 || !memcmp(p->key.data(), key, key_len))) {
                    b = p;
                    break;
                }
            }
            if (!b) {
                if (free_list.empty() && (int)all.size() < max_active + 2 && !alloc_failed) {
                    // A new batch needs page-locked buffers: tens of milliseconds of hipHostMalloc for the large ones.
                    // ONE caller allocates, with the mutex released (everybody else keeps joining, running and
                    // leaving meanwhile); callers that would need the same buffer go alone for that while, as they
                    // would without a combiner.
                    if (allocating) {
                        if (peak > 8) {
                            // a crowd: going alone would put hundreds of one-unit calls in line for the eight stream
                            // slots (measured at 256 callers: 65 ms for the calls that met an allocation) -- wait the
                            // 10-30 ms the allocation takes and join the new batch
                            if (!have_ticket) {
                                have_ticket = true;
                                ticket = next_ticket++;
                            }
                            if (!queueing) queueing.emplace("combiner: another caller's batch-buffer allocation", this);
                            cv_pool.wait_for(lock, std::chrono::milliseconds(2));
                            if (ticket >= serving) serving = ticket + 1;   // (woken by the allocator, or impatient: look again)
                            continue;
                        }
                        admitted();
                        lock.unlock();
                        return guarded([&]() -> C_KZG_RET { return solo(); });
                    }
                    allocating = true;
                    lock.unlock();
                    Batch *nb = allocate_batch();
                    lock.lock();
                    allocating = false;
                    if (nb) {
                        all.push_back(nb);         // (capacity reserved)
                        free_list.push_back(nb);   // (capacity reserved)
                    } else {
                        alloc_failed = true;       // do not try again on every call
                        no_memory_now = true;
                    }
                    cv_pool.notify_all();
                    // The world has moved on while the mutex was released: look again (idle path, a joinable batch, ...),
                    // ALSO when the allocation failed.  Round 5 fell through in that case and took a batch that had come
                    // back to the free list meanwhile -- on a device that had gone idle meanwhile as well: the batch was
                    // opened with no launch in flight to release it, and its opener slept for good.  That was the stall
                    // of the round-5 driver run (tests/test_gpu_alloc_failures.py, the coalesced callers; reproduced on
                    // the CPU by tests/native/combiner_stress.cpp: failed_allocation_on_a_device_gone_idle).
                    continue;
                }
                b = fresh_batch();
                if (b) {
                    bool queued = false;
                    try {
                        b->key.assign((const uint8_t *)key, (const uint8_t *)key + key_len);
                        pending.push_back(b);
                        queued = true;
                    } catch (...) {
                    }
                    if (!queued) {
                        free_list.push_back(b);   // (capacity reserved)
                        b = nullptr;
                        no_memory_now = true;
                    }
                }
            }
            if (b) {
                // A launch place is free while others are in flight: callers of the launch that has just ended are on
                // their way back, so the batch goes when it holds most of its share of the callers seen lately -- or
                // when another launch ends (launch_done), whichever comes first.  No timer: the wait is bounded by
                // launches that are in flight.
                if (active < places() && b == pending.front() && b->n + 1 >= (gathering ? gather_target() : go_threshold())) {
                    pending.pop_front();
                    active++;
                    release_now = true;
                }
                admitted();
                break;
            }
            if (all.empty() || no_memory_now) {
                // no page-locked memory / no memory for the bookkeeping: this call goes alone, unqueued
                admitted();
                lock.unlock();
                return guarded([&]() -> C_KZG_RET { return solo(); });
            }
            if (!have_ticket) {   // every batch buffer is in use: wait for one, or for a launch place, in line
                have_ticket = true;
                ticket = next_ticket++;
            }
            if (ticket < serving) {
                // admitted, but nothing to take yet: a buffer coming back or a launch ending notifies; in slices, so that a
                // notification that went missing costs a slice and not the call
                if (!queueing) queueing.emplace("combiner: a batch buffer or a launch place", this);
                (void)cv_pool.wait_for(lock, std::chrono::nanoseconds(WAIT_SLICE_NS));
            }
        }
        const size_t idx = b->n++;
        b->refs.fetch_add(1, std::memory_order_relaxed);
        lock.unlock();
        if (release_now) release(b);
        copy_in(b->h_in, idx);
        b->copied.fetch_add(1, std::memory_order_release);
        // the opener of a batch of a gathering operation looks after it: while the batch is open it wakes every gather_ns,
        // and if the device is idle by then (nothing in flight to release the batch when it ends) it releases it itself
        const bool caretaker = gather_ns > 0 && idx == 0;
        // The open batch of a device that has gone idle is released by whoever notices: its opener within gather_ns (a
        // gathering operation), any member within a slice otherwise -- by the protocol that cannot happen to an
        // operation that does not gather (a batch is only opened while a launch is in flight, and every launch that
        // ends releases the oldest open batch), so this is the net under the protocol, not a path of it.
        auto release_if_device_idle = [&](bool by_the_opener_in_time) {
            bool mine = false;
            lock.lock();
            if (active == 0 && !pending.empty() && pending.front() == b) {
                pending.pop_front();
                active++;
                mine = true;
                if (!by_the_opener_in_time) {
                    st.rescued++;
                    fprintf(stderr, "[ckzg-hip] combiner: an open batch (%zu members) sat on an idle device and was released by a "
                                    "member's periodic look: peak=%d inside=%d pending=%zu free=%zu calls=%llu solo=%llu launches=%llu\n",
                            b->n, peak, inside.load(std::memory_order_relaxed), pending.size(), free_list.size(),
                            (unsigned long long)st.calls, (unsigned long long)st.solo, (unsigned long long)st.batches);
                }
            }
            lock.unlock();
            if (mine) release(b);
        };
        std::optional<dev::WaitNote> waiting;
        bool gave_up = false;
        for (;;) {
            uint32_t s = b->state.load(std::memory_order_acquire);
            if (s == DONE) break;
            if (s == RELEASED) {
                if (b->state.compare_exchange_strong(s, RUNNING, std::memory_order_acq_rel)) {
                    run_batch(b, run);
                    break;
                }
                continue;
            }
            if (s == OPEN && caretaker) {
                futex_wait_for(&b->state, s, gather_ns);
                if (b->state.load(std::memory_order_acquire) != OPEN) continue;
                release_if_device_idle(true);
                continue;
            }
            // asleep until the state word changes -- in slices: a wake-up that went missing costs a slice, not the call
            if (!waiting) waiting.emplace("combiner: the batch this call joined", b);
            futex_wait_for(&b->state, s, WAIT_SLICE_NS);
            if (b->state.load(std::memory_order_acquire) != s) continue;
            if (s == OPEN) release_if_device_idle(false);
            if (waiting->expired()) {
                // OPEN: nothing released the batch and the device never went idle (launches of other keys keep it busy and
                // this batch is not the oldest); RUNNING: the member that runs it has not come back (a device wait of its
                // own would have expired first).  This caller gets C_KZG_ERROR; the others have deadlines of their own.
                gave_up = true;
                break;
            }
        }
        if (gave_up) {
            lock.lock();
            st.gave_up++;
            if (b->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                // the last member to leave: a batch nobody is left to run must not stay in the queue (a launch that
                // ends would release it to nobody and its launch place would never come back)
                const uint32_t st = b->state.load(std::memory_order_acquire);
                if (st == OPEN) {
                    for (auto it = pending.begin(); it != pending.end(); ++it) {
                        if (*it == b) {
                            pending.erase(it);
                            break;
                        }
                    }
                }
                if (st == OPEN || st == DONE) {
                    free_list.push_back(b);   // (capacity reserved)
                    cv_pool.notify_all();
                }
            }
            lock.unlock();
            return C_KZG_ERROR;
        }
        const size_t n = b->n;
        const bool retry = b->status[idx] == RETRY_SOLO;
        const C_KZG_RET mine = b->status[idx] ? (C_KZG_RET)b->status[idx] : b->ret;
        if (mine == C_KZG_OK) copy_out((const uint8_t *)b->h_out, idx, n);
        if (b->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            lock.lock();
            free_list.push_back(b);   // (capacity reserved)
            serving = serving + max_batch < next_ticket ? serving + max_batch : next_ticket;   // the oldest waiters' turn
            cv_pool.notify_all();
            lock.unlock();
        }
        if (retry) {
            {
                std::lock_guard<detail::AdaptiveMutex> relock(mu);
                st.retried++;
            }
            return guarded([&]() -> C_KZG_RET { return solo(); });   // (unqueued: it leases a slot of its own)
        }
        return mine;
    }

   private:
    enum : uint32_t { OPEN = 0, RELEASED = 1, RUNNING = 2, DONE = 3 };
    struct Batch {
        uint8_t *h_in = nullptr, *h_out = nullptr;   // page-locked
        std::vector<uint8_t> status, key;
        size_t n = 0;                                // members; grows under mu while the batch is in `pending`
        std::atomic<uint32_t> state{OPEN};           // OPEN -> RELEASED (it may run) -> RUNNING (claimed) -> DONE
        std::atomic<uint32_t> refs{0};               // members that have not left yet
        std::atomic<size_t> copied{0};               // members whose inputs are in h_in
        C_KZG_RET ret = C_KZG_OK;
    };

    // The calling member has claimed the batch (RELEASED -> RUNNING): n is final, the batch is out of `pending`.
    template <class Run>
    void run_batch(Batch *b, Run &run) {
        const size_t n = b->n;
        bool inputs_in = true;
        if (b->copied.load(std::memory_order_acquire) != n) {   // members still copying in: microseconds
            dev::WaitNote copying("combiner: members copying their inputs in", b);
            while (b->copied.load(std::memory_order_acquire) != n) {
                std::this_thread::yield();
                if (copying.expired()) {
                    inputs_in = false;
                    break;
                }
            }
        }
        memset(b->status.data(), 0, n);
        const auto t_run = std::chrono::steady_clock::now();
        C_KZG_RET r = C_KZG_ERROR;
        if (inputs_in) {
            CKZG_COMBINER_RESERVE_SCALE(max_batch, n);   // buffers that grow in this launch grow for the largest batch
            r = guarded([&]() -> C_KZG_RET { return run((const uint8_t *)b->h_in, b->h_out, b->status.data(), n); });
        }
        const uint64_t run_us =
            (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_run).count();
        if (r == C_KZG_BADARGS) {
            // flagged units answer for themselves; without a flag the verdict concerns the whole batch
            for (size_t i = 0; i < n; i++) {
                if (b->status[i]) {
                    r = C_KZG_OK;
                    break;
                }
            }
        }
        b->ret = r;
        Batch *next;
        {
            std::lock_guard<detail::AdaptiveMutex> lock(mu);
            st.batches++;
            st.batched += n;
            if (n > st.largest) st.largest = n;
            st.run_us += run_us;
            decay_peak();
            next = launch_done();
        }
        release(next);   // the next launch starts while this one's members are being woken
        b->state.store(DONE, std::memory_order_release);
        futex_wake(&b->state, INT_MAX);
    }

    // mu held.  Forget callers that have stopped calling: an eighth per finished launch, but always at least one (an
    // integer eighth of a peak below 8 is zero: after one burst of 4..7 callers the estimate never came down again and
    // the `solo_below` regime was lost for the lifetime of the KZGSettings -- round-4 advisor finding), never below the
    // number of threads inside right now.
    void decay_peak() {
        const int now = inside.load(std::memory_order_relaxed);
        const int dec = peak - (peak / 8 > 1 ? peak / 8 : 1);
        peak = dec > now ? dec : now;
    }

    // mu held.  A launch has ended: its place goes to the oldest open batch (returned, to be release()d once mu is
    // dropped), or is given up.
    Batch *launch_done() {
        // (solo launches of a `solo_below` operation may exceed the places batches rotate through: a batch only takes
        // the place of a launch that leaves fewer than max_active behind)
        if (!pending.empty() && active <= places()) {
            Batch *nb = pending.front();
            if (gather_ns > 0 && active == 1 && nb->n < gather_target()) {
                // a gathering operation, the device is about to be idle and the next batch holds only a few of the callers
                // seen lately -- the others are the members of the launch that has just ended, on their way back.  Leave it
                // open: they join it (the one who completes it releases it) or its opener releases it within gather_ns.
                // (Released now, it would run half empty while they queue behind it, and the two groups would alternate
                // for good: 8 callers of compute_cells_and_kzg_proofs settled into launches of 1 and 7.)
                active--;
                cv_pool.notify_all();
                return nullptr;
            }
            pending.pop_front();
            return nb;
        }
        active--;
        cv_pool.notify_all();
        return nullptr;
    }
    // `peak` estimates how many threads are calling this operation concurrently.  Their fair division over the
    // launch places keeps `max_active` batches of similar size rotating; one batch that swallows every caller
    // would leave the device idle while they all copy out, come back and copy in.
    // members an open batch waits for while a launch place is free: three quarters of a place's share
    size_t go_threshold() const {
        static const long pct = dev::ab_knob("CKZG_HIP_COALESCE_GO_PCT", 75);
        const size_t t = (size_t)((long)peak * pct / 100) / (size_t)max_active;
        return t < 1 ? 1 : (t > max_batch ? max_batch : t);
    }
    // launch places batches rotate through.  A gathering operation (launches of a few milliseconds whatever their
    // size) keeps ONE launch in flight while at most 64 callers are about: everybody shares it and comes back together
    // (8 callers of compute_cells_and_kzg_proofs: 1.0 k calls/s at 7.7 ms -> 1.7 k at 4.7 ms; two callers no longer
    // run two chip-filling one-blob launches against each other).  From there two batches rotate as for the other
    // operations, so that one is copied in and out while the other computes (128 callers: 7.3 k against 6.9 k calls/s).
    int places() const { return gather_ns > 0 && peak <= 64 ? 1 : max_active; }
    // members a gathering batch waits for (at most gather_ns): everybody seen lately
    size_t gather_target() const {
        const size_t t = (size_t)(peak < 1 ? 1 : peak);
        return t > max_batch ? max_batch : t;
    }
    // members a batch takes before later callers open the next one: a place's share and a quarter
    size_t batch_cap() const {
        static const long cap_pct = dev::ab_knob("CKZG_HIP_COALESCE_CAP_PCT", 125);
        const size_t c = ((size_t)((long)peak * cap_pct / 100) + (size_t)max_active - 1) / (size_t)max_active;
        return c < 1 ? 1 : (c > max_batch ? max_batch : c);
    }
    // the batch may run: whichever member sees this first claims it; one sleeper is woken in case all of them sleep
    static void release(Batch *nb) {
        if (!nb) return;
        nb->state.store(RELEASED, std::memory_order_release);
        futex_wake(&nb->state, 1);
    }

    // NOT under mu.  A new batch with its page-locked buffers, or nullptr.
    Batch *allocate_batch() {
        Batch *b = new (std::nothrow) Batch();
        bool ok = b != nullptr;
        // Portable: any device of a multi-device load may DMA from / into it
        ok = ok && CKZG_COMBINER_PINNED_ALLOC(&b->h_in, in_bytes ? in_bytes : 1);
        ok = ok && CKZG_COMBINER_PINNED_ALLOC(&b->h_out, out_bytes ? out_bytes : 1);
        if (ok) {
            try {
                b->status.resize(max_batch);
            } catch (...) {
                ok = false;
            }
        }
        if (!ok) {
            (void)hipGetLastError();
            if (b && b->h_in) CKZG_COMBINER_PINNED_FREE(b->h_in);
            if (b && b->h_out) CKZG_COMBINER_PINNED_FREE(b->h_out);
            delete b;
            return nullptr;
        }
        CKZG_TSAN_NEW_MEMORY(b->h_in, in_bytes);
        CKZG_TSAN_NEW_MEMORY(b->h_out, out_bytes);
        return b;
    }

    // mu held.  A batch with buffers from the free list, reset; nullptr if there is none now.
    Batch *fresh_batch() {
        if (free_list.empty()) return nullptr;
        Batch *b = free_list.back();
        free_list.pop_back();
        b->n = 0;
        b->refs.store(0, std::memory_order_relaxed);
        b->copied.store(0, std::memory_order_relaxed);
        b->state.store(OPEN, std::memory_order_relaxed);
        b->ret = C_KZG_OK;
        return b;
    }

    const size_t max_batch, in_bytes, out_bytes;
    const int max_active, solo_below;
    const long gather_ns;
    detail::AdaptiveMutex mu;
    std::condition_variable_any cv_pool;
    std::deque<Batch *> pending;       // open batches, oldest first
    std::vector<Batch *> all, free_list;
    int active = 0;                    // launches in flight (solo calls and batches)
    std::atomic<int> inside{0};        // threads inside submit()
    int peak = 0;                      // recent maximum of `inside` (decays by an eighth per batch launch)
    bool alloc_failed = false, allocating = false;
    uint64_t next_ticket = 0, serving = 0;   // the line of callers waiting for a batch buffer (first come, first served)
    Stats st;
};

}  // namespace api
}  // namespace ckzg
