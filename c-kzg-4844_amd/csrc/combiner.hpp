// combiner.hpp -- turns concurrent one-unit callers of the ckzg.h entry points into batch launches.
//
// The reference API is one blob per call (src/eip4844/eip4844.h:43-81, src/eip7594/eip7594.h:35-57); its only
// parallel shape is N threads each making one-blob calls on a shared KZGSettings (bindings/go/main_test.go:953-971,
// bindings/rust/src/bindings/mod.rs:910-913).  A GPU serves N such callers best with ONE launch over N blobs, so a
// Combiner sits between the unchanged C-ABI functions and the batch paths that back the additive *_batch symbols:
//
//   * While fewer than `max_active` launches of this operation are in flight, a caller runs its own one-unit call
//     at once, exactly as without a combiner: no queue, no timer, no staging -- the idle path keeps its latency.
//   * Otherwise the caller JOINS the open batch for its key (or opens one and becomes its owner): it copies its
//     own inputs into the batch's page-locked buffer -- every caller moves its own bytes, in parallel -- and
//     sleeps.  When a launch finishes, the oldest open batch is closed and its owner promoted: it leases a slot,
//     runs the batch path once over everything that was queued behind it (inputs DMA'd from the page-locked
//     buffer in place, results DMA'd into the batch's page-locked output buffer), and wakes the members, who
//     copy their own results out and leave.  Whoever leaves a batch last returns its buffers to the free list.
//   * A unit the batch path flags in its per-unit status (a non-canonical field element, an invalid commitment)
//     fails ITS caller only; a failure of the launch itself (HIP error, out of memory) fails every member.
//
// Batches are per key: requests may only share a launch when the batch path treats them alike (same operation,
// same outputs wanted; for recover_cells_and_kzg_proofs the same set of cell indices).
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "api_common.hpp"

namespace ckzg {
namespace api {

class Combiner {
   public:
    // in_bytes / out_bytes: page-locked bytes a batch of max_batch units needs; layout is the call site's business
    Combiner(size_t max_batch_, size_t in_bytes_, size_t out_bytes_, int max_active_)
        : max_batch(max_batch_), in_bytes(in_bytes_), out_bytes(out_bytes_), max_active(max_active_ < 1 ? 1 : max_active_) {}
    Combiner(const Combiner &) = delete;
    Combiner &operator=(const Combiner &) = delete;
    ~Combiner() {
        // free_trusted_setup must not race with calls (src/setup/setup.c:162-190): nobody is inside
        for (Batch *b : all) {
            if (b->h_in) (void)hipHostFree(b->h_in);
            if (b->h_out) (void)hipHostFree(b->h_out);
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
    };
    Stats stats() {
        std::lock_guard<std::mutex> lock(mu);
        return st;
    }

    // solo():                               the caller's own one-unit call -> C_KZG_RET
    // copy_in(h_in, idx):                   place this caller's inputs as unit idx of the batch buffer
    // run(h_in, h_out, status, n):          the batch path over units 0..n-1; status[i] != 0 flags unit i
    // copy_out(h_out, idx, n):              fetch unit idx's results (called only if the unit succeeded)
    template <class Solo, class CopyIn, class Run, class CopyOut>
    C_KZG_RET submit(const void *key, size_t key_len, Solo &&solo, CopyIn &&copy_in, Run &&run, CopyOut &&copy_out) {
        std::unique_lock<std::mutex> lock(mu);
        st.calls++;
        Batch *b = nullptr;
        for (;;) {
            if (active < max_active) {
                // invariant: pending is empty here (a finishing launch promotes before it gives its place up)
                active++;
                st.solo++;
                lock.unlock();
                C_KZG_RET r = guarded([&]() -> C_KZG_RET { return solo(); });
                lock.lock();
                leader_done();
                return r;
            }
            for (Batch *p : pending) {
                if (p->n < max_batch && p->key.size() == key_len && (key_len == 0 || !memcmp(p->key.data(), key, key_len))) {
                    b = p;
                    break;
                }
            }
            if (b) break;
            b = fresh_batch();
            if (b) {
                b->key.assign((const uint8_t *)key, (const uint8_t *)key + key_len);
                pending.push_back(b);
                break;
            }
            if (all.empty()) {
                // no page-locked memory to be had at all: this call goes alone, unqueued
                lock.unlock();
                return guarded([&]() -> C_KZG_RET { return solo(); });
            }
            cv_pool.wait(lock);   // every batch buffer is in use: wait for one, or for a launch place
        }
        const size_t idx = b->n++;
        b->refs++;
        lock.unlock();
        copy_in(b->h_in, idx);
        b->copied.fetch_add(1, std::memory_order_release);
        lock.lock();
        if (idx == 0) {
            b->cv_owner.wait(lock, [&]() { return b->promoted; });
            const size_t n = b->n;   // final: a promoted batch is no longer in `pending`
            st.batches++;
            st.batched += n;
            if (n > st.largest) st.largest = n;
            lock.unlock();
            while (b->copied.load(std::memory_order_acquire) != n) std::this_thread::yield();   // members still copying in: microseconds
            memset(b->status.data(), 0, n);
            const auto t_run = std::chrono::steady_clock::now();
            C_KZG_RET r = guarded([&]() -> C_KZG_RET { return run((const uint8_t *)b->h_in, b->h_out, b->status.data(), n); });
            const uint64_t run_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_run).count();
            if (r == C_KZG_BADARGS) {
                // flagged units answer for themselves; without a flag the verdict concerns the whole batch
                for (size_t i = 0; i < n; i++) {
                    if (b->status[i]) {
                        r = C_KZG_OK;
                        break;
                    }
                }
            }
            lock.lock();
            st.run_us += run_us;
            b->ret = r;
            b->done = true;
            b->cv_done.notify_all();
            leader_done();
        } else {
            b->cv_done.wait(lock, [&]() { return b->done; });
        }
        const size_t n = b->n;
        const C_KZG_RET mine = b->status[idx] ? (C_KZG_RET)b->status[idx] : b->ret;
        lock.unlock();
        if (mine == C_KZG_OK) copy_out((const uint8_t *)b->h_out, idx, n);
        lock.lock();
        if (--b->refs == 0) recycle(b);
        return mine;
    }

   private:
    struct Batch {
        uint8_t *h_in = nullptr, *h_out = nullptr;   // page-locked
        std::vector<uint8_t> status, key;
        size_t n = 0, refs = 0;
        std::atomic<size_t> copied{0};
        bool promoted = false, done = false;
        C_KZG_RET ret = C_KZG_OK;
        std::condition_variable cv_owner, cv_done;
    };

    // mu held.  A launch has ended: hand its place to the oldest open batch, or give it up.
    void leader_done() {
        if (!pending.empty()) {
            Batch *nb = pending.front();
            pending.pop_front();
            nb->promoted = true;
            nb->cv_owner.notify_one();
        } else {
            active--;
            cv_pool.notify_all();
        }
    }

    // mu held.  A batch with buffers, from the free list or newly allocated; nullptr if neither is possible now.
    Batch *fresh_batch() {
        Batch *b = nullptr;
        if (!free_list.empty()) {
            b = free_list.back();
            free_list.pop_back();
        } else if ((int)all.size() < max_active + 2 && !alloc_failed) {
            b = new (std::nothrow) Batch();
            if (b) {
                // Portable: any device of a multi-device load may DMA from / into it
                bool ok = hipHostMalloc((void **)&b->h_in, in_bytes ? in_bytes : 1, hipHostMallocPortable) == hipSuccess;
                ok = ok && hipHostMalloc((void **)&b->h_out, out_bytes ? out_bytes : 1, hipHostMallocPortable) == hipSuccess;
                if (ok) {
                    try {
                        b->status.resize(max_batch);
                    } catch (...) {
                        ok = false;
                    }
                }
                if (!ok) {
                    (void)hipGetLastError();
                    if (b->h_in) (void)hipHostFree(b->h_in);
                    if (b->h_out) (void)hipHostFree(b->h_out);
                    delete b;
                    b = nullptr;
                    alloc_failed = true;   // do not try again on every call
                } else {
                    CKZG_TSAN_NEW_MEMORY(b->h_in, in_bytes);
                    CKZG_TSAN_NEW_MEMORY(b->h_out, out_bytes);
                    try {
                        all.push_back(b);
                    } catch (...) {
                        (void)hipHostFree(b->h_in);
                        (void)hipHostFree(b->h_out);
                        delete b;
                        b = nullptr;
                    }
                }
            }
        }
        if (b) {
            b->n = b->refs = 0;
            b->copied.store(0, std::memory_order_relaxed);
            b->promoted = b->done = false;
            b->ret = C_KZG_OK;
        }
        return b;
    }

    void recycle(Batch *b) {
        free_list.push_back(b);
        cv_pool.notify_all();
    }

    const size_t max_batch, in_bytes, out_bytes;
    const int max_active;
    std::mutex mu;
    std::condition_variable cv_pool;
    std::deque<Batch *> pending;       // open batches, oldest first
    std::vector<Batch *> all, free_list;
    int active = 0;                    // launches in flight (solo calls and batches)
    bool alloc_failed = false;
    Stats st;
};

}  // namespace api
}  // namespace ckzg
