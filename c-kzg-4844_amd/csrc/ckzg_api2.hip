// ckzg_api2.hip -- C-ABI entry points for point proofs, verification and recovery
// (src/eip4844/eip4844.c:313-844, src/eip7594/eip7594.c:177-974).  Host code here is protocol
// glue (transcripts, argument checks, the final two-pairing check); every MSM, NTT, polynomial
// evaluation and point validation is dispatched to the kernels in msm.hip / ntt.hip / fk20.hip /
// verify.hip.
#include <chrono>
#include <functional>
#include <thread>

#include "api_common.hpp"
#include <optional>
#include "combiner.hpp"

using namespace ckzg;
using namespace ckzg::host;
using namespace ckzg::api;

namespace {

template <class T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    bool alloc(size_t count) {
        n = count;
        return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)) == hipSuccess;
    }
    bool up(const T *h, size_t count) { return hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice) == hipSuccess; }
    bool down(T *h, size_t count) const { return hipMemcpy(h, p, count * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess; }
    ~DBuf() {
        if (p) (void)hipFree(p);
    }
};

#define RC(expr)                          \
    do {                                  \
        int _rc = (expr);                 \
        if (_rc) return (C_KZG_RET)_rc;   \
    } while (0)
#define OKB(expr)                         \
    do {                                  \
        if (!(expr)) return C_KZG_ERROR;  \
    } while (0)
#define OKM(expr)                         \
    do {                                  \
        if (!(expr)) return C_KZG_MALLOC; \
    } while (0)

// Where the Fiat-Shamir challenges of an n-blob batch are hashed (compute_challenge, eip4844.c:147-178: one SHA-256
// over 131,152 bytes per blob).  The host hashes them on T threads underneath the chunked blob copy, so that form costs
// max(copy, hash) before the tail; the GPU hash (k_sha256_challenges) needs the blobs in HBM first and the evaluation
// after it: copy + GPU_SHA_US (whatever n <= 65,536: 2,050 dependent compressions per blob) + evaluation.  T is this
// process's share of the host (host_thread_budget: cpus / ranks on the host) and the hash rate of one thread is
// MEASURED on first use (44-66 us per blob with the x86 SHA extensions, ~320 us without): a rank of an 8-GPU job in a
// 16-core container has 2 threads and hashes a 512-blob shard in 11-17 ms on the host, in 4.9 + 1.2 ms on the GPU; one
// process with 16 threads keeps a 4096-blob batch on the host (11 ms under a 9.8 ms copy, against 16.6 ms).
static constexpr double GPU_SHA_US = 3900.0;   // k_sha256_challenges at n <= 4096 with its SIMDs to itself (profiles/r06_cu_partition_ab.txt; 4.9 ms shared)
Fr challenge_from_bytes(const uint8_t *blob, const uint8_t *commitment48);   // below
static double host_sha_us_per_blob() {
    static const double us = []() {
        std::vector<uint8_t> blob(BYTES_PER_BLOB, 0x5a);
        uint8_t c48[48] = {0xc0};
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            Fr z = challenge_from_bytes(blob.data(), c48);
            const double dt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (z.l[0] == 0x12345678u) best += 1e-9;   // (keeps the hash alive)
            if (dt < best) best = dt;
        }
        return best < 10.0 ? 10.0 : best;
    }();
    return us;
}
static bool challenges_on_gpu(size_t n) {
    const int opt = g_gpu_sha_min.load();
    if (opt > 0) return n >= (size_t)opt;  // ckzg_hip_set_option("gpu_sha_min", n)
    if (n < 64) return false;              // a few waves: the GPU hash is pure latency
    size_t t = (size_t)host_thread_budget();
    if (t > 32) t = 32;
    const double host_us = (double)n * host_sha_us_per_blob() / (double)t;
    const double copy_us = (double)n * 2.4;     // 55 GB/s
    const double eval_us = (double)n * 0.21;    // k_eval_tree over the blobs' bytes (0.85 ms at n = 4096), after the GPU hash
    const double gpu_us = GPU_SHA_US * (double)((n + 65535) / 65536);
    return (host_us > copy_us ? host_us : copy_us) > copy_us + gpu_us + eval_us;
}

struct RawScalar {
    uint32_t l[8];
};

RawScalar raw_of(const Fr &a) {
    RawScalar r;
    to_raw<FrParams>(r.l, a);
    return r;
}

// every caller passes a validated point (commitment, proof, generator): GLV applies
G1Jac g1_mul_fr(const G1Jac &p, const Fr &k) {
    RawScalar r = raw_of(k);
    return host::g1_mul_glv_host(p, r.l);
}

// src/eip4844/eip4844.c:80-106
bool fr_batch_inv(Fr *out, const Fr *a, size_t len) {
    Fr acc = Fr::one();
    for (size_t i = 0; i < len; i++) {
        out[i] = acc;
        acc = mul(acc, a[i]);
    }
    if (acc.is_zero()) return false;
    acc = fr_inv(acc);
    for (size_t i = len; i-- > 0;) {
        out[i] = mul(out[i], acc);
        acc = mul(acc, a[i]);
    }
    return true;
}

Fr fr_pow_u64(Fr a, uint64_t n) {
    Fr out = Fr::one();
    while (true) {
        if (n & 1) out = mul(out, a);
        if ((n >>= 1) == 0) break;
        a = sqr(a);
    }
    return out;
}

// src/eip4844/blob.c:31-38
C_KZG_RET blob_to_polynomial(Fr *p, const Blob *blob) {
    for (size_t i = 0; i < FIELD_ELEMENTS_PER_BLOB; i++) {
        if (!fr_from_bytes_canonical(p[i], blob->bytes + 32 * i)) return C_KZG_BADARGS;
    }
    return C_KZG_OK;
}

// src/eip4844/eip4844.c:192-240 (host form, used where the inverses are needed anyway)
C_KZG_RET evaluate_host(Fr &out, const Fr *poly, const Fr &x, const KZGSettings *s) {
    const size_t n = FIELD_ELEMENTS_PER_BLOB;
    const Fr *dom = as_fr(s->brp_roots_of_unity);
    std::vector<Fr> den(n), inv(n);
    for (size_t i = 0; i < n; i++) {
        if (x == dom[i]) {
            out = poly[i];
            return C_KZG_OK;
        }
        den[i] = sub(x, dom[i]);
    }
    if (!fr_batch_inv(inv.data(), den.data(), n)) return C_KZG_BADARGS;
    Fr acc = Fr::zero();
    for (size_t i = 0; i < n; i++) acc = add(acc, mul(mul(inv[i], dom[i]), poly[i]));
    acc = mul(acc, fr_inv(fr_from_u64(n)));
    out = mul(acc, sub(fr_pow_u64(x, n), Fr::one()));
    return C_KZG_OK;
}

// "FSBLOBVERIFY_V1_" | u64be 0 | u64be 4096 | blob | commitment  (eip4844.c:147-178)
Fr challenge_from_bytes(const uint8_t *blob, const uint8_t *commitment48) {
    Sha256 h;
    uint8_t head[32], out[32];
    memcpy(head, "FSBLOBVERIFY_V1_", 16);
    be64(head + 16, 0);
    be64(head + 24, FIELD_ELEMENTS_PER_BLOB);
    h.update(head, 32);
    h.update(blob, BYTES_PER_BLOB);
    h.update(commitment48, 48);
    h.finish(out);
    return fr_from_bytes_reduce(out);
}

// The reference checks e(C - [y]G1, G2) == e(proof, [s]G2 - [z]G2) (eip4844.c:359-383).  Moving
// [z]proof to the G1 side gives the equivalent e(C - [y]G1 + [z]proof, G2) == e(proof, [s]G2),
// whose G2 arguments are setup constants with precomputed line tables (host_pairing.hpp).
// [k]G1 from the generator table of host_pairing.hpp (64 additions, no doubling)
G1Jac g1_gen_mul_fr(const Fr &k) {
    RawScalar r = raw_of(k);
    return host::g1_gen_mul(r.l);
}

// The half of the check that does not depend on the evaluation y: [z]proof and the Miller loop of
// e(-proof, [s]G2).  verify_blob_kzg_proof runs it on the host while the GPU evaluates the polynomial.
struct ProofSide {
    G1Jac zp;
    host::Fp12 miller;
};
ProofSide verify_proof_side(const Fr &z, const G1Jac &proof, const PreparedG2 *pg) {
    ProofSide ps;
    ps.zp = g1_mul_fr(proof, z);
    ps.miller = host::miller_product_prepared(jac_to_affine_fast(jac_neg(proof)), pg->s1, G1Affine::inf(), pg->s1);
    return ps;
}
bool verify_with_proof_side(const G1Jac &commitment, const Fr &y, const ProofSide &ps, const PreparedG2 *pg) {
    G1Jac lhs = jac_add(jac_add(commitment, jac_neg(g1_gen_mul_fr(y))), ps.zp);
    host::Fp12 f = host::miller_product_prepared(jac_to_affine_fast(lhs), pg->gen, G1Affine::inf(), pg->gen);
    return host::final_exp(host::mul(f, ps.miller)).is_one();
}

bool verify_kzg_proof_impl(const G1Jac &commitment, const Fr &z, const Fr &y, const G1Jac &proof,
                           const PreparedG2 *pg) {
    G1Jac lhs = jac_add(jac_add(commitment, jac_neg(g1_gen_mul_fr(y))), g1_mul_fr(proof, z));
    return pairing_product_is_one(jac_to_affine_fast(lhs), pg->gen, jac_to_affine_fast(jac_neg(proof)), pg->s1);
}

// quotient polynomial and its commitment (eip4844.c:417-494); the 4096-term MSM runs on the GPU
C_KZG_RET compute_kzg_proof_impl(KZGProof *proof_out, Fr &y_out, const Fr *poly, const Fr &z,
                                 const KZGSettings *s, dev::DeviceCtx *ctx) {
    const size_t n = FIELD_ELEMENTS_PER_BLOB;
    const Fr *dom = as_fr(s->brp_roots_of_unity);
    C_KZG_RET ret = evaluate_host(y_out, poly, z, s);
    if (ret != C_KZG_OK) return ret;
    std::vector<Fr> den(n), inv(n), q(n);
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (z == dom[i]) {
            m = i + 1;
            den[i] = Fr::one();
            q[i] = Fr::zero();
            continue;
        }
        q[i] = sub(poly[i], y_out);
        den[i] = sub(dom[i], z);
    }
    if (!fr_batch_inv(inv.data(), den.data(), n)) return C_KZG_BADARGS;
    for (size_t i = 0; i < n; i++) q[i] = mul(q[i], inv[i]);
    if (m != 0) {
        m--;
        q[m] = Fr::zero();
        for (size_t i = 0; i < n; i++) {
            if (i == m) continue;
            den[i] = mul(sub(z, dom[i]), z);
        }
        den[m] = Fr::one();
        if (!fr_batch_inv(inv.data(), den.data(), n)) return C_KZG_BADARGS;
        for (size_t i = 0; i < n; i++) {
            if (i == m) continue;
            q[m] = add(q[m], mul(mul(sub(poly[i], y_out), dom[i]), inv[i]));
        }
    }
    std::vector<RawScalar> raw(n);
    for (size_t i = 0; i < n; i++) raw[i] = raw_of(q[i]);
    DBuf<RawScalar> d_sc;
    DBuf<uint8_t> d_out;
    OKM(d_sc.alloc(n) && d_out.alloc(48));
    OKB(d_sc.up(raw.data(), n));
    RC(dev::msm_commit_table_raw_device(ctx, d_out.p, (const uint32_t *)d_sc.p, 1));
    OKB(d_out.down(proof_out->bytes, 48));
    return C_KZG_OK;
}

// fn(0) ... fn(n - 1) on up to 32 threads: the caller takes part, the rest are jobs on the process-wide worker pool
// (no thread is created or joined per call).  fn must not throw and must not itself wait for pool jobs -- a pool
// worker never blocks on another pool job; the callers of this function are never pool workers.
void parallel_for(size_t n, const std::function<void(size_t)> &fn) {
    size_t nt = (size_t)host_thread_budget();
    if (nt > 32) nt = 32;
    if (nt > n) nt = n;
    if (nt <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    struct Shared {
        std::atomic<size_t> next{0};
        std::atomic<uint32_t> active{0};   // pool jobs of this call that have not returned yet (futex word)
    } sh;
    auto loop = [&sh, &fn, n]() {
        for (;;) {
            const size_t i = sh.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            fn(i);
        }
    };
    for (size_t t = 1; t < nt; t++) {
        sh.active.fetch_add(1, std::memory_order_relaxed);
        if (!WorkerPool::get().submit([&sh, loop]() {
                loop();
                if (sh.active.fetch_sub(1, std::memory_order_acq_rel) == 1) futex_wake(&sh.active, INT_MAX);
            }))
            sh.active.fetch_sub(1, std::memory_order_relaxed);
    }
    loop();
    wait_host_work_done(&sh.active, "parallel_for jobs");   // (asleep, not spinning: the stragglers may need this core)
}

// Several variable-base lincombs sum_i k_i P_i in ONE launch.  Job j takes n points starting at
// d_pts + pt_off[j] and its own scalar vector; every job is padded to a multiple of 64 lanes so
// that a workgroup's partial sum belongs to exactly one job.
struct LincombJob {
    const G1Affine *d_pts;
    const std::vector<RawScalar> *k;       // scalars on the host, or ...
    const RawScalar *d_k = nullptr;        // ... already in HBM (k == nullptr), n_dev of them
    size_t n_dev = 0;
    size_t size() const { return k ? k->size() : n_dev; }
};

// Which kernels compute a call's variable-base sums: 1 = one GLV ladder per term (verify.hip), 2 = bucket
// accumulation (pippenger.hip).  Both end in ~128 sequential doublings of one lane (~1.4 ms: inherent to a
// 128-bit scalar), and at the sizes this path sees (n <= ~10^4 cells or blobs per call) the ladders still fit
// the chip in one or two rounds of waves, so the buckets' smaller operation count does not show: measured
// A/B inside verify_cell_kzg_proof_batch (tools/bench_lincomb.sh) the ladders win at
// every n up to 65,536.  The default is therefore the ladders; CKZG_HIP_BUCKET_MIN=n (or algo = 2 at the
// ckzg_hip_g1_lincomb boundary) routes sums of at least n terms to the buckets.
static size_t bucket_min_terms() {
    static const long v = dev::ab_knob("CKZG_HIP_BUCKET_MIN", -1);   // never, unless an A/B build says otherwise
    return v < 0 ? ~(size_t)0 : (size_t)v;
}

C_KZG_RET gpu_lincomb_multi(dev::DeviceCtx *ctx, G1Jac *outs, const LincombJob *jobs, int njobs, int algo = 0) {
    size_t total = 0, max_job = 0;
    std::vector<size_t> off(njobs), nb(njobs);
    for (int j = 0; j < njobs; j++) {
        off[j] = total;
        nb[j] = (jobs[j].size() + 63) / 64;
        if (nb[j] == 0) nb[j] = 1;
        total += nb[j] * 64;
        if (jobs[j].size() > max_job) max_job = jobs[j].size();
    }
    if (algo == 0) {
        static const int forced = (int)dev::ab_knob("CKZG_HIP_LINCOMB", 0);
        algo = forced ? forced : (max_job >= bucket_min_terms() ? 2 : 1);
    }
    // ladders: four lanes per half-term (k_lincomb_partial_quad) while 8 lanes per term still fit the chip in
    // about two waves per SIMD; beyond that the one-lane-per-half form does fewer lane-products in total
    // (algo 3 / 4 force the one-lane / four-lane ladders)
    static const size_t quad_max = (size_t)dev::ab_knob("CKZG_HIP_QUAD_MAX", 8192);
    const bool quad = algo == 4 || (algo == 1 && total <= quad_max);
    if (algo == 3 || algo == 4) algo = 1;
    const int wbits = dev::bucket_msm_wbits(max_job);
    const size_t bucket_scratch = algo == 2 ? dev::bucket_msm_scratch_bytes(total, njobs, wbits) : 0;
    Arena &ar = ctx->lc_arena;
    OKM(ar.begin(total * (sizeof(RawScalar) + sizeof(G1Affine)) + (total / 8) * sizeof(G1XYZZ) +
                 njobs * sizeof(G1Affine) + (njobs + 1) * 4 + bucket_scratch + 1024));
    struct { RawScalar *p; } d_k = {ar.get<RawScalar>(total)};
    struct { G1Affine *p; } d_p = {ar.get<G1Affine>(total)}, d_out = {ar.get<G1Affine>(njobs)};
    struct { G1XYZZ *p; } d_part = {ar.get<G1XYZZ>(total / 8)};  // one partial per 64 lanes = 32 terms (8 in quad form)
    uint32_t *d_off = ar.get<uint32_t>(njobs + 1);
    uint8_t *d_bucket = algo == 2 ? ar.get<uint8_t>(bucket_scratch) : nullptr;
    OKM(d_k.p && d_p.p && d_out.p && d_part.p && d_off && (algo != 2 || d_bucket));
    OKB(hipMemsetAsync(d_p.p, 0, total * sizeof(G1Affine), ctx->stream) == hipSuccess);  // (0,0) = infinity
    OKB(hipMemsetAsync(d_k.p, 0, total * sizeof(RawScalar), ctx->stream) == hipSuccess);
    for (int j = 0; j < njobs; j++) {
        size_t n = jobs[j].size();
        OKB(hipMemcpyAsync(d_p.p + off[j], jobs[j].d_pts, n * sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess);
        if (jobs[j].k) {
            OKB(hipMemcpyAsync(d_k.p + off[j], jobs[j].k->data(), n * sizeof(RawScalar), hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
        } else {
            OKB(hipMemcpyAsync(d_k.p + off[j], jobs[j].d_k, n * sizeof(RawScalar), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess);
        }
    }
    if (algo == 2) {
        std::vector<uint32_t> job_off(njobs + 1);
        for (int j = 0; j < njobs; j++) job_off[j] = (uint32_t)off[j];
        job_off[njobs] = (uint32_t)total;
        RC(dev::bucket_msm_enqueue(ctx, d_out.p, d_p.p, (const uint32_t *)d_k.p, total, job_off.data(), njobs, wbits, d_bucket));
        OKB(dev::sync_stream(ctx->stream) == hipSuccess);
    } else {
        const size_t per = quad ? 8 : 32;
        std::vector<uint32_t> part_off(njobs + 1);
        for (int j = 0; j < njobs; j++) part_off[j] = (uint32_t)(off[j] / per);
        part_off[njobs] = (uint32_t)(total / per);
        RC(dev::lincomb_multi_device(ctx, d_out.p, d_part.p, d_off, d_p.p, (const uint32_t *)d_k.p, total, part_off.data(), njobs, quad));
    }
    std::vector<G1Affine> res(njobs);
    OKB(hipMemcpy(res.data(), d_out.p, njobs * sizeof(G1Affine), hipMemcpyDeviceToHost) == hipSuccess);
    for (int j = 0; j < njobs; j++) outs[j] = jac_from_affine(res[j]);
    return C_KZG_OK;
}

// sum_i k_i P_i on the host, for the handful of points of a small call
G1Jac host_lincomb(const std::vector<G1Jac> &pts, const std::vector<Fr> &k) {
    G1Jac acc = G1Jac::inf();
    for (size_t i = 0; i < pts.size(); i++) acc = jac_add(acc, g1_mul_fr(pts[i], k[i]));
    return acc;
}

// Below this many blobs the few G1 scalar multiplications of a verification (point validation,
// random-linear-combination sums) stay on the host next to the pairing: a single 255-bit scalar
// multiplication is ~0.25 ms on a CPU core but ~1 ms of dependent latency even on four GPU lanes.  The
// data-parallel part (bytes -> Fr, 4096-term evaluation) runs on the GPU for every n.
// Measured (tools/bench_verify_small.py): host path 1.4 / 2.0 / 2.5 / 3.1 ms for n = 1 / 2 / 3 / 4, GPU path
// 2.6 ms for any n up to ~16 -> hand-over after 3.
constexpr uint64_t SMALL_VERIFY_N = 3;

// The batch challenge's transcript (eip4844.c:637-664: one SHA-256 stream over every C_i, z_i, y_i, proof_i) hashed
// while the batch is still in flight: the pipelined form downloads each chunk's evaluations into page-locked memory
// right behind the kernel that produced them, and this thread feeds them to the hash in order as they land.  When
// the last chunk has been evaluated only its own 256 entries are left to hash (~0.03 ms) instead of all of them
// (0.49 ms at n = 4096, after the last byte and before everything that needs r).
struct TranscriptHasher {
    Sha256 h;
    // chunks whose download has been enqueued (its event recorded); futex word.  ABORT in the same word: a separate
    // flag checked before futex_wait could be set (and the wake sent) between the check and the wait, and the job would
    // sleep for good with `published` unchanged -- the destructor then hangs in wait() (round-4 advisor finding).
    std::atomic<uint32_t> published{0};
    static constexpr uint32_t ABORT = UINT32_MAX;
    bool failed = false;
    std::atomic<uint32_t> running{0};     // 1 while the pool job has not returned; futex word
    bool on_pool = false;
    std::thread t;                      // only when the worker pool did not take the job
    TranscriptHasher() = default;
    TranscriptHasher(const TranscriptHasher &) = delete;
    TranscriptHasher &operator=(const TranscriptHasher &) = delete;
    void wait() {
        if (on_pool) {
            wait_host_work_done(&running, "transcript hasher job");   // (the job's own waits are bounded: it ends)
        } else if (t.joinable()) {
            t.join();
        }
    }
    ~TranscriptHasher() {
        published.store(ABORT, std::memory_order_release);
        futex_wake(&published, INT_MAX);   // the job may be asleep waiting for the next chunk
        wait();
    }
    void start(int device, size_t n, size_t chunk, const hipEvent_t *landed, const Bytes48 *cb, const Bytes48 *pb,
               const Fr *z, const Fr *h_y) {
        uint8_t head[32];
        memcpy(head, "RCKZGBATCH___V1_", 16);
        be64(head + 16, FIELD_ELEMENTS_PER_BLOB);
        be64(head + 24, n);
        h.update(head, 32);
        auto body = [=]() {
            if (hipSetDevice(device) != hipSuccess) {
                failed = true;
                return;
            }
            const size_t nch = (n + chunk - 1) / chunk;
            uint8_t zb[64];
            for (size_t c = 0; c < nch; c++) {
                // asleep until the caller publishes the next chunk (12 ms of a spinning pool worker per pipelined
                // verification otherwise)
                if (!wait_word_until(&published, [c](uint32_t v) { return v > c; }, "transcript hasher: next chunk published")) {
                    failed = true;
                    return;
                }
                if (published.load(std::memory_order_acquire) == ABORT) return;   // the owner is being destroyed: nobody will ask for the digest
                if (dev::sync_event(landed[c]) != hipSuccess) {
                    failed = true;
                    return;
                }
                const size_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
                for (size_t i = lo; i < hi; i++) {
                    h.update(cb[i].bytes, 48);
                    fr_to_bytes(zb, z[i]);
                    fr_to_bytes(zb + 32, h_y[i]);
                    h.update(zb, 64);
                    h.update(pb[i].bytes, 48);
                }
            }
        };
        // (a pool job may wait for the GPU and for the publishing caller, never for another pool job)
        running.store(1, std::memory_order_relaxed);
        on_pool = WorkerPool::get().submit([this, body]() {
            body();
            running.store(0, std::memory_order_release);
            futex_wake(&running, INT_MAX);
        });
        if (!on_pool) {
            running.store(0, std::memory_order_relaxed);
            t = std::thread(body);
        }
    }
    void publish(size_t chunks) {
        published.store((uint32_t)chunks, std::memory_order_release);
        futex_wake(&published, INT_MAX);
    }
    bool finish(uint8_t digest[32]) {
        wait();
        if (failed) return false;
        h.finish(digest);
        return true;
    }
};

// Host threads that hash the blobs' Fiat-Shamir challenges IN ORDER and say how far they are: the pipelined form
// of the batch verification enqueues chunk c's evaluation as soon as the first (c + 1) * chunk challenges exist,
// while later blobs are still crossing PCIe.  (compute_challenge, eip4844.c:147-178)
struct OrderedHasher {
    std::atomic<size_t> next{0};
    std::vector<std::atomic<uint32_t>> done;   // per chunk: blobs hashed
    std::atomic<uint32_t> active{0};           // pool jobs of this call that have not returned yet (futex word)
    size_t n, chunk;
    JoinThreads th;                            // only when the process-wide pool could not be used
    OrderedHasher(size_t n_, size_t chunk_) : done((n_ + chunk_ - 1) / chunk_), n(n_), chunk(chunk_) {
        for (auto &d : done) d.store(0);
    }
    OrderedHasher(const OrderedHasher &) = delete;
    OrderedHasher &operator=(const OrderedHasher &) = delete;
    // nothing of this object (or of z / blobs / cb) may be touched by a worker once the call has returned
    ~OrderedHasher() {
        next.store(n, std::memory_order_relaxed);   // an abandoned call: the workers stop at their next blob
        wait_host_work_done(&active, "challenge hashing jobs");
    }
    void start(Fr *z, const Blob *blobs, const Bytes48 *cb) {
        size_t nt = (size_t)host_thread_budget();
        if (nt > 32) nt = 32;
        auto loop = [this, z, blobs, cb]() {
            for (;;) {
                const size_t i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= n) return;
                z[i] = challenge_from_bytes(blobs[i].bytes, cb[i].bytes);
                const size_t c = i / chunk, want = n - c * chunk < chunk ? n - c * chunk : chunk;
                if (done[c].fetch_add(1, std::memory_order_acq_rel) + 1 == want) futex_wake(&done[c], INT_MAX);   // chunk complete
            }
        };
        for (size_t t = 0; t < nt; t++) {
            active.fetch_add(1, std::memory_order_relaxed);
            if (!WorkerPool::get().submit([this, loop]() {
                    loop();
                    if (active.fetch_sub(1, std::memory_order_acq_rel) == 1) futex_wake(&active, INT_MAX);
                })) {
                active.fetch_sub(1, std::memory_order_relaxed);
                try {
                    th.spawn(loop);
                } catch (...) {   // no pool and no thread: the last resort hashes here, before the pipeline starts
                    if (t == 0) loop();
                }
            }
        }
    }
    // false: the chunk's challenges were not there by the deadline (the call fails; the destructor stops the workers)
    bool wait_chunk(size_t c) const {
        const size_t lo = c * chunk, want = (n - lo < chunk ? n - lo : chunk);
        auto *w = const_cast<std::atomic<uint32_t> *>(&done[c]);
        return wait_word_until(w, [want](uint32_t v) { return v >= want; }, "challenge hashing: chunk complete");
    }
};

// The masked streams of the compute-unit partition (device.hpp: sha_stream / side_stream), made on first use: a quarter of
// the compute units for the SHA-256 chain, the rest for what runs underneath it.  The mask bits of a multi-XCD part are
// dealt round the XCDs, so "the first quarter of the bits" is the same share of every XCD, which is where the workgroups
// of a launch go as well.  false (and plain streams) where the runtime refuses.
// Verifications with a GPU hash in flight per device: the partition confines EVERY such call's hash to the same quarter
// of the compute units, so a second concurrent one would crowd the first where, unpartitioned, it spreads over the
// chip.  Only a call that finds no other takes the partition.
static std::atomic<int> g_gpu_hash_calls[64];
struct GpuHashCall {
    int dev;
    bool alone;
    explicit GpuHashCall(int d) : dev(d >= 0 && d < 64 ? d : 0), alone(g_gpu_hash_calls[dev].fetch_add(1, std::memory_order_acq_rel) == 0) {}
    ~GpuHashCall() { g_gpu_hash_calls[dev].fetch_sub(1, std::memory_order_acq_rel); }
    GpuHashCall(const GpuHashCall &) = delete;
    GpuHashCall &operator=(const GpuHashCall &) = delete;
};

static bool ensure_cu_partition(dev::DeviceCtx *ctx) {
    if (ctx->cu_partition_tried) return ctx->sha_stream != nullptr;
    ctx->cu_partition_tried = true;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 64 || cus > 1024) {
        (void)hipGetLastError();
        return false;
    }
    const int words = (cus + 31) / 32, quarter = cus / 4;
    uint32_t sha_mask[32] = {}, side_mask[32] = {};
    for (int b = 0; b < cus; b++) (b < quarter ? sha_mask : side_mask)[b >> 5] |= 1u << (b & 31);
    hipStream_t made[3] = {nullptr, nullptr, nullptr};
    bool good = hipExtStreamCreateWithCUMask(&made[0], (uint32_t)words, sha_mask) == hipSuccess &&
                hipExtStreamCreateWithCUMask(&made[1], (uint32_t)words, side_mask) == hipSuccess &&
                hipExtStreamCreateWithCUMask(&made[2], (uint32_t)words, side_mask) == hipSuccess;
    if (!good) {
        (void)hipGetLastError();
        for (auto st : made) {
            if (st) (void)hipStreamDestroy(st);
        }
        return false;
    }
    ctx->sha_stream = made[0];
    ctx->side_stream[0] = made[1];
    ctx->side_stream[1] = made[2];
    return true;
}


// Shared core of verify_blob_kzg_proof and verify_blob_kzg_proof_batch (eip4844.c:537-595,
// 697-844).  Per blob, on the GPU: point validation, bytes -> Fr, evaluation at the challenge;
// then three lincombs over all blobs; host: transcripts and the pairing check
//   e(sum r^i proof_i, [s]G2) == e(sum r^i (C_i - [y_i]G1) + sum r^i z_i proof_i, G2).
// Three forms of the per-blob stage:
//   * classic (n < 1024, or challenges hashed on the GPU): one copy of all blobs, then the kernels;
//   * pipelined (host pointers, n >= 1024): the blobs cross PCIe in chunks on the copy stream -- DMA'd in place
//     from page-locked caller memory, through pinned staging otherwise -- while earlier chunks are converted and
//     evaluated and host threads hash the challenges in order, so that only the transcript, the three sums and
//     the pairing follow the last byte;
//   * resident (ckzg_hip_verify_blob_kzg_proof_batch_device): blobs, commitments and proofs already in HBM, the
//     challenges hashed by k_sha256_challenges; only 96 + 64 bytes per blob travel to the host for the transcript.
C_KZG_RET verify_blobs_core(bool *ok, const Blob *blobs, const Bytes48 *cb, const Bytes48 *pb, uint64_t n,
                            const KZGSettings *s, dev::DeviceCtx *ctx, bool resident = false) {
    const bool small = !resident && n <= SMALL_VERIFY_N;
    Trace tr("verify_blobs");
    std::vector<G1Jac> hc, hp;  // host copies of the validated points (small n)
    if (small) {
        hc.resize(n);
        hp.resize(n);
        for (size_t i = 0; i < n; i++) {
            if (validate_kzg_g1(hc[i], cb[i].bytes) != C_KZG_OK) return C_KZG_BADARGS;
            if (validate_kzg_g1(hp[i], pb[i].bytes) != C_KZG_OK) return C_KZG_BADARGS;
        }
    }
    tr.mark("host point validation");
    // (never for the small path: its commitments are validated on the host and are not in d_ptb)
    const bool gpu_sha = resident || (!small && challenges_on_gpu(n));
    const size_t pipe_min = (size_t)g_verify_pipe_min.load(std::memory_order_relaxed);   // option "verify_pipe_min"
    const bool piped = !resident && !small && !gpu_sha && n >= pipe_min;
    // Call-time table (msm.hip): while the blobs of a batch cross PCIe (or, in the resident form, while one lane per
    // blob hashes them) the GPU is mostly idle and the batch challenge does not exist yet, so the 128 doublings per
    // term of the three sums are done early, as a narrow fixed-base table over the 2n validated points built on a side
    // stream.  In accumulator form (X28: no inversions) the build is the window-base ladders on quad lanes, ~1 ms of
    // latency whatever n, plus one chain of full additions per (window, point): 2.3 ms for n = 4096, where the affine
    // form it replaced took 7.7 ms and only paid from 2560 blobs.  Measured with the table off / on
    // (profiles/r03_verify_x28_sweep.txt), page-locked source: n = 1024 5.30 -> 4.41 ms, 2048 8.09 -> 7.17,
    // 4096 13.8 -> 11.9, and in the one-copy form of smaller batches n = 8 2.48 -> 2.25, 64 3.00 -> 2.33,
    // 512 3.81 -> 3.29, 768 4.76 -> 3.70; resident: 1024 9.4 -> 8.4, 4096 12.6 -> 11.4.  From 8 blobs upwards (below
    // that: the host path of SMALL_VERIFY_N, or ladders).
    // (option "verify_call_table" = 0 keeps the ladder sums: the path a device too full for the table takes)
    static const int call_table_wbits = (int)dev::ab_knob("CKZG_HIP_VERIFY_TABLE_WBITS", 6);
    static const size_t call_table_min = (size_t)dev::ab_knob("CKZG_HIP_VERIFY_TABLE_MIN", 8);
    bool use_table = g_verify_call_table.load(std::memory_order_relaxed) != 0 && call_table_wbits >= 4 && call_table_wbits <= 10 &&
                     n >= call_table_min && !small;
    dev::FixedBaseTable tbl;
    size_t tbl_bytes = 0, tbl_tmp = 0, sums_scratch = 0;
    Arena &ar = ctx->api_arena;
    // (no converted polynomials: the evaluation reads the blobs' bytes -- verify.hip: k_eval_tree's BYTES form)
    const size_t plain_bytes = (resident ? n * 160 : n * BYTES_PER_BLOB) + 2 * n * sizeof(Fr) + n * 4 +
                               2 * n * (48 + 2 + sizeof(G1Affine)) + 8192;   // (resident: the transcript rows, no blobs)
    if (use_table) {
        dev::call_table_geometry(&tbl, (int)(2 * n), call_table_wbits);
        tbl_bytes = dev::call_table_bytes(tbl);
        tbl_tmp = dev::call_table_tmp_bytes(tbl);
        sums_scratch = dev::table_sums_scratch_bytes(tbl, 3);
        // the table is an optimisation (1.3 GB at n = 4096): a device too full for it still verifies, by ladders
        if (!ar.begin(plain_bytes + tbl_bytes + tbl_tmp + sums_scratch + 6 * n * 32 + 1024)) {
            use_table = false;
            tbl_bytes = tbl_tmp = sums_scratch = 0;
        }
    }
    if (!use_table) OKM(ar.begin(plain_bytes));
    ABuf<uint8_t> d_ptb(ar, 2 * n * 48), d_st(ar, 2 * n), d_st2(ar, 2 * n), d_blobs_own(ar, resident ? 1 : n * BYTES_PER_BLOB);
    ABuf<G1Affine> d_pts(ar, 2 * n);
    ABuf<Fr> d_z(ar, n), d_y(ar, n);
    ABuf<uint32_t> d_bad(ar, n);
    OKM(d_ptb.p && d_st.p && d_st2.p && d_blobs_own.p && d_pts.p && d_z.p && d_y.p && d_bad.p);
    ABuf<uint8_t> d_tbl(ar, use_table ? tbl_bytes : 1), d_tbl_tmp(ar, use_table ? tbl_tmp : 1), d_sums_scr(ar, use_table ? sums_scratch : 1);
    ABuf<uint32_t> d_sc(ar, use_table ? 6 * n * 8 : 1);
    ABuf<G1XYZZ> d_sums(ar, 3);
    OKM(d_tbl.p && d_tbl_tmp.p && d_sums_scr.p && d_sc.p && d_sums.p);
    ArenaTrim trim(ar);
    tr.mark("arena");
    // resident: the device copies of the inputs ARE the caller's buffers, and nothing of them is read on the host -- the
    // rows of the batch transcript (commitment | z | y | proof) are assembled on the device and come back in one copy
    const uint8_t *d_blob_bytes = resident ? reinterpret_cast<const uint8_t *>(blobs) : d_blobs_own.p;
    const Bytes48 *d_cb = cb, *d_pb = pb;
    ABuf<uint8_t> d_rows(ar, resident ? n * 160 : 1);
    OKM(d_rows.p);
    if (resident) {
        OKM(ensure_pinned(ctx->h_out, ctx->h_out_bytes, n * 160));   // rows | evaluations + flags
        cb = pb = nullptr;
        blobs = nullptr;   // never dereferenced on the host
        OKB(hipEventRecord(ctx->ev[1], ctx->stream) == hipSuccess);
    }
    // whatever path leaves this function, the second stream must be idle before the arena is reused
    // (declared before the first enqueue on it: an early error return drains it too)
    struct StreamDrain {
        hipStream_t s;
        ~StreamDrain() {
            if (s) (void)dev::sync_stream(s);
        }
    } drain{ctx->copy_stream};
    const bool split_validation = !small && !piped && !resident && n < 1024;
    // Resident form: the hash chain on its own quarter of the compute units, validation and table build on the rest.
    // Measured (tools/ubench/sha_contention_probe.py, profiles/r06_cu_partition_ab.txt): sharing the chip, the chain of a
    // 4096-blob batch takes 4.9 ms -- a validation or table wave that lands on a hash wave's SIMD takes issue slots from
    // it for as long as it lives, and the launch ends with its slowest wave -- against 3.8 ms alone; partitioned, the
    // call goes 7.5 -> 6.3 ms (2048 blobs: 6.7 -> 5.7).  Below ~640 blobs the side work rarely collides (768 blobs: 3 calls of 16 took the extra 1.1 ms) and the
    // masked streams only cost their 0.07 ms.
    static const size_t partition_min = (size_t)dev::ab_knob("CKZG_HIP_CU_PARTITION_MIN", 640);
    // (host-pointer batches whose challenges are hashed on the GPU -- a rank with few host threads -- partition the same
    // way: there the validation runs before the copy on the main stream, hash and table build after it)
    // (up to 8192 blobs: 128 hash workgroups of two waves have a SIMD per wave on a quarter of 256 compute units; a
    // larger batch would crowd the quarter and spreads over the chip instead)
    const bool partition_wanted = gpu_sha && !small && n >= partition_min && n <= 8192 &&
                                  g_verify_cu_partition.load(std::memory_order_relaxed) != 0;
    std::optional<GpuHashCall> hash_call;   // (counted only by calls that could partition; lives until the call returns)
    if (partition_wanted) hash_call.emplace(ctx->device);
    const bool partition = partition_wanted && hash_call->alone && ensure_cu_partition(ctx);
    StreamDrain drain_sha{partition ? ctx->sha_stream : nullptr}, drain_val{partition ? ctx->side_stream[0] : nullptr};
    // The hash of the challenges: the enqueue-only step every GPU-hash form shares.  The resident form runs it HERE --
    // its inputs are the caller's buffers, nothing precedes it, and every 10 us the longest kernel of the call starts
    // earlier is 10 us off the call --, the host-pointer form further down, behind its copies.
    bool hash_enqueued = false;
    auto enqueue_hash = [&]() -> C_KZG_RET {
        RC(dev::sha256_challenges_device(ctx, d_z.p, d_blob_bytes, resident ? reinterpret_cast<const uint8_t *>(d_cb) : d_ptb.p, n,
                                         partition ? ctx->sha_stream : nullptr));
        if (partition) {
            // (stage_ev[1] is free in these forms: only the split validation of small host-pointer batches records it)
            if (!ctx->stage_ev[1]) OKB(hipEventCreateWithFlags(&ctx->stage_ev[1], hipEventDisableTiming) == hipSuccess);
            OKB(hipEventRecord(ctx->stage_ev[1], ctx->sha_stream) == hipSuccess);
            OKB(hipStreamWaitEvent(ctx->stream, ctx->stage_ev[1], 0) == hipSuccess);   // the challenges, before the evaluation
        }
        hash_enqueued = true;
        return C_KZG_OK;
    };
    if (resident && gpu_sha) RC(enqueue_hash());
    if (!small) {
        // commitments [0,n), proofs [n,2n): decompress + subgroup-check on the GPU, on the second stream
        // so that the ladder kernel runs under the (host-blocking, pageable) copy of the blobs
        // (only for batches whose blob copy is short: measured, a concurrent kernel slows a long pageable
        // copy by more than the ~1.3 ms it hides -- n = 4096: 30 -> 37 ms; n = 64: 8.0 -> 5.6 ms)
        // Below 1024 blobs the validation is also split: event 0 marks the decompressed points, the
        // subgroup test (event 1) keeps running on the second stream underneath evaluation and sums.
        // (resident inputs: nothing blocks the host, so the validation ladders always run on the second stream,
        // underneath the challenge hashing -- the longest kernel of that form -- and the evaluation)
        hipStream_t vs = (split_validation || resident) ? ctx->copy_stream : ctx->stream;
        if (partition && resident) vs = ctx->side_stream[0];
        if (!ctx->stage_ev[0]) OKB(hipEventCreateWithFlags(&ctx->stage_ev[0], hipEventDisableTiming) == hipSuccess);
        if (!ctx->stage_ev[1]) OKB(hipEventCreateWithFlags(&ctx->stage_ev[1], hipEventDisableTiming) == hipSuccess);
        const hipMemcpyKind kind = resident ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        OKB(hipMemcpyAsync(d_ptb.p, d_cb, n * 48, kind, vs) == hipSuccess);
        OKB(hipMemcpyAsync(d_ptb.p + n * 48, d_pb, n * 48, kind, vs) == hipSuccess);
        if (split_validation) {
            RC(dev::decompress_g1_batch_device(ctx, d_pts.p, d_st.p, d_ptb.p, 2 * n, vs));
            OKB(hipEventRecord(ctx->stage_ev[0], vs) == hipSuccess);
            RC(dev::subgroup_g1_batch_device(ctx, d_st2.p, d_pts.p, 2 * n, vs));
            OKB(hipEventRecord(ctx->stage_ev[1], vs) == hipSuccess);
        } else {
            RC(dev::validate_g1_batch_device(ctx, d_pts.p, d_st.p, d_ptb.p, 2 * n, vs));
            OKB(hipEventRecord(ctx->stage_ev[0], vs) == hipSuccess);
        }
    }
    hipStream_t table_stream = nullptr;
    if (use_table) {
        if (partition) {
            table_stream = ctx->side_stream[1];
        } else {
            if (!ctx->aux_stream) OKB(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) == hipSuccess);
            table_stream = ctx->aux_stream;
        }
    }
    StreamDrain drain_aux{table_stream};
    if (use_table) {
        OKB(hipStreamWaitEvent(table_stream, ctx->stage_ev[0], 0) == hipSuccess);   // the validated points
        RC(dev::call_table_enqueue(table_stream, &tbl, d_tbl.p, d_tbl_tmp.p, d_pts.p));
        if (!ctx->table_ev) OKB(hipEventCreateWithFlags(&ctx->table_ev, hipEventDisableTiming) == hipSuccess);
        OKB(hipEventRecord(ctx->table_ev, table_stream) == hipSuccess);
    }
    tr.mark("validation (+ call-time table) enqueued");
    std::vector<Fr> z(n), y(n);
    ProofSide ps;
    uint8_t digest[32];
    bool have_digest = false;   // the pipelined form hashes the batch transcript while the batch is in flight
    if (piped) {
        // ---- pipelined host-pointer form ----
        // 256 blobs = 32 MB per chunk: ~0.6 ms of PCIe, 16 chunks at n = 4096 (profiles/r03_verify_pipeline_sweep.txt)
        static const size_t CH = (size_t)(dev::ab_knob("CKZG_HIP_VERIFY_CHUNK", 256) < 16 ? 16 : dev::ab_knob("CKZG_HIP_VERIFY_CHUNK", 256));
        const size_t nch = (n + CH - 1) / CH;
        const bool src_pinned = host_pointer_is_pinned(blobs);
        if (!src_pinned) OKM(ensure_pinned(ctx->h_stage, ctx->h_stage_bytes, CH * (size_t)BYTES_PER_BLOB));
        for (int i = 2; i < 4; i++) {
            if (!ctx->stage_ev[i]) OKB(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming) == hipSuccess);
        }
        hipEvent_t *copied = ctx->stage_ev + 2;
        // what the host needs back -- evaluations per chunk, the flags of the blobs and of the points at the end --
        // lands in page-locked memory behind the kernels that produce it, never through a blocking copy
        OKM(ensure_pinned(ctx->h_out, ctx->h_out_bytes, n * sizeof(Fr)));
        const Fr *h_y = static_cast<const Fr *>(ctx->h_out[0]);
        uint8_t *h_y_bytes = static_cast<uint8_t *>(ctx->h_out[0]);
        uint32_t *h_bad = static_cast<uint32_t *>(ctx->h_out[1]);
        uint8_t *h_st = static_cast<uint8_t *>(ctx->h_out[1]) + n * 4;
        // one event per chunk for "copied" and one for "evaluations landed": kept in the slot, not created per call
        while (ctx->chunk_ev.size() < 2 * nch) {
            hipEvent_t e;
            OKB(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess);
            ctx->chunk_ev.push_back(e);
        }
        hipEvent_t *const chunk_copied = ctx->chunk_ev.data(), *const landed = ctx->chunk_ev.data() + nch;
        OKB(hipMemsetAsync(d_bad.p, 0, n * 4, ctx->stream) == hipSuccess);
        // Page-locked source: every chunk's DMA is enqueued up front, each with its own event, so that the copy engine
        // runs at link speed from the first microsecond instead of at the pace this loop is allowed to advance by the
        // hashers (n = 4096: GPU idle again 1.5 ms earlier).  Pageable source: the staging copy IS the pace.
        if (src_pinned) {
            for (size_t c = 0; c < nch; c++) {
                const size_t off = c * CH, k = n - off < CH ? n - off : CH;
                OKB(hipMemcpyAsync(d_blobs_own.p + off * BYTES_PER_BLOB, blobs + off, k * BYTES_PER_BLOB, hipMemcpyHostToDevice,
                                   ctx->copy_stream) == hipSuccess);
                OKB(hipEventRecord(chunk_copied[c], ctx->copy_stream) == hipSuccess);
            }
        }
        // (started after the DMAs of a page-locked source are on their way: waking the workers is host time the copy
        // engine need not wait for)
        OrderedHasher hasher(n, CH);
        hasher.start(z.data(), blobs, cb);   // its destructor waits for the workers on every exit path
        TranscriptHasher transcript;         // likewise
        transcript.start(ctx->device, n, CH, landed, cb, pb, z.data(), h_y);
        bool used[2] = {false, false};
        for (size_t c = 0; c < nch; c++) {
            const size_t off = c * CH, k = n - off < CH ? n - off : CH;
            const int b = (int)(c & 1);
            if (src_pinned) {
                OKB(hipStreamWaitEvent(ctx->stream, chunk_copied[c], 0) == hipSuccess);
            } else {
                if (used[b]) OKB(dev::sync_event(copied[b]) == hipSuccess);   // the DMA out of this staging buffer is done
                staged_copy(ctx->h_stage[b], blobs + off, k * BYTES_PER_BLOB);
                OKB(hipMemcpyAsync(d_blobs_own.p + off * BYTES_PER_BLOB, ctx->h_stage[b], k * BYTES_PER_BLOB, hipMemcpyHostToDevice,
                                   ctx->copy_stream) == hipSuccess);
                OKB(hipEventRecord(copied[b], ctx->copy_stream) == hipSuccess);
                used[b] = true;
                OKB(hipStreamWaitEvent(ctx->stream, copied[b], 0) == hipSuccess);
            }
            // the evaluation of chunk c - 1 is enqueued once its challenges exist: one chunk of slack, so that this
            // thread never waits for the hashers while there is a copy to issue
            if (c >= 1) {
                const size_t po = (c - 1) * CH;
                OKB(hasher.wait_chunk(c - 1));
                OKB(hipMemcpyAsync(d_z.p + po, z.data() + po, CH * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
                RC(dev::eval_blob_bytes_batch_device(ctx, d_y.p + po, d_bad.p + po, d_blobs_own.p + po * BYTES_PER_BLOB, d_z.p + po, CH));
                OKB(hipMemcpyAsync(h_y_bytes + po * sizeof(Fr), d_y.p + po, CH * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
                OKB(hipEventRecord(landed[c - 1], ctx->stream) == hipSuccess);
                transcript.publish(c);
            }
        }
        {
            const size_t po = (nch - 1) * CH, k = n - po;
            OKB(hasher.wait_chunk(nch - 1));
            OKB(hipMemcpyAsync(d_z.p + po, z.data() + po, k * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
            RC(dev::eval_blob_bytes_batch_device(ctx, d_y.p + po, d_bad.p + po, d_blobs_own.p + po * BYTES_PER_BLOB, d_z.p + po, k));
            OKB(hipMemcpyAsync(h_y_bytes + po * sizeof(Fr), d_y.p + po, k * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
            OKB(hipEventRecord(landed[nch - 1], ctx->stream) == hipSuccess);
            transcript.publish(nch);
        }
        OKB(hipMemcpyAsync(h_bad, d_bad.p, n * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
        OKB(hipMemcpyAsync(h_st, d_st.p, 2 * n, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
        tr.mark("chunked H2D + evaluation enqueued (challenges hashed in order on host threads)");
        OKB(dev::sync_stream(ctx->stream) == hipSuccess);
        tr.mark("wait for GPU");
        for (size_t i = 0; i < 2 * n; i++) {
            if (h_st[i]) return C_KZG_BADARGS;
        }
        for (size_t i = 0; i < n; i++) {
            if (h_bad[i]) return C_KZG_BADARGS;
        }
        memcpy(y.data(), h_y, n * sizeof(Fr));
        tr.mark("flags checked");
        OKB(transcript.finish(digest));
        have_digest = true;
        tr.mark("transcript thread joined");
    } else {
    // Challenges: on host threads, started BEFORE the blob copy -- a copy from pageable memory blocks
    // this thread for its whole duration (3.5 us per blob), and with the x86 SHA extensions the hashing
    // (2 us per blob on 32 threads) finishes underneath it.  Hosts without the extensions hash large
    // batches on the GPU instead (a lane per blob; ~6 ms whatever the batch size).
    struct Joiner {
        std::thread t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } hasher;
    auto hash_all = [&]() {
        parallel_for(n, [&](size_t i) { z[i] = challenge_from_bytes(blobs[i].bytes, cb[i].bytes); });
    };
    const bool threaded = !gpu_sha && n >= 16;  // a thread costs ~0.2 ms: not for the single-blob call
    if (threaded) hasher.t = std::thread(hash_all);
    if (!resident) OKB(hipMemcpyAsync(d_blobs_own.p, blobs, n * BYTES_PER_BLOB, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
    OKB(hipMemsetAsync(d_bad.p, 0, n * 4, ctx->stream) == hipSuccess);
    if (!small && !resident) OKB(hipStreamWaitEvent(ctx->stream, ctx->stage_ev[0], 0) == hipSuccess);  // d_ptb, d_pts, d_st ready
    if (gpu_sha) {
        if (partition && !resident) {
            // the blobs and the commitments' bytes reach HBM on the main stream: the hash stream starts behind them
            if (!ctx->stage_ev[2]) OKB(hipEventCreateWithFlags(&ctx->stage_ev[2], hipEventDisableTiming) == hipSuccess);
            OKB(hipEventRecord(ctx->stage_ev[2], ctx->stream) == hipSuccess);
            OKB(hipStreamWaitEvent(ctx->sha_stream, ctx->stage_ev[2], 0) == hipSuccess);
        }
        if (!hash_enqueued) RC(enqueue_hash());
    }
    tr.mark("enqueue H2D (+ GPU validation, GPU challenges)");
    if (hasher.t.joinable()) hasher.t.join();
    if (!gpu_sha && !threaded) hash_all();
    tr.mark("host SHA-256 challenges");
    if (resident) {
        // resident inputs: nothing to learn from the host before the evaluation -- enqueue it straight away
        RC(dev::eval_blob_bytes_batch_device(ctx, d_y.p, d_bad.p, d_blob_bytes, d_z.p, n));
        OKB(hipStreamWaitEvent(ctx->stream, ctx->stage_ev[0], 0) == hipSuccess);   // the validated points, their status, d_ptb
        RC(dev::batch_transcript_rows_device(ctx, d_rows.p, d_ptb.p, d_z.p, d_y.p, n));
        OKB(hipEventRecord(ctx->ev[2], ctx->stream) == hipSuccess);
        uint8_t *h1 = static_cast<uint8_t *>(ctx->h_out[1]);
        OKB(hipMemcpyAsync(ctx->h_out[0], d_rows.p, n * 160, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
        OKB(hipMemcpyAsync(h1, d_y.p, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
        OKB(hipMemcpyAsync(h1 + n * sizeof(Fr), d_bad.p, n * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
        OKB(hipMemcpyAsync(h1 + n * (sizeof(Fr) + 4), d_st.p, 2 * n, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
    }
    OKB(dev::sync_stream(ctx->stream) == hipSuccess);
    tr.mark("wait for GPU");
    if (resident) {
        const uint8_t *h1 = static_cast<const uint8_t *>(ctx->h_out[1]);
        const uint8_t *st = h1 + n * (sizeof(Fr) + 4);
        uint32_t any = 0;
        for (size_t i = 0; i < 2 * n; i++) any |= st[i];
        if (any) return C_KZG_BADARGS;
        const uint32_t *bad = reinterpret_cast<const uint32_t *>(h1 + n * sizeof(Fr));
        for (size_t i = 0; i < n; i++) any |= bad[i];
        if (any) return C_KZG_BADARGS;
        memcpy(y.data(), h1, n * sizeof(Fr));
        if (!use_table) OKB(d_z.down(z.data(), n));   // the ladder sums take their scalars from the host
        // r's transcript (eip4844.c:597-680): the header, then the rows as the device left them
        Sha256 h;
        uint8_t head[32];
        memcpy(head, "RCKZGBATCH___V1_", 16);
        be64(head + 16, FIELD_ELEMENTS_PER_BLOB);
        be64(head + 24, n);
        h.update(head, 32);
        h.update(static_cast<const uint8_t *>(ctx->h_out[0]), n * 160);
        h.finish(digest);
        have_digest = true;
    } else {
    if (!small) {
        std::vector<uint8_t> st(2 * n);
        OKB(d_st.down(st.data(), 2 * n));
        for (size_t i = 0; i < 2 * n; i++) {
            if (st[i]) return C_KZG_BADARGS;
        }
    }
    if (gpu_sha) {
        OKB(d_z.down(z.data(), n));
    } else {
        OKB(d_z.up(z.data(), n));
    }
    RC(dev::eval_blob_bytes_batch_device(ctx, d_y.p, d_bad.p, d_blob_bytes, d_z.p, n));
    if (n == 1) ps = verify_proof_side(z[0], hp[0], prepared_of(ctx));   // host work underneath the GPU's evaluation
    OKB(dev::sync_stream(ctx->stream) == hipSuccess);
    // (the evaluation is what reads the field elements: a blob with one >= r is known now, bytes.c:52-70)
    std::vector<uint32_t> bad(n);
    OKB(d_bad.down(bad.data(), n));
    for (size_t i = 0; i < n; i++) {
        if (bad[i]) return C_KZG_BADARGS;
    }
    OKB(d_y.down(y.data(), n));
    }
    }
    tr.mark("GPU evaluation (+ the proof's half of the check on the host)");
    if (n == 1 && !resident) {
        // the single-blob form of the check (eip4844.c:537-595)
        *ok = verify_with_proof_side(hc[0], y[0], ps, prepared_of(ctx));
        tr.mark("[y]G1 + the commitment's Miller loop + final exponentiation");
        return C_KZG_OK;
    }
    // r = H("RCKZGBATCH___V1_" | u64be 4096 | u64be n | (C_i | z_i | y_i | proof_i)*)  (eip4844.c:597-680);
    // valid compressed encodings are canonical, so the input bytes are the re-compressed bytes
    if (!have_digest) {
        Sha256 h;
        uint8_t head[32], zb[64];
        memcpy(head, "RCKZGBATCH___V1_", 16);
        be64(head + 16, FIELD_ELEMENTS_PER_BLOB);
        be64(head + 24, n);
        h.update(head, 32);
        for (size_t i = 0; i < n; i++) {
            h.update(cb[i].bytes, 48);
            fr_to_bytes(zb, z[i]);
            fr_to_bytes(zb + 32, y[i]);
            h.update(zb, 64);
            h.update(pb[i].bytes, 48);
        }
        h.finish(digest);
    }
    Fr r = fr_from_bytes_reduce(digest);
    tr.mark("batch challenge r (one SHA-256 stream over every C, z, y, proof)");
    G1Jac lc[3];  // sum r^i proof_i, sum r^i z_i proof_i, sum r^i C_i
    Fr ysum = Fr::zero();
    if (use_table) {
        // The sums over the call-time table.  Their scalars are made where their digits are needed: one lane per blob
        // raises r to its index (k_rlc_scalars; the challenges z are in d_z since their chunks were evaluated), so only
        // r crosses PCIe, and while the GPU recodes and accumulates the host adds up sum r^i y_i for its side of the check.
        if (resident) OKB(hipEventRecord(ctx->ev[3], ctx->stream) == hipSuccess);
        OKB(hipStreamWaitEvent(ctx->stream, ctx->table_ev, 0) == hipSuccess);   // the table is complete
        RC(dev::rlc_scalars_enqueue(ctx->stream, d_sc.p, d_z.p, r, n));
        RC(dev::table_sums_enqueue(ctx->stream, tbl, d_sums.p, d_sc.p, 3, d_sums_scr.p));
        Fr pw = Fr::one();
        for (size_t i = 0; i < n; i++) {
            ysum = add(ysum, mul(pw, y[i]));
            pw = mul(pw, r);
        }
        tr.mark("powers of r (host: sum r^i y_i; GPU: the scalars of the sums)");
        G1XYZZ hs[3];
        OKB(d_sums.down(hs, 3));
        for (int j = 0; j < 3; j++) lc[j] = jac_from_xyzz(hs[j]);
        if (resident) {
            float a = 0, b = 0;
            OKB(hipEventRecord(ctx->ev[4], ctx->stream) == hipSuccess && dev::sync_event(ctx->ev[4]) == hipSuccess);
            if (hipEventElapsedTime(&a, ctx->ev[1], ctx->ev[2]) == hipSuccess &&
                hipEventElapsedTime(&b, ctx->ev[3], ctx->ev[4]) == hipSuccess) {
                ctx->last_ms[3] = a + b;
                ctx->last_ms[0] = a;
                ctx->last_ms[2] = b;
            }
        }
    } else {
    std::vector<Fr> rpf(n), rzf(n);
    Fr pw = Fr::one();
    for (size_t i = 0; i < n; i++) {
        rpf[i] = pw;
        rzf[i] = mul(pw, z[i]);
        ysum = add(ysum, mul(pw, y[i]));
        pw = mul(pw, r);
    }
    tr.mark("powers of r");
    if (small) {
        lc[0] = host_lincomb(hp, rpf);
        lc[1] = host_lincomb(hp, rzf);
        lc[2] = host_lincomb(hc, rpf);
    } else {
        std::vector<RawScalar> rp(n), rz(n);
        for (size_t i = 0; i < n; i++) {
            rp[i] = raw_of(rpf[i]);
            rz[i] = raw_of(rzf[i]);
        }
        if (resident) OKB(hipEventRecord(ctx->ev[3], ctx->stream) == hipSuccess);
        LincombJob jobs[3] = {{d_pts.p + n, &rp}, {d_pts.p + n, &rz}, {d_pts.p, &rp}};
        C_KZG_RET ret = gpu_lincomb_multi(ctx, lc, jobs, 3);
        if (ret != C_KZG_OK) return ret;
        if (resident) {
            // kernel-only time of the resident form (ckzg_hip_last_kernel_ms, which = 3): validation + conversion +
            // challenges + evaluation, and the three sums; the host transcript between them is not GPU time
            float a = 0, b = 0;
            OKB(hipEventRecord(ctx->ev[4], ctx->stream) == hipSuccess && dev::sync_event(ctx->ev[4]) == hipSuccess);
            if (hipEventElapsedTime(&a, ctx->ev[1], ctx->ev[2]) == hipSuccess &&
                hipEventElapsedTime(&b, ctx->ev[3], ctx->ev[4]) == hipSuccess) {
                ctx->last_ms[3] = a + b;
                ctx->last_ms[0] = a;
                ctx->last_ms[2] = b;
            }
        }
    }
    }
    if (split_validation) {
        OKB(dev::sync_event(ctx->stage_ev[1]) == hipSuccess);
        std::vector<uint8_t> st2(2 * n);
        OKB(d_st2.down(st2.data(), 2 * n));
        for (size_t i = 0; i < 2 * n; i++) {
            if (st2[i]) return C_KZG_BADARGS;  // a point outside G1: the sums above are discarded
        }
    }
    tr.mark("transcript + lincombs");
    // sum r^i (C_i - [y_i]G) = sum r^i C_i - [sum r^i y_i]G
    G1Jac rhs = jac_add(jac_add(lc[2], jac_neg(g1_gen_mul_fr(ysum))), lc[1]);
    // e(sum r^i proof_i, [s]G2) == e(rhs, G2)
    *ok = pairing_product_is_one(jac_to_affine_fast(jac_neg(lc[0])), prepared_of(ctx)->s1, jac_to_affine_fast(rhs),
                                 prepared_of(ctx)->gen);
    tr.mark("pairing check");
    return C_KZG_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// EIP-4844
// ------------------------------------------------------------------------------------------

extern "C" void compute_challenge(fr_t *eval_challenge_out, const Blob *blob, const g1_t *commitment) {
    uint8_t c48[48];
    g1_compress_affine(c48, jac_to_affine_fast(*as_g1(commitment)));
    *as_fr(eval_challenge_out) = challenge_from_bytes(blob->bytes, c48);
}

static C_KZG_RET compute_kzg_proof_on(dev::DeviceCtx *ctx, KZGProof *proof_out, Bytes32 *y_out, const Blob *blob,
                                      const Bytes32 *z_bytes, const KZGSettings *s) {
    // eip4844.c:386-415.  Evaluation, quotient polynomial and MSM on the GPU; a z inside the evaluation
    // domain (eip4844.c:458-481) takes the host form of the quotient instead.
    Fr z, y;
    if (!fr_from_bytes_canonical(z, z_bytes->bytes)) return C_KZG_BADARGS;
    {
        Arena &ar = ctx->api_arena;
        OKM(ar.begin(BYTES_PER_BLOB + 2 * FIELD_ELEMENTS_PER_BLOB * sizeof(Fr) + 2 * sizeof(Fr) + 64));
        ABuf<uint8_t> d_blob(ar, BYTES_PER_BLOB), d_out(ar, 48);
        ABuf<Fr> d_poly(ar, FIELD_ELEMENTS_PER_BLOB), d_z(ar, 1), d_y(ar, 1);
        ABuf<uint32_t> d_bad(ar, 1), d_q(ar, FIELD_ELEMENTS_PER_BLOB * 8);
        ABuf<int> d_hit(ar, 1);
        OKM(d_blob.p && d_out.p && d_poly.p && d_z.p && d_y.p && d_bad.p && d_q.p && d_hit.p);
        OKB(hipMemcpyAsync(d_blob.p, blob->bytes, BYTES_PER_BLOB, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
        OKB(hipMemcpyAsync(d_z.p, &z, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
        OKB(hipMemsetAsync(d_bad.p, 0, 4, ctx->stream) == hipSuccess);
        RC(dev::bytes_to_fr_batch(ctx, d_poly.p, d_bad.p, d_blob.p, FIELD_ELEMENTS_PER_BLOB, FIELD_ELEMENTS_PER_BLOB));
        RC(dev::eval_quotient_batch_device(ctx, d_y.p, d_q.p, d_hit.p, d_poly.p, d_z.p, 1));
        RC(dev::msm_commit_table_raw_device(ctx, d_out.p, d_q.p, 1));
        uint32_t bad = 0;
        int hit = -1;
        OKB(d_bad.down(&bad, 1) && d_hit.down(&hit, 1));
        if (bad) return C_KZG_BADARGS;  // a non-canonical field element in the blob (blob.c:31-38)
        if (hit < 0) {
            OKB(d_y.down(&y, 1) && d_out.down(proof_out->bytes, 48));
            fr_to_bytes(y_out->bytes, y);
            return C_KZG_OK;
        }
    }
    std::vector<Fr> poly(FIELD_ELEMENTS_PER_BLOB);
    C_KZG_RET ret = blob_to_polynomial(poly.data(), blob);
    if (ret != C_KZG_OK) return ret;
    ret = compute_kzg_proof_impl(proof_out, y, poly.data(), z, s, ctx);
    if (ret != C_KZG_OK) return ret;
    fr_to_bytes(y_out->bytes, y);
    return C_KZG_OK;
}

extern "C" C_KZG_RET compute_kzg_proof(KZGProof *proof_out, Bytes32 *y_out, const Blob *blob,
                                       const Bytes32 *z_bytes, const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        Lease lease(s);
        if (!lease.ctx) return C_KZG_ERROR;
        return compute_kzg_proof_on(lease.ctx, proof_out, y_out, blob, z_bytes, s);
    });
}

// compute_blob_kzg_proof with the evaluation and the quotient polynomial on the host (only the MSM on
// the GPU): the general form, which also covers a challenge that falls inside the evaluation domain
// (eip4844.c:458-481).  The batch entry point sends such blobs here.
static C_KZG_RET blob_proof_host_quotient(dev::DeviceCtx *ctx, KZGProof *out, const Blob *blob,
                                          const Bytes48 *commitment_bytes, const KZGSettings *s) {
    std::vector<Fr> poly(FIELD_ELEMENTS_PER_BLOB);
    G1Jac c;
    Fr y;
    C_KZG_RET ret = validate_kzg_g1(c, commitment_bytes->bytes);
    if (ret != C_KZG_OK) return ret;
    ret = blob_to_polynomial(poly.data(), blob);
    if (ret != C_KZG_OK) return ret;
    Fr z = challenge_from_bytes(blob->bytes, commitment_bytes->bytes);
    return compute_kzg_proof_impl(out, y, poly.data(), z, s, ctx);
}

// compute_blob_kzg_proof for a batch: challenges on the host (SHA-256), evaluation + quotient
// polynomial and the 4096-term MSMs on the GPU.
static C_KZG_RET blob_proof_batch_on(dev::DeviceCtx *ctx, KZGProof *proofs, uint8_t *status, const Blob *blobs,
                                     const Bytes48 *commitments_bytes, uint64_t n, const KZGSettings *s) {
    if (n == 0) return C_KZG_OK;
    C_KZG_RET ret = C_KZG_OK;
    std::vector<uint8_t> st(n, 0);
    std::vector<int> redo;
    {
        const uint64_t CH = 256;
        const uint64_t m = n < CH ? n : CH;
        Arena &ar = ctx->api_arena;
        OKM(ar.begin(m * (BYTES_PER_BLOB + 48 + 1 + sizeof(G1Affine) + 48 + 2 * FIELD_ELEMENTS_PER_BLOB * sizeof(Fr) +
                          2 * sizeof(Fr) + 8)));
        ArenaTrim trim(ar);
        ABuf<uint8_t> d_blobs(ar, m * BYTES_PER_BLOB), d_ptb(ar, m * 48), d_pst(ar, m), d_out(ar, m * 48);
        ABuf<G1Affine> d_pts(ar, m);
        ABuf<Fr> d_poly(ar, m * FIELD_ELEMENTS_PER_BLOB), d_z(ar, m), d_y(ar, m);
        ABuf<uint32_t> d_bad(ar, m), d_q(ar, m * FIELD_ELEMENTS_PER_BLOB * 8);
        ABuf<int> d_hit(ar, m);
        OKM(d_blobs.p && d_ptb.p && d_pst.p && d_out.p && d_pts.p && d_poly.p && d_z.p && d_y.p && d_bad.p && d_q.p &&
            d_hit.p);
        struct StreamDrain {  // the second stream must be idle before the arena is reused, on every exit path
            hipStream_t s;
            ~StreamDrain() { (void)dev::sync_stream(s); }
        } drain{ctx->copy_stream};
        std::vector<Fr> z(m);
        std::vector<uint8_t> pst(m);
        std::vector<uint32_t> bad(m);
        std::vector<int> hit(m);
        for (uint64_t off = 0; off < n; off += CH) {
            const uint64_t k = n - off < CH ? n - off : CH;
            // commitments must be valid G1 points (bytes_to_kzg_commitment, eip4844.c:513)
            if (k > SMALL_VERIFY_N) {
                // only a verdict is needed: the whole validation runs on the second stream, underneath
                // the copy, evaluation and MSM of this chunk
                OKB(hipMemcpyAsync(d_ptb.p, commitments_bytes + off, k * 48, hipMemcpyHostToDevice, ctx->copy_stream) == hipSuccess);
                RC(dev::validate_g1_batch_device(ctx, d_pts.p, d_pst.p, d_ptb.p, k, ctx->copy_stream));
            } else {
                for (uint64_t i = 0; i < k; i++) {
                    G1Jac c;
                    pst[i] = validate_kzg_g1(c, commitments_bytes[off + i].bytes) == C_KZG_OK ? 0 : 1;
                }
            }
            {
                // the challenges are hashed by host threads underneath the (thread-blocking) blob copy
                struct Joiner {
                    std::thread t;
                    ~Joiner() {
                        if (t.joinable()) t.join();
                    }
                } hasher;
                auto hash_all = [&]() {
                    parallel_for(k, [&](size_t i) {
                        z[i] = challenge_from_bytes(blobs[off + i].bytes, commitments_bytes[off + i].bytes);
                    });
                };
                const bool threaded = k >= 16;  // a thread costs ~0.2 ms: not for the single-blob call
                if (threaded) hasher.t = std::thread(hash_all);
                OKB(hipMemcpyAsync(d_blobs.p, blobs + off, k * BYTES_PER_BLOB, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
                OKB(hipMemsetAsync(d_bad.p, 0, k * 4, ctx->stream) == hipSuccess);
                RC(dev::bytes_to_fr_batch(ctx, d_poly.p, d_bad.p, d_blobs.p, k * FIELD_ELEMENTS_PER_BLOB, FIELD_ELEMENTS_PER_BLOB));
                if (!threaded) hash_all();
            }
            OKB(hipMemcpyAsync(d_z.p, z.data(), k * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
            RC(dev::eval_quotient_batch_device(ctx, d_y.p, d_q.p, d_hit.p, d_poly.p, d_z.p, k));
            RC(dev::msm_commit_table_raw_device(ctx, d_out.p, d_q.p, k));
            if (k > SMALL_VERIFY_N) {
                OKB(dev::sync_stream(ctx->copy_stream) == hipSuccess);
                OKB(d_pst.down(pst.data(), k));
            }
            OKB(d_bad.down(bad.data(), k) && d_hit.down(hit.data(), k));
            OKB(hipMemcpy(proofs + off, d_out.p, k * 48, hipMemcpyDeviceToHost) == hipSuccess);
            for (uint64_t i = 0; i < k; i++) {
                if (pst[i] || bad[i]) {
                    st[off + i] = C_KZG_BADARGS;
                    ret = C_KZG_BADARGS;
                } else if (hit[i] >= 0) {
                    redo.push_back((int)(off + i));  // challenge inside the domain: scalar path below
                }
            }
        }
    }
    for (int i : redo) {
        C_KZG_RET r = blob_proof_host_quotient(ctx, &proofs[i], &blobs[i], &commitments_bytes[i], s);
        if (r != C_KZG_OK) {
            st[i] = (uint8_t)r;
            ret = r;
        }
    }
    if (status) memcpy(status, st.data(), n);
    return ret;
}

extern "C" C_KZG_RET ckzg_hip_compute_blob_kzg_proof_batch(KZGProof *proofs, uint8_t *status, const Blob *blobs,
                                                           const Bytes48 *commitments_bytes, uint64_t n,
                                                           const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        if (!settings_of(s)) return C_KZG_ERROR;
        if (n == 0) return C_KZG_OK;
        return for_each_device_shard(s, n, 64, [&](dev::DeviceCtx *ctx, uint64_t lo, uint64_t hi) {
            return blob_proof_batch_on(ctx, proofs + lo, status ? status + lo : nullptr, blobs + lo,
                                       commitments_bytes + lo, hi - lo, s);
        });
    });
}

extern "C" C_KZG_RET compute_blob_kzg_proof(KZGProof *out, const Blob *blob, const Bytes48 *commitment_bytes,
                                            const KZGSettings *s) {
    // eip4844.c:496-535; evaluation, quotient and MSM on the GPU: a batch of one for a lone caller, one batch
    // launch for callers that arrive while others are in flight (combiner.hpp)
    return guarded([&]() -> C_KZG_RET {
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        auto solo = [&]() -> C_KZG_RET {
            uint8_t st = 0;
            return ckzg_hip_compute_blob_kzg_proof_batch(out, &st, blob, commitment_bytes, 1, s);
        };
        Combiner *cb = sc->comb[CB_BLOB_PROOF];
        if (!cb) return solo();
        const size_t UNITS = 256;   // device_ctx.hip: create_settings_ctx
        return cb->submit(
            nullptr, 0, solo,
            [&](uint8_t *h_in, size_t idx) {
                memcpy(h_in + idx * BYTES_PER_BLOB, blob, BYTES_PER_BLOB);
                memcpy(h_in + UNITS * BYTES_PER_BLOB + idx * 48, commitment_bytes, 48);
            },
            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *st, size_t n) -> C_KZG_RET {
                Lease lease(s);
                if (!lease.ctx) return C_KZG_ERROR;
                C_KZG_RET r = blob_proof_batch_on(lease.ctx, reinterpret_cast<KZGProof *>(h_out), st, reinterpret_cast<const Blob *>(h_in),
                                                  reinterpret_cast<const Bytes48 *>(h_in + UNITS * BYTES_PER_BLOB), n, s);
                // blob_proof_batch_on writes flags only on its last path, where a non-OK return is the code of a flagged
                // unit (BADARGS, or ERROR / MALLOC from one unit's host quotient): the verdict is per unit, and the
                // combiner demotes exactly BADARGS-with-flags to that -- an unflagged member must not inherit it
                if (r != C_KZG_OK)
                    for (size_t i = 0; i < n; i++)
                        if (st[i]) return C_KZG_BADARGS;
                return r;
            },
            [&](const uint8_t *h_out, size_t idx, size_t) { memcpy(out, h_out + idx * 48, 48); });
    });
}

extern "C" C_KZG_RET verify_kzg_proof(bool *ok, const Bytes48 *commitment_bytes, const Bytes32 *z_bytes,
                                      const Bytes32 *y_bytes, const Bytes48 *proof_bytes,
                                      const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        G1Jac c, p;
        Fr z, y;
        *ok = false;
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        if (validate_kzg_g1(c, commitment_bytes->bytes) != C_KZG_OK) return C_KZG_BADARGS;
        if (!fr_from_bytes_canonical(z, z_bytes->bytes)) return C_KZG_BADARGS;
        if (!fr_from_bytes_canonical(y, y_bytes->bytes)) return C_KZG_BADARGS;
        if (validate_kzg_g1(p, proof_bytes->bytes) != C_KZG_OK) return C_KZG_BADARGS;
        *ok = verify_kzg_proof_impl(c, z, y, p, &sc->prepared);
        return C_KZG_OK;
    });
}

// eip4844.c:537-595.  Threads that verify single blobs concurrently on one KZGSettings (what a consensus client does
// with the blobs of a block) are served by ONE verify_blob_kzg_proof_batch launch over all of them (combiner.hpp): if
// that batch comes out true every member is valid (the batch equation's soundness error is 2^-255); if it does not --
// a wrong proof, a malformed point, a non-canonical field element somewhere in it -- nothing is known about any single
// member and each one runs its own verification afterwards (RETRY_SOLO), so an invalid blob costs its batch one wasted
// launch and never changes another caller's answer.  A lone caller takes the single-blob path at once, as before.
extern "C" C_KZG_RET verify_blob_kzg_proof(bool *ok, const Blob *blob, const Bytes48 *commitment_bytes,
                                           const Bytes48 *proof_bytes, const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        *ok = false;
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        auto solo = [&]() -> C_KZG_RET {
            *ok = false;
            Lease lease(s);
            if (!lease.ctx) return C_KZG_ERROR;
            return verify_blobs_core(ok, blob, commitment_bytes, proof_bytes, 1, s, lease.ctx);
        };
        Combiner *cb = sc->comb[CB_VERIFY_BLOB];
        if (!cb) return solo();
        const size_t UNITS = 128;   // device_ctx.hip: create_settings_ctx
        return cb->submit(
            nullptr, 0, solo,
            [&](uint8_t *h_in, size_t idx) {
                memcpy(h_in + idx * BYTES_PER_BLOB, blob, BYTES_PER_BLOB);
                memcpy(h_in + UNITS * BYTES_PER_BLOB + idx * 48, commitment_bytes, 48);
                memcpy(h_in + UNITS * (BYTES_PER_BLOB + 48) + idx * 48, proof_bytes, 48);
            },
            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *st, size_t n) -> C_KZG_RET {
                bool all = false;
                C_KZG_RET r;
                {
                    Lease lease(s);
                    if (!lease.ctx) return C_KZG_ERROR;
                    r = verify_blobs_core(&all, reinterpret_cast<const Blob *>(h_in),
                                          reinterpret_cast<const Bytes48 *>(h_in + UNITS * BYTES_PER_BLOB),
                                          reinterpret_cast<const Bytes48 *>(h_in + UNITS * (BYTES_PER_BLOB + 48)), n, s, lease.ctx);
                }
                if (r != C_KZG_OK && r != C_KZG_BADARGS) return r;   // the launch itself failed: every member hears it
                const bool good = r == C_KZG_OK && all;
                for (size_t i = 0; i < n; i++) {
                    h_out[i] = good ? 1 : 0;
                    st[i] = good ? 0 : Combiner::RETRY_SOLO;
                }
                return C_KZG_OK;
            },
            [&](const uint8_t *h_out, size_t idx, size_t) { *ok = h_out[idx] != 0; });
    });
}

extern "C" C_KZG_RET verify_blob_kzg_proof_batch(bool *ok, const Blob *blobs, const Bytes48 *commitments_bytes,
                                                 const Bytes48 *proofs_bytes, uint64_t n,
                                                 const KZGSettings *s) {
    if (n == 0) {  // eip4844.c:791-794
        *ok = true;
        return C_KZG_OK;
    }
    if (n == 1) return verify_blob_kzg_proof(ok, blobs, commitments_bytes, proofs_bytes, s);
    return guarded([&]() -> C_KZG_RET {
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        // With several devices each one verifies a contiguous shard with its own random linear combination
        // and pairing check and the verdicts are AND-ed (SURVEY section 8e sketches one global challenge with
        // gathered partial sums; independent shards give the same verdict -- soundness error 2^-255 per shard --
        // with no exchange step, at the price of one ~0.8 ms host pairing per device instead of one per call).
        std::atomic<int> all_ok(1);
        C_KZG_RET ret = for_each_device_shard(s, n, 256, [&](dev::DeviceCtx *ctx, uint64_t lo, uint64_t hi) {
            bool res = false;
            C_KZG_RET r = verify_blobs_core(&res, blobs + lo, commitments_bytes + lo, proofs_bytes + lo, hi - lo, s, ctx);
            if (r == C_KZG_OK && !res) all_ok.store(0);
            return r;
        });
        if (ret == C_KZG_OK) *ok = all_ok.load() != 0;
        return ret;
    });
}

// verify_blob_kzg_proof_batch with blobs, commitments and proofs resident in HBM (device pointers on one GPU):
// nothing but 96 + 64 bytes per blob (commitment, proof, challenge, evaluation -- the Fiat-Shamir transcript of
// eip4844.c:597-680 is hashed on the host) leaves the device.  The verdict is written to the HOST bool *ok.
extern "C" C_KZG_RET ckzg_hip_verify_blob_kzg_proof_batch_device(bool *ok, const void *d_blobs, const void *d_commitments,
                                                                 const void *d_proofs, uint64_t n, const KZGSettings *s) {
    if (!ok) return C_KZG_BADARGS;
    if (n == 0) {
        *ok = true;
        return C_KZG_OK;
    }
    return guarded([&]() -> C_KZG_RET {
        *ok = false;
        SettingsCtx *sc = settings_of(s);
        if (!sc) return C_KZG_ERROR;
        const void *ptrs[3] = {d_blobs, d_commitments, d_proofs};
        const int pool = pool_of_pointers(sc, ptrs, 3);
        if (pool < 0 || !d_blobs || !d_commitments || !d_proofs) return C_KZG_BADARGS;
        Lease lease(s, pool);
        if (!lease.ctx) return C_KZG_ERROR;
        return verify_blobs_core(ok, static_cast<const Blob *>(d_blobs), static_cast<const Bytes48 *>(d_commitments),
                                 static_cast<const Bytes48 *>(d_proofs), n, s, lease.ctx, /*resident=*/true);
    });
}

// ------------------------------------------------------------------------------------------
// EIP-7594: recovery
// ------------------------------------------------------------------------------------------

// (x - r_0)...(x - r_{n-1}), coefficients low to high  (recovery.c:46-75)
static void vanishing_poly_from_roots(std::vector<Fr> &poly, const std::vector<Fr> &roots) {
    size_t n = roots.size();
    poly.assign(n + 1, Fr::zero());
    poly[0] = neg(roots[0]);
    for (size_t i = 1; i < n; i++) {
        Fr nr = neg(roots[i]);
        poly[i] = add(nr, poly[i - 1]);
        for (size_t j = i - 1; j > 0; j--) poly[j] = add(mul(poly[j], nr), poly[j - 1]);
        poly[0] = mul(poly[0], nr);
    }
    poly[n] = Fr::one();
}

// recover_cells (recovery.c:200-365) on the GPU for `count` extended blobs that miss the SAME
// cells.  d_e holds count x 8192 values in cell (bit-reversed) order with zeros at the missing
// cells and is overwritten with the recovered values, same order.  Everything that depends only
// on the missing set (Z over the domain, 1/Z over the coset) is computed once.
static C_KZG_RET recover_cells_gpu(dev::DeviceCtx *ctx, Fr *d_e, size_t count, const uint64_t *cell_indices,
                                   size_t num_cells, const KZGSettings *s) {
    const size_t n = FIELD_ELEMENTS_PER_EXT_BLOB;
    std::vector<Fr> roots;
    const Fr *rou = as_fr(s->roots_of_unity);
    bool have[CELLS_PER_EXT_BLOB] = {false};
    for (size_t k = 0; k < num_cells; k++) have[cell_indices[k]] = true;
    for (size_t i = 0; i < CELLS_PER_EXT_BLOB; i++)
        if (!have[i]) roots.push_back(rou[reverse_bits_limited(CELLS_PER_EXT_BLOB, i) * (n / CELLS_PER_EXT_BLOB)]);
    if (roots.empty() || roots.size() >= CELLS_PER_EXT_BLOB) return C_KZG_BADARGS;  // recovery.c:103-106
    std::vector<Fr> shortp, zc(n, Fr::zero()), ones(n, Fr::one());
    vanishing_poly_from_roots(shortp, roots);
    for (size_t i = 0; i < shortp.size(); i++) zc[i * FIELD_ELEMENTS_PER_CELL] = shortp[i];
    // (the caller's arena scope is open: these three vectors come out of it too)
    ABuf<Fr> d_zc(ctx->api_arena, n), d_zev(ctx->api_arena, n), d_zinv(ctx->api_arena, n);
    OKM(d_zc.p && d_zev.p && d_zinv.p);
    OKB(d_zc.up(zc.data(), n));
    OKB(d_zinv.up(ones.data(), n));
    OKB(hipMemcpyAsync(d_zev.p, d_zc.p, n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess);
    // Z over the domain, in bit-reversed order = the order of d_e
    RC(dev::fr_ntt_batch(ctx, d_zev.p, 1, 13, true, false, false));
    // 1 / Z over the coset (the divisor of recovery.c:322-328, inverted once for the whole batch)
    RC(dev::fr_mul_inplace_device(ctx, d_zc.p, ctx->d_shift, n, n));
    RC(dev::fr_ntt_batch(ctx, d_zc.p, 1, 13, true, false, false));
    RC(dev::fr_div_inplace_device(ctx, d_zinv.p, d_zc.p, n));
    const size_t tot = count * n;
    RC(dev::fr_mul_inplace_device(ctx, d_e, d_zev.p, tot, n));           // (E * Z)(w^i)
    RC(dev::fr_ntt_batch(ctx, d_e, count, 13, false, true, true));       // -> coefficients
    RC(dev::fr_mul_inplace_device(ctx, d_e, ctx->d_shift, tot, n));      // coset_fft: scale by 7^i ...
    RC(dev::fr_ntt_batch(ctx, d_e, count, 13, true, false, false));      // ... and transform
    RC(dev::fr_mul_inplace_device(ctx, d_e, d_zinv.p, tot, n));          // recovery.c:322-328
    RC(dev::fr_ntt_batch(ctx, d_e, count, 13, false, true, true));       // coset_ifft ...
    RC(dev::fr_mul_inplace_device(ctx, d_e, ctx->d_unshift, tot, n));    // ... unscale by 7^-i
    RC(dev::fr_ntt_batch(ctx, d_e, count, 13, true, false, false));      // evaluations, cell order
    OKB(dev::sync_stream(ctx->stream) == hipSuccess);
    return C_KZG_OK;
}

static C_KZG_RET recover_batch_on(dev::DeviceCtx *ctx, Cell *recovered_cells, KZGProof *recovered_proofs,
                                  uint8_t *status, const uint64_t *cell_indices, const Cell *cells,
                                  uint64_t num_cells, uint64_t num_blobs, const KZGSettings *s) {
    if (num_blobs == 0) return C_KZG_OK;
    const size_t n = FIELD_ELEMENTS_PER_EXT_BLOB;
    const size_t CH = 512;  // 512 rows: 128 MiB of byte image + 128 MiB of Fr + 64 MiB of coefficients
    const size_t m = num_blobs < CH ? (size_t)num_blobs : CH;
    // Batches drain their outputs (256 KB of cells + 6 KB of proofs per row) through an OutPipe: the cells
    // of a chunk cross PCIe while its proofs are computed, the proofs while the next chunk starts.  Output
    // buffers alternate between chunks.  A call with a few rows copies directly (no helper thread).
    const bool piped = num_blobs > 8;
    const int nbuf = piped ? 2 : 1;
    std::vector<uint32_t> idx32(num_cells);
    for (size_t i = 0; i < num_cells; i++) idx32[i] = (uint32_t)cell_indices[i];
    std::vector<uint32_t> bad(m);
    C_KZG_RET result = C_KZG_OK;
    Arena &ar = ctx->api_arena;
    const size_t nchunks = (size_t)((num_blobs + CH - 1) / CH);  // recover_cells_gpu takes 3 vectors per chunk
    OKM(ar.begin(m * (nbuf * n * 32 + num_cells * BYTES_PER_CELL + n * sizeof(Fr) + 4) + num_cells * 4 +
                 nchunks * 3 * (n * sizeof(Fr) + 256) +
                 (recovered_proofs ? m * (nbuf * CELLS_PER_EXT_BLOB * 48 + FIELD_ELEMENTS_PER_BLOB * sizeof(Fr)) : 0) + 4096));
    ArenaTrim trim(ar);
    ABuf<uint8_t> d_img0(ar, m * n * 32), d_img1(ar, piped ? m * n * 32 : 1), d_in(ar, m * num_cells * BYTES_PER_CELL);
    ABuf<uint8_t> d_pr0(ar, recovered_proofs ? m * CELLS_PER_EXT_BLOB * 48 : 1);
    ABuf<uint8_t> d_pr1(ar, recovered_proofs && piped ? m * CELLS_PER_EXT_BLOB * 48 : 1);
    ABuf<Fr> d_e(ar, m * n), d_poly(ar, recovered_proofs ? m * FIELD_ELEMENTS_PER_BLOB : 1);
    ABuf<uint32_t> d_bad(ar, m), d_idx(ar, num_cells);
    OKM(d_img0.p && d_img1.p && d_in.p && d_pr0.p && d_pr1.p && d_e.p && d_poly.p && d_bad.p && d_idx.p);
    uint8_t *img_buf[2] = {d_img0.p, piped ? d_img1.p : d_img0.p}, *pr_buf[2] = {d_pr0.p, piped ? d_pr1.p : d_pr0.p};
    OKB(d_idx.up(idx32.data(), num_cells));
    OutPipe pipe(ctx);
    struct Drain {  // nothing may still read the arena when this function leaves, on any path
        dev::DeviceCtx *c;
        OutPipe &p;
        ~Drain() {
            (void)p.finish();
            (void)dev::sync_stream(c->stream);
        }
    } drain{ctx, pipe};
    std::vector<size_t> mark;
    size_t chunk = 0;
    OKB(hipEventRecord(ctx->ev[1], ctx->stream) == hipSuccess);
    for (size_t off = 0; off < num_blobs; off += CH, chunk++) {
        const size_t k = num_blobs - off < CH ? (size_t)(num_blobs - off) : CH;
        const size_t in_bytes = k * num_cells * BYTES_PER_CELL;
        uint8_t *d_img = img_buf[chunk & 1], *d_proofs = pr_buf[chunk & 1];
        if (piped && chunk >= 2) pipe.wait_for(mark[chunk - 2]);
        OKB(hipMemcpyAsync(d_in.p, cells + off * num_cells, in_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
        OKB(hipMemsetAsync(d_img, 0, k * n * 32, ctx->stream) == hipSuccess);
        OKB(hipMemsetAsync(d_bad.p, 0, k * 4, ctx->stream) == hipSuccess);
        RC(dev::scatter_cells_device(ctx, d_img, d_in.p, d_idx.p, (uint32_t)num_cells, k));
        RC(dev::bytes_to_fr_batch(ctx, d_e.p, d_bad.p, d_img, k * n, (uint32_t)n));
        OKB(dev::sync_stream(ctx->stream) == hipSuccess);
        OKB(d_bad.down(bad.data(), k));
        bool any_bad = false;
        for (size_t i = 0; i < k; i++) {
            if (status) status[off + i] = bad[i] ? (uint8_t)C_KZG_BADARGS : 0;
            any_bad |= bad[i] != 0;
        }
        if (any_bad) result = C_KZG_BADARGS;  // rows flagged in status[] hold unspecified output
        if (num_cells != CELLS_PER_EXT_BLOB) {
            C_KZG_RET ret = recover_cells_gpu(ctx, d_e.p, k, cell_indices, num_cells, s);
            if (ret != C_KZG_OK) return ret;
            if (recovered_cells) RC(dev::fr_to_bytes_batch(ctx, d_img, d_e.p, k * n));
        }
        if (recovered_cells) {
            if (piped) {
                OKB(pipe.push(d_img, recovered_cells + off * CELLS_PER_EXT_BLOB, k * n * 32));
            } else {
                OKB(dev::sync_stream(ctx->stream) == hipSuccess);
                OKB(hipMemcpy(recovered_cells + off * CELLS_PER_EXT_BLOB, d_img, k * n * 32, hipMemcpyDeviceToHost) == hipSuccess);
            }
        }
        if (recovered_proofs) {
            // cell order is bit-reversed evaluation order: DIT inverse gives the coefficients
            // (poly_lagrange_to_monomial over 8192 points, eip7594.c:270); FK20 reads the low 4096
            RC(dev::fr_ntt_batch(ctx, d_e.p, k, 13, false, true, true));
            OKB(hipMemcpy2DAsync(d_poly.p, FIELD_ELEMENTS_PER_BLOB * sizeof(Fr), d_e.p, n * sizeof(Fr),
                                 FIELD_ELEMENTS_PER_BLOB * sizeof(Fr), k, hipMemcpyDeviceToDevice,
                                 ctx->stream) == hipSuccess);
            RC(dev::fk20_proofs_device(ctx, d_proofs, d_poly.p, k));
            if (piped) {
                OKB(pipe.push(d_proofs, recovered_proofs + off * CELLS_PER_EXT_BLOB, k * CELLS_PER_EXT_BLOB * 48));
            } else {
                OKB(hipMemcpy(recovered_proofs + off * CELLS_PER_EXT_BLOB, d_proofs, k * CELLS_PER_EXT_BLOB * 48,
                              hipMemcpyDeviceToHost) == hipSuccess);
            }
        }
        mark.push_back(pipe.pushed_count());
    }
    OKB(hipEventRecord(ctx->ev[4], ctx->stream) == hipSuccess);
    if (pipe.finish() != C_KZG_OK) return C_KZG_ERROR;
    OKB(dev::sync_stream(ctx->stream) == hipSuccess);
    {   // ckzg_hip_last_kernel_ms: 3 = the device section of the call, 1 / 4 = k_msm_small / G1 FFTs of the last chunk
        float ms;
        if (hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[4]) == hipSuccess) ctx->last_ms[3] = ms;
        if (recovered_proofs) dev::fk20_collect_times(ctx);
        (void)hipGetLastError();
    }
    return result;
}

extern "C" C_KZG_RET ckzg_hip_recover_cells_and_kzg_proofs_batch(Cell *recovered_cells, KZGProof *recovered_proofs,
                                                                 uint8_t *status, const uint64_t *cell_indices,
                                                                 const Cell *cells, uint64_t num_cells,
                                                                 uint64_t num_blobs, const KZGSettings *s) {
    // eip7594.c:177-304, for num_blobs rows that all hold the same num_cells columns
    return guarded([&]() -> C_KZG_RET {
        if (recovered_cells == NULL && recovered_proofs == NULL) return C_KZG_BADARGS;
        if (num_cells > CELLS_PER_EXT_BLOB || num_cells < CELLS_PER_BLOB) return C_KZG_BADARGS;
        if (cell_indices == NULL || cells == NULL) return C_KZG_BADARGS;   // (the reference dereferences both)
        for (size_t i = 0; i < num_cells; i++) {
            if (cell_indices[i] >= CELLS_PER_EXT_BLOB) return C_KZG_BADARGS;
            if (i > 0 && cell_indices[i] <= cell_indices[i - 1]) return C_KZG_BADARGS;
        }
        if (!settings_of(s)) return C_KZG_ERROR;
        if (num_blobs == 0) return C_KZG_OK;
        return for_each_device_shard(s, num_blobs, 16, [&](dev::DeviceCtx *ctx, uint64_t lo, uint64_t hi) {
            return recover_batch_on(ctx, recovered_cells ? recovered_cells + lo * CELLS_PER_EXT_BLOB : nullptr,
                                    recovered_proofs ? recovered_proofs + lo * CELLS_PER_EXT_BLOB : nullptr,
                                    status ? status + lo : nullptr, cell_indices, cells + lo * num_cells, num_cells,
                                    hi - lo, s);
        });
    });
}

// eip7594.c:177-304.  Concurrent callers that hold the SAME set of columns (the PeerDAS case: a node
// reconstructs every blob of a block from the columns it custodies) share one launch of the batch path, which
// builds the vanishing polynomial of the missing set once (combiner.hpp; the key is the index list + the outputs
// wanted).
extern "C" C_KZG_RET recover_cells_and_kzg_proofs(Cell *recovered_cells, KZGProof *recovered_proofs,
                                                  const uint64_t *cell_indices, const Cell *cells,
                                                  uint64_t num_cells, const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        auto solo = [&]() -> C_KZG_RET {
            return ckzg_hip_recover_cells_and_kzg_proofs_batch(recovered_cells, recovered_proofs, NULL, cell_indices,
                                                               cells, num_cells, 1, s);
        };
        SettingsCtx *sc = settings_of(s, false);
        Combiner *cb = sc ? sc->comb[CB_RECOVER] : nullptr;
        // arguments the batch entry point rejects outright never queue (eip7594.c:191-213)
        if (!cb || (recovered_cells == NULL && recovered_proofs == NULL) || num_cells > CELLS_PER_EXT_BLOB ||
            num_cells < CELLS_PER_BLOB || cell_indices == NULL || cells == NULL)
            return solo();
        std::vector<uint64_t> key(num_cells + 1);
        key[0] = (recovered_cells ? 1u : 0u) | (recovered_proofs ? 2u : 0u);
        memcpy(key.data() + 1, cell_indices, num_cells * sizeof(uint64_t));
        const size_t in_per = (size_t)num_cells * BYTES_PER_CELL;
        const size_t cells_per = (size_t)CELLS_PER_EXT_BLOB * BYTES_PER_CELL, proofs_per = (size_t)CELLS_PER_EXT_BLOB * 48;
        return cb->submit(
            key.data(), key.size() * sizeof(uint64_t), solo,
            [&](uint8_t *h_in, size_t idx) { memcpy(h_in + idx * in_per, cells, in_per); },
            [&](const uint8_t *h_in, uint8_t *h_out, uint8_t *st, size_t n) -> C_KZG_RET {
                return ckzg_hip_recover_cells_and_kzg_proofs_batch(
                    recovered_cells ? reinterpret_cast<Cell *>(h_out) : nullptr,
                    recovered_proofs ? reinterpret_cast<KZGProof *>(h_out + (recovered_cells ? n * cells_per : 0)) : nullptr, st,
                    cell_indices, reinterpret_cast<const Cell *>(h_in), num_cells, n, s);
            },
            [&](const uint8_t *h_out, size_t idx, size_t n) {
                if (recovered_cells) memcpy(recovered_cells, h_out + idx * cells_per, cells_per);
                if (recovered_proofs)
                    memcpy(recovered_proofs, h_out + (recovered_cells ? n * cells_per : 0) + idx * proofs_per, proofs_per);
            });
    });
}

// ------------------------------------------------------------------------------------------
// EIP-7594: cell proof batch verification
// ------------------------------------------------------------------------------------------

extern "C" C_KZG_RET compute_verify_cell_kzg_proof_batch_challenge(
    fr_t *challenge_out, const Bytes48 *commitments_bytes, uint64_t num_commitments,
    const uint64_t *commitment_indices, const uint64_t *cell_indices, const Cell *cells,
    const Bytes48 *proofs_bytes, uint64_t num_cells) {
    // eip7594.c:390-482
    Sha256 h;
    uint8_t head[48], idx[16], digest[32];
    memcpy(head, "RCKZGCBATCH__V1_", 16);
    be64(head + 16, FIELD_ELEMENTS_PER_BLOB);
    be64(head + 24, FIELD_ELEMENTS_PER_CELL);
    be64(head + 32, num_commitments);
    be64(head + 40, num_cells);
    h.update(head, 48);
    for (uint64_t i = 0; i < num_commitments; i++) h.update(commitments_bytes[i].bytes, 48);
    for (uint64_t i = 0; i < num_cells; i++) {
        be64(idx, commitment_indices[i]);
        be64(idx + 8, cell_indices[i]);
        h.update(idx, 16);
        h.update(cells[i].bytes, BYTES_PER_CELL);
        h.update(proofs_bytes[i].bytes, 48);
    }
    h.finish(digest);
    *as_fr(challenge_out) = fr_from_bytes_reduce(digest);
    return C_KZG_OK;
}

static C_KZG_RET verify_cells_on(dev::DeviceCtx *ctx, bool *ok, const Bytes48 *commitments_bytes,
                                 const uint64_t *cell_indices, const Cell *cells, const Bytes48 *proofs_bytes,
                                 uint64_t num_cells, const KZGSettings *s) {
    *ok = false;
    const size_t n = num_cells, l = FIELD_ELEMENTS_PER_CELL;
    Trace tr("verify_cells");
    const Fr *rou = as_fr(s->roots_of_unity);
    // deduplicate commitments (eip7594.c:345-376)
    std::vector<Bytes48> uniq;
    std::vector<uint64_t> cidx(n);
    for (size_t i = 0; i < n; i++) {
        size_t j;
        for (j = 0; j < uniq.size(); j++) {
            if (memcmp(uniq[j].bytes, commitments_bytes[i].bytes, 48) == 0) break;
        }
        if (j == uniq.size()) uniq.push_back(commitments_bytes[i]);
        cidx[i] = j;
    }
    const size_t nc = uniq.size();
    // Call-time table (msm.hip), as in verify_blobs_core: the transcript of a cell batch is ONE SHA-256 stream over
    // every cell (1 us per cell on a SHA-NI core) during which the GPU has nothing to do once the points are
    // decompressed -- time enough to build a 6-bit fixed-base table (accumulator form, ~1.2 ms of latency + 0.15 us per
    // point) over the batch's proofs, its distinct commitments and the 64 setup points of the interpolation
    // commitment, so that the four sums that follow the challenge are table sums (0.45 ms) instead of ladders
    // (1.0-2.2 ms).  Measured with the table off / on (tools/bench_verify_cells.py, same box,
    // profiles/r03_verify_x28_sweep.txt): n = 1024 3.20 -> 2.88 ms, 2048 4.56 -> 3.20, 4096 7.54 -> 5.50,
    // 6144 9.81 -> 7.67, and n = 128 2.38 -> 2.31, 256 2.42 -> 2.32, 384 2.55 -> 2.36, 768 2.91 -> 2.49; below a
    // blob's worth of cells the two forms tie at the 2.3 ms latency floor of the call: from 128 cells upwards.
    static const int call_table_wbits = (int)dev::ab_knob("CKZG_HIP_VERIFY_TABLE_WBITS", 6);
    static const size_t cell_table_min = (size_t)dev::ab_knob("CKZG_HIP_VERIFY_CELL_TABLE_MIN", 128);
    bool use_table = g_verify_call_table.load(std::memory_order_relaxed) != 0 && call_table_wbits >= 4 && call_table_wbits <= 10 &&
                     n >= cell_table_min;
    size_t npts = n + nc + (use_table ? l : 0);   // proofs, distinct commitments [, g1_values_monomial[0..63]]
    dev::FixedBaseTable tbl;
    size_t tbl_bytes = 0, tbl_tmp = 0, sums_scratch = 0;
    Arena &ar = ctx->api_arena;
    const size_t plain_bytes = (n + nc + l) * (48 + 2 + sizeof(G1Affine)) + ((size_t)CELLS_PER_EXT_BLOB * l + l + n * l + n) * sizeof(Fr) +
                               n * BYTES_PER_CELL + (n + CELLS_PER_EXT_BLOB + 1 + n) * 4 + 4096;
    if (use_table) {
        dev::call_table_geometry(&tbl, (int)npts, call_table_wbits);
        tbl_bytes = dev::call_table_bytes(tbl);
        tbl_tmp = dev::call_table_tmp_bytes(tbl);
        sums_scratch = dev::table_sums_scratch_bytes(tbl, 4);
        // the table is an optimisation: a device too full for it still verifies, by ladders
        if (!ar.begin(plain_bytes + tbl_bytes + tbl_tmp + sums_scratch + 4 * npts * 32 + (2 * n + nc + 1) * 4 + 4096)) {
            use_table = false;
            tbl_bytes = tbl_tmp = sums_scratch = 0;
            npts = n + nc;
        }
    }
    if (!use_table) OKM(ar.begin(plain_bytes));
    ABuf<uint8_t> d_ptb(ar, (n + nc) * 48), d_st(ar, n + nc), d_st2(ar, n + nc), d_cells(ar, n * BYTES_PER_CELL);
    ABuf<G1Affine> d_pts(ar, npts);
    ABuf<uint8_t> d_tbl(ar, use_table ? tbl_bytes : 1), d_tbl_tmp(ar, use_table ? tbl_tmp : 1), d_sums_scr(ar, use_table ? sums_scratch : 1);
    ABuf<uint32_t> d_sc(ar, use_table ? 4 * npts * 8 : 1);
    ABuf<uint32_t> d_grp(ar, use_table ? 2 * n + nc + 1 : 1);   // column of each cell [n] | commitment groups: start [nc + 1], members [n]
    ABuf<G1XYZZ> d_sums(ar, 4);
    OKM(d_tbl.p && d_tbl_tmp.p && d_sums_scr.p && d_sc.p && d_grp.p && d_sums.p);
    ABuf<Fr> d_agg(ar, (size_t)CELLS_PER_EXT_BLOB * l), d_interp(ar, l), d_cellfr(ar, n * l), d_rp(ar, n);
    ABuf<uint32_t> d_bad(ar, n), d_csr(ar, CELLS_PER_EXT_BLOB + 1 + n);
    OKM(d_ptb.p && d_st.p && d_st2.p && d_cells.p && d_pts.p && d_agg.p && d_interp.p && d_cellfr.p && d_rp.p &&
        d_bad.p && d_csr.p);
    ArenaTrim trim(ar);
    tr.mark("dedup");
    // The transcript is ONE SHA-256 stream over every cell (eip7594.c:390-482): the longest thing in this call, on one
    // host core.  It starts now, on a worker thread, and everything below that does not need the challenge -- the
    // copies (blocking, from pageable memory), the GPU validation, the grouping of cells by column and by commitment,
    // the check of the validation flags -- happens underneath it.
    Fr r;
    struct HashJob {
        std::atomic<uint32_t> running{0};   // futex word
        bool submitted = false;
        void wait() {
            if (submitted) wait_host_work_done(&running, "cell transcript hash job");
        }
        ~HashJob() { wait(); }   // nothing the worker reads or writes may die before it is through
    } hash_job;
    const uint64_t *cidx_p = cidx.data();
    const Bytes48 *uniq_p = uniq.data();
    auto hash_all = [&r, uniq_p, nc, cidx_p, cell_indices, cells, proofs_bytes, n]() {
        compute_verify_cell_kzg_proof_batch_challenge((fr_t *)&r, uniq_p, nc, cidx_p, cell_indices, cells, proofs_bytes, n);
    };
    if (n >= 256) {
        HashJob *hj = &hash_job;
        hash_job.running.store(1, std::memory_order_relaxed);
        hash_job.submitted = WorkerPool::get().submit([hash_all, hj]() {
            hash_all();
            hj->running.store(0, std::memory_order_release);
            futex_wake(&hj->running, INT_MAX);
        });
    }
    OKM(ensure_pinned(ctx->h_out, ctx->h_out_bytes, 2 * (n + nc) + n * 4));
    uint8_t *h_st = static_cast<uint8_t *>(ctx->h_out[0]), *h_st2 = h_st + (n + nc);
    uint32_t *h_bad = static_cast<uint32_t *>(ctx->h_out[1]);
    // proofs [0,n), unique commitments [n, n+nc): decompression and subgroup checks start on the GPU,
    // followed by the cells' bytes -> Fr conversion, while the transcript is being hashed
    // (all copies from pageable memory first: such a copy returns only when it is done, so it must not
    // queue behind the validation kernel)
    OKB(hipMemcpyAsync(d_ptb.p, proofs_bytes, n * 48, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
    OKB(hipMemcpyAsync(d_ptb.p + n * 48, uniq.data(), nc * 48, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
    OKB(hipMemcpyAsync(d_cells.p, cells, n * BYTES_PER_CELL, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
    OKB(hipMemsetAsync(d_bad.p, 0, n * 4, ctx->stream) == hipSuccess);
    RC(dev::bytes_to_fr_batch(ctx, d_cellfr.p, d_bad.p, d_cells.p, n * l, (uint32_t)l));
    OKB(hipMemcpyAsync(h_bad, d_bad.p, n * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
    // Validation in two launches: decompression (a square root, ~0.35 ms) here, the subgroup test
    // (~1 ms of dependent doublings) on the second stream, next to the table build that already uses the points.
    // A point outside the subgroup makes table and sums meaningless, not unsafe; the call ends in BADARGS below.
    RC(dev::decompress_g1_batch_device(ctx, d_pts.p, d_st.p, d_ptb.p, n + nc));
    OKB(hipMemcpyAsync(h_st, d_st.p, n + nc, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess);
    if (use_table)
        OKB(hipMemcpyAsync(d_pts.p + n + nc, ctx->d_mono, l * sizeof(G1Affine), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess);
    for (int i = 0; i < 4; i++) {
        if (!ctx->stage_ev[i]) OKB(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming) == hipSuccess);
    }
    OKB(hipEventRecord(ctx->stage_ev[0], ctx->stream) == hipSuccess);
    OKB(hipStreamWaitEvent(ctx->copy_stream, ctx->stage_ev[0], 0) == hipSuccess);
    RC(dev::subgroup_g1_batch_device(ctx, d_st2.p, d_pts.p, n + nc, ctx->copy_stream));
    OKB(hipMemcpyAsync(h_st2, d_st2.p, n + nc, hipMemcpyDeviceToHost, ctx->copy_stream) == hipSuccess);
    OKB(hipEventRecord(ctx->stage_ev[1], ctx->copy_stream) == hipSuccess);
    // whatever path leaves this function, the other streams must be idle before the arena is reused
    struct StreamDrain {
        hipStream_t s;
        ~StreamDrain() {
            if (s) (void)dev::sync_stream(s);
        }
    } drain{ctx->copy_stream};
    if (use_table) {
        if (!ctx->aux_stream) OKB(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) == hipSuccess);
        OKB(hipStreamWaitEvent(ctx->aux_stream, ctx->stage_ev[0], 0) == hipSuccess);
        RC(dev::call_table_enqueue(ctx->aux_stream, &tbl, d_tbl.p, d_tbl_tmp.p, d_pts.p));
        OKB(hipEventRecord(ctx->stage_ev[2], ctx->aux_stream) == hipSuccess);
    }
    StreamDrain drain_aux{use_table ? ctx->aux_stream : nullptr};
    // cells grouped by column (counting sort) for the aggregation kernel
    std::vector<uint32_t> csr(CELLS_PER_EXT_BLOB + 1 + n, 0);
    for (size_t i = 0; i < n; i++) csr[cell_indices[i] + 1]++;
    for (size_t c = 0; c < CELLS_PER_EXT_BLOB; c++) csr[c + 1] += csr[c];
    {
        std::vector<uint32_t> fill(csr.begin(), csr.begin() + CELLS_PER_EXT_BLOB);
        for (size_t i = 0; i < n; i++) csr[CELLS_PER_EXT_BLOB + 1 + fill[cell_indices[i]]++] = (uint32_t)i;
    }
    OKB(hipMemcpyAsync(d_csr.p, csr.data(), csr.size() * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
    if (use_table) {
        // ... and by commitment, for the weights; the column of each cell, for its coset factor (k_cell_rlc_scalars)
        std::vector<uint32_t> grp(2 * n + nc + 1, 0);
        uint32_t *col = grp.data(), *start = grp.data() + n, *members = grp.data() + n + nc + 1;
        for (size_t i = 0; i < n; i++) {
            col[i] = (uint32_t)cell_indices[i];
            start[cidx[i] + 1]++;
        }
        for (size_t j = 0; j < nc; j++) start[j + 1] += start[j];
        std::vector<uint32_t> fill(start, start + nc);
        for (size_t i = 0; i < n; i++) members[fill[cidx[i]]++] = (uint32_t)i;
        OKB(hipMemcpyAsync(d_grp.p, grp.data(), grp.size() * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess);
        OKB(hipMemsetAsync(d_sc.p, 0, 4 * npts * 32, ctx->stream) == hipSuccess);
    }
    // The validation flags.  A large batch: both streams are through long before the transcript is, so they are
    // checked here, underneath it.  A small one (ladder sums): the subgroup test (~1 ms of dependent doublings) keeps
    // running on the second stream next to the sums, and the flags are checked after those.
    auto flags_ok = [&]() -> C_KZG_RET {
        OKB(dev::sync_event(ctx->stage_ev[3]) == hipSuccess && dev::sync_event(ctx->stage_ev[1]) == hipSuccess);
        for (size_t i = 0; i < n + nc; i++) {
            if (h_st[i] || h_st2[i]) return C_KZG_BADARGS;  // bad encoding / off the curve / outside G1
        }
        for (size_t i = 0; i < n; i++) {
            if (h_bad[i]) return C_KZG_BADARGS;  // a non-canonical field element in a cell (bytes.c:67)
        }
        return C_KZG_OK;
    };
    OKB(hipEventRecord(ctx->stage_ev[3], ctx->stream) == hipSuccess);
    if (use_table) RC(flags_ok());
    tr.mark("copies, validation, grouping (underneath the transcript hash)");
    if (hash_job.submitted)
        hash_job.wait();
    else
        hash_all();
    tr.mark("transcript hash");
    const size_t row = npts * 8;   // words per scalar vector of the table path
    std::vector<RawScalar> rp_raw, wrp_raw, wts_raw;
    if (use_table) {
        // scalars made on the GPU from r: [proofs | distinct commitments | 64 monomial setup points] x 4 vectors
        RC(dev::cell_rlc_scalars_enqueue(ctx, d_rp.p, d_sc.p + 0 * row, d_sc.p + 2 * row, d_sc.p + 1 * row + n * 8, d_grp.p,
                                         d_grp.p + n, d_grp.p + n + nc + 1, r, n, nc));
    } else {
        std::vector<Fr> rp(n);
        {
            Fr pw = Fr::one();
            for (size_t i = 0; i < n; i++) {
                rp[i] = pw;
                pw = mul(pw, r);
            }
        }
        rp_raw.resize(n);
        wrp_raw.resize(n);
        wts_raw.resize(nc);
        std::vector<Fr> wts(nc, Fr::zero());
        for (size_t i = 0; i < n; i++) {
            rp_raw[i] = raw_of(rp[i]);
            wts[cidx[i]] = add(wts[cidx[i]], rp[i]);
            size_t rb = reverse_bits_limited(CELLS_PER_EXT_BLOB, cell_indices[i]);
            wrp_raw[i] = raw_of(mul(rp[i], rou[rb * l]));  // r^i * h_k^64 (eip7594.c:784-812)
        }
        for (size_t j = 0; j < nc; j++) wts_raw[j] = raw_of(wts[j]);
        OKB(d_rp.up(rp.data(), n));
    }
    tr.mark("powers of r + weights");
    // aggregated column data: sum of r^i * cell_i per column (eip7594.c:661-683)
    RC(dev::cell_aggregate_device(ctx, d_agg.p, d_cellfr.p, d_rp.p, d_csr.p, d_csr.p + CELLS_PER_EXT_BLOB + 1, n));
    // per column: cell data is in bit-reversed order -> DIT inverse NTT(64) gives the interpolation
    // polynomial over the coset; unused columns are all-zero and stay zero
    RC(dev::fr_ntt_batch(ctx, d_agg.p, CELLS_PER_EXT_BLOB, 6, false, true, true));
    RC(dev::interp_sum_device(ctx, d_interp.p, d_agg.p));
    // all four lincombs (eip7594.c:926, :530, :807 and the commitment to the aggregated interpolation
    // polynomial over the first 64 monomial setup points, :758) in one launch
    G1Jac lc[4];
    if (use_table) {
        OKB(hipMemcpyAsync(d_sc.p + 3 * row + (n + nc) * 8, d_interp.p, l * 32, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess);
        OKB(hipStreamWaitEvent(ctx->stream, ctx->stage_ev[2], 0) == hipSuccess);   // the table is complete
        RC(dev::table_sums_enqueue(ctx->stream, tbl, d_sums.p, d_sc.p, 4, d_sums_scr.p));
        G1XYZZ hs[4];
        OKB(d_sums.down(hs, 4));
        for (int j = 0; j < 4; j++) lc[j] = jac_from_xyzz(hs[j]);
    } else {
        LincombJob jobs[4] = {{d_pts.p, &rp_raw}, {d_pts.p + n, &wts_raw}, {d_pts.p, &wrp_raw},
                              {ctx->d_mono, nullptr, (const RawScalar *)d_interp.p, l}};
        C_KZG_RET ret = gpu_lincomb_multi(ctx, lc, jobs, 4);
        if (ret != C_KZG_OK) return ret;
        RC(flags_ok());   // a point outside G1 makes the sums above meaningless, not unsafe: discarded
    }
    tr.mark("aggregation + IFFTs + four lincombs");
    const G1Jac &proof_lc = lc[0], &csum = lc[1], &wsum = lc[2], &interp_commit = lc[3];
    G1Jac final_sum = jac_add(jac_add(csum, jac_neg(interp_commit)), wsum);
    // e(final_sum, G2) == e(proof_lc, [s^64]G2)
    *ok = pairing_product_is_one(jac_to_affine_fast(final_sum), prepared_of(ctx)->gen, jac_to_affine_fast(jac_neg(proof_lc)),
                                 prepared_of(ctx)->s64);
    tr.mark("pairing check");
    return C_KZG_OK;
}

extern "C" C_KZG_RET verify_cell_kzg_proof_batch(bool *ok, const Bytes48 *commitments_bytes,
                                                 const uint64_t *cell_indices, const Cell *cells,
                                                 const Bytes48 *proofs_bytes, uint64_t num_cells,
                                                 const KZGSettings *s) {
    // eip7594.c:825-974
    return guarded([&]() -> C_KZG_RET {
        *ok = false;
        if (num_cells == 0) {
            *ok = true;
            return C_KZG_OK;
        }
        for (size_t i = 0; i < num_cells; i++) {
            if (cell_indices[i] >= CELLS_PER_EXT_BLOB) return C_KZG_BADARGS;
        }
        if (!settings_of(s)) return C_KZG_ERROR;
        // several devices: contiguous shards of cells, each with its own challenge and pairing check, AND-ed
        std::atomic<int> all_ok(1);
        C_KZG_RET ret = for_each_device_shard(s, num_cells, 2048, [&](dev::DeviceCtx *ctx, uint64_t lo, uint64_t hi) {
            bool res = false;
            C_KZG_RET r = verify_cells_on(ctx, &res, commitments_bytes + lo, cell_indices + lo, cells + lo,
                                          proofs_bytes + lo, hi - lo, s);
            if (r == C_KZG_OK && !res) all_ok.store(0);
            return r;
        });
        if (ret == C_KZG_OK) *ok = all_ok.load() != 0;
        return ret;
    });
}

// ------------------------------------------------------------------------------------------
// g1_lincomb_fast (src/common/lincomb.c:65-123) at the boundary: the variable-base sum on its own
// ------------------------------------------------------------------------------------------

extern "C" C_KZG_RET ckzg_hip_g1_lincomb(g1_t *out, const g1_t *p, const fr_t *coeffs, uint64_t len, int algo,
                                         const KZGSettings *s) {
    return guarded([&]() -> C_KZG_RET {
        if (algo < 0 || algo > 4) return C_KZG_BADARGS;
        Lease lease(s);
        dev::DeviceCtx *ctx = lease.ctx;
        if (!ctx) return C_KZG_ERROR;
        if (len == 0) {  // lincomb.c:76-79: the empty sum is the identity
            *as_g1(out) = G1Jac::inf();
            return C_KZG_OK;
        }
        std::vector<G1Affine> aff(len);
        std::vector<RawScalar> k(len);
        for (uint64_t i = 0; i < len; i++) {
            aff[i] = jac_to_affine_fast(*as_g1(&p[i]));
            k[i] = raw_of(*as_fr(&coeffs[i]));
        }
        Arena &ar = ctx->api_arena;
        OKM(ar.begin(len * (sizeof(G1Affine) + 1) + 1024));
        ABuf<G1Affine> d_pts(ar, len);
        ABuf<uint8_t> d_st(ar, len);
        OKM(d_pts.p && d_st.p);
        OKB(d_pts.up(aff.data(), len));
        // the kernels use the endomorphism: only valid on the prime-order subgroup (every caller inside the
        // library passes validated points; an outside caller gets the check here)
        RC(dev::subgroup_g1_batch_device(ctx, d_st.p, d_pts.p, len));
        OKB(dev::sync_stream(ctx->stream) == hipSuccess);
        std::vector<uint8_t> st(len);
        OKB(d_st.down(st.data(), len));
        for (uint8_t b : st) {
            if (b) return C_KZG_BADARGS;
        }
        LincombJob job{d_pts.p, &k};
        G1Jac r;
        C_KZG_RET ret = gpu_lincomb_multi(ctx, &r, &job, 1, algo);
        if (ret != C_KZG_OK) return ret;
        *as_g1(out) = r;
        return C_KZG_OK;
    });
}
