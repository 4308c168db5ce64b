/* fanout_callers.c -- N threads, each making ONE-UNIT calls of the unchanged ckzg.h API on a shared KZGSettings for a
 * fixed time: the shape of the reference's parallel benchmarks (bindings/go/main_test.go:953-971: goroutines that
 * each call BlobToKZGCommitment / ComputeCellsAndKZGProofs on their own blob).  Plain C against include/ckzg.h;
 * the library under test is whatever defines those symbols in the process (bench.py and the tests load
 * libckzg_hip.so RTLD_GLOBAL first, then this file's libckzg_callers.so (fanout.py)).  Test / measurement infrastructure, not product.
 *
 * Every thread owns one input and one output buffer; results of the LAST call of each thread are left in `outs`
 * so that the caller can check them against the oracle.  Return codes are counted per thread. */
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ckzg.h"

enum { OP_COMMIT = 0, OP_CELLS_PROOFS = 1, OP_BLOB_PROOF = 2, OP_RECOVER = 3, OP_CELLS = 4, OP_PROOFS = 5, OP_VERIFY_BLOB = 6 };

#define HIST_BINS 200       /* 8 bins per octave from 1 us: bin 199 starts at ~30 s */
static int hist_bin(double ms) {
    const double us = ms * 1e3;
    if (us <= 1.0) return 0;
    int b = (int)(8.0 * log2(us));
    return b < 0 ? 0 : (b >= HIST_BINS ? HIST_BINS - 1 : b);
}
static double hist_bin_upper_ms(int b) { return exp2((b + 1) / 8.0) * 1e-3; }

typedef struct {
    int op, id;
    const KZGSettings *s;
    const uint8_t *in;        /* this thread's blob (or its recover input cells) */
    const uint8_t *aux;       /* commitment (OP_BLOB_PROOF) / uint64_t cell indices (OP_RECOVER) */
    uint64_t aux_n;           /* number of cells (OP_RECOVER) */
    uint8_t *out;             /* 48 B | 128 cells + 128 proofs */
    atomic_int *go, *stop;
    uint64_t calls, not_ok;
    int last_ret;
    double worst_ms, total_ms;
    uint64_t max_calls;       /* 0: until *stop */
    uint32_t hist[HIST_BINS]; /* call durations, 8 bins per octave from 1 us (percentiles over all threads) */
} Worker;

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int one_call(Worker *w) {
    const size_t cells_b = (size_t)CELLS_PER_EXT_BLOB * BYTES_PER_CELL;
    switch (w->op) {
        case OP_COMMIT: return blob_to_kzg_commitment((KZGCommitment *)w->out, (const Blob *)w->in, w->s);
        case OP_CELLS_PROOFS:
            return compute_cells_and_kzg_proofs((Cell *)w->out, (KZGProof *)(w->out + cells_b), (const Blob *)w->in, w->s);
        case OP_CELLS: return compute_cells_and_kzg_proofs((Cell *)w->out, NULL, (const Blob *)w->in, w->s);
        case OP_PROOFS: return compute_cells_and_kzg_proofs(NULL, (KZGProof *)(w->out + cells_b), (const Blob *)w->in, w->s);
        case OP_BLOB_PROOF: return compute_blob_kzg_proof((KZGProof *)w->out, (const Blob *)w->in, (const Bytes48 *)w->aux, w->s);
        case OP_RECOVER:
            return recover_cells_and_kzg_proofs((Cell *)w->out, (KZGProof *)(w->out + cells_b), (const uint64_t *)w->aux,
                                                (const Cell *)w->in, w->aux_n, w->s);
        case OP_VERIFY_BLOB: {   /* aux = commitment | proof; the verdict goes to out[0] */
            bool ok = false;
            const int r = verify_blob_kzg_proof(&ok, (const Blob *)w->in, (const Bytes48 *)w->aux, (const Bytes48 *)(w->aux + 48), w->s);
            w->out[0] = ok ? 1 : 0;
            return r;
        }
        default: return C_KZG_BADARGS;
    }
}

static void *worker_main(void *arg) {
    Worker *w = (Worker *)arg;
    while (!atomic_load(w->go)) sched_yield();
    while (!atomic_load(w->stop) && (w->max_calls == 0 || w->calls < w->max_calls)) {
        const double t0 = now_ms();
        const int r = one_call(w);
        const double dt = now_ms() - t0;
        w->calls++;
        w->total_ms += dt;
        if (dt > w->worst_ms) w->worst_ms = dt;
        w->hist[hist_bin(dt)]++;
        if (r != C_KZG_OK) w->not_ok++;
        w->last_ret = r;
    }
    return NULL;
}

/* Runs `threads` callers of `op` for `seconds` (or `max_calls` calls each when > 0).
 * ins: threads buffers of in_stride bytes; auxs: threads buffers of aux_stride bytes (or NULL; aux_stride 0 = one
 * shared buffer); outs: threads buffers of out_stride bytes.
 * stats[0] = calls, [1] = calls with a non-OK return, [2] = wall seconds, [3] = worst single call in ms,
 * [4] = mean call in ms, [5] / [6] / [7] = p50 / p99 / p99.9 of the call durations in ms (upper edge of the histogram
 * bin, 8 bins per octave: +9 % at most); last_rets[threads] = return code of every thread's last call (may be NULL).
 * `stats` must hold 8 doubles.
 * Returns 0, or -1 if the threads could not be started. */
int callers_run(const KZGSettings *s, int op, int threads, double seconds, uint64_t max_calls, const uint8_t *ins,
                uint64_t in_stride, const uint8_t *auxs, uint64_t aux_stride, uint64_t aux_n, uint8_t *outs,
                uint64_t out_stride, double *stats, int *last_rets) {
    if (threads < 1 || threads > 4096) return -1;
    Worker *w = (Worker *)calloc((size_t)threads, sizeof(Worker));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!w || !th) {
        free(w);
        free(th);
        return -1;
    }
    atomic_int go = 0, stop = 0;
    int started = 0;
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 1 << 20);
    for (int i = 0; i < threads; i++) {
        w[i].op = op;
        w[i].id = i;
        w[i].s = s;
        w[i].in = ins + (uint64_t)i * in_stride;
        w[i].aux = auxs ? auxs + (uint64_t)i * aux_stride : NULL;
        w[i].aux_n = aux_n;
        w[i].out = outs + (uint64_t)i * out_stride;
        w[i].go = &go;
        w[i].stop = &stop;
        w[i].max_calls = max_calls;
        if (pthread_create(&th[i], &attr, worker_main, &w[i]) != 0) break;
        started++;
    }
    pthread_attr_destroy(&attr);
    const double t0 = now_ms();
    atomic_store(&go, 1);
    if (started == threads && max_calls == 0) {
        struct timespec nap = {(time_t)seconds, (long)((seconds - (double)(time_t)seconds) * 1e9)};
        nanosleep(&nap, NULL);
    }
    if (started != threads || max_calls == 0) atomic_store(&stop, 1);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    const double wall = (now_ms() - t0) * 1e-3;
    double calls = 0, bad = 0, worst = 0, total = 0;
    for (int i = 0; i < started; i++) {
        calls += (double)w[i].calls;
        bad += (double)w[i].not_ok;
        total += w[i].total_ms;
        if (w[i].worst_ms > worst) worst = w[i].worst_ms;
        if (last_rets) last_rets[i] = w[i].last_ret;
    }
    if (stats) {
        stats[0] = calls;
        stats[1] = bad;
        stats[2] = wall;
        stats[3] = worst;
        stats[4] = calls > 0 ? total / calls : 0;
        static const double pct[3] = {0.50, 0.99, 0.999};
        for (int k = 0; k < 3; k++) {
            const double want = pct[k] * calls;
            double seen = 0;
            stats[5 + k] = 0;
            for (int b = 0; b < HIST_BINS && calls > 0; b++) {
                for (int i = 0; i < started; i++) seen += (double)w[i].hist[b];
                if (seen >= want) {
                    stats[5 + k] = hist_bin_upper_ms(b) < worst ? hist_bin_upper_ms(b) : worst;
                    break;
                }
            }
        }
    }
    free(w);
    free(th);
    return started == threads ? 0 : -1;
}
