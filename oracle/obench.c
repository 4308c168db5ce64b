/* obench.c -- TEST / MEASUREMENT INFRASTRUCTURE (part of the CPU oracle, never of the product).
 * N native threads, each calling the oracle's blob_to_kzg_commitment on its own copy of a blob, on a shared
 * OKZGSettings: the shape of the reference's parallel benchmarks (bindings/go/main_test.go:953-971, one goroutine per
 * blob).  bench.py's cpu_baseline.all_cores uses it so that the many-core CPU figure does not depend on Python
 * threads, the GIL or ctypes (round-4 review: "use the pthread driver against the oracle so the harness is not a
 * suspect"). */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "okzg.h"

typedef struct {
    const OKZGSettings *s;
    uint8_t *blob;
    uint8_t out[48];
    int calls, bad;
    atomic_int *go;
} OWorker;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + t.tv_nsec * 1e-9;
}

static void *oworker(void *arg) {
    OWorker *w = (OWorker *)arg;
    while (!atomic_load(w->go)) sched_yield();
    for (int i = 0; i < w->calls; i++)
        if (okzg_blob_to_kzg_commitment(w->out, w->blob, w->s) != 0) w->bad++;
    return NULL;
}

/* Returns commitments per second over all threads (wall clock from the common start to the last join), or a negative
 * number: -1 allocation / thread start failed, -2 a call failed, -3 the threads disagree on the commitment.
 * first_out (48 bytes, may be NULL) receives thread 0's commitment for the caller to check. */
double okzg_bench_commit_threads(const OKZGSettings *s, const uint8_t *blob, int threads, int calls_per_thread,
                                 uint8_t *first_out) {
    if (threads < 1 || threads > 4096 || calls_per_thread < 1) return -1.0;
    OWorker *w = (OWorker *)calloc((size_t)threads, sizeof(OWorker));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    atomic_int go = 0;
    int started = 0;
    double rate = -1.0;
    if (!w || !th) goto out;
    for (int i = 0; i < threads; i++) {
        w[i].s = s;
        w[i].calls = calls_per_thread;
        w[i].go = &go;
        w[i].blob = (uint8_t *)malloc(131072);
        if (!w[i].blob) break;
        memcpy(w[i].blob, blob, 131072);
        if (pthread_create(&th[i], NULL, oworker, &w[i]) != 0) {
            free(w[i].blob);
            w[i].blob = NULL;
            break;
        }
        started++;
    }
    const double t0 = now_s();
    if (started != threads)
        for (int i = 0; i < started; i++) w[i].calls = 0;
    atomic_store(&go, 1);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    const double dt = now_s() - t0;
    if (started == threads) {
        rate = (double)threads * calls_per_thread / dt;
        for (int i = 0; i < threads; i++) {
            if (w[i].bad) rate = -2.0;
            else if (memcmp(w[i].out, w[0].out, 48) != 0) rate = -3.0;
        }
        if (first_out) memcpy(first_out, w[0].out, 48);
    }
out:
    if (w)
        for (int i = 0; i < started; i++) free(w[i].blob);
    free(w);
    free(th);
    return rate;
}
