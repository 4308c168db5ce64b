// TEST INFRASTRUCTURE (never linked into the product): an LD_PRELOAD interposer that makes chosen device / page-locked
// allocations of the process fail with hipErrorOutOfMemory, so that tests/test_gpu_alloc_failures.py can walk every
// allocation of every entry point and check the reference's error convention -- C_KZG_MALLOC, nothing leaked, the
// next call fine (src/common/ret.h:24-29, src/common/alloc.c:34-50; the reference has no such test, its allocations
// are plain calloc).
//
//   failalloc_arm(n, sticky)   the n-th allocation from now (0-based) fails; sticky: so does every later one
//   failalloc_class(c)         what "allocation" means: 0 device / page-locked memory (default), 1 streams and events
//                              (those fail with hipErrorOutOfMemory too; the library reports C_KZG_ERROR or _MALLOC)
//   failalloc_disarm()
//   failalloc_fired()          allocations failed since the last arm
//   failalloc_seen()           allocations seen since the last arm
//   failalloc_free_bytes()     hipMemGetInfo's free figure (leak checks)
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <stdatomic.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#define HIP_ERROR_OUT_OF_MEMORY 2

static atomic_long g_countdown = -1;  // < 0: disarmed
static atomic_int g_sticky = 0, g_class = 0;
static atomic_long g_fired = 0, g_seen = 0;

void failalloc_arm(long nth, int sticky) {
    atomic_store(&g_fired, 0);
    atomic_store(&g_seen, 0);
    atomic_store(&g_sticky, sticky);
    atomic_store(&g_countdown, nth);
}
void failalloc_class(int c) { atomic_store(&g_class, c); }
void failalloc_disarm(void) { atomic_store(&g_countdown, -1); }
long failalloc_fired(void) { return atomic_load(&g_fired); }
long failalloc_seen(void) { return atomic_load(&g_seen); }

static int this_one_fails(int cls) {
    if (cls != atomic_load(&g_class)) return 0;
    atomic_fetch_add(&g_seen, 1);
    long c = atomic_load(&g_countdown);
    for (;;) {
        if (c < 0) return 0;
        if (c == 0) {
            if (atomic_load(&g_sticky)) break;  // stays at 0: every later allocation fails too
            if (atomic_compare_exchange_weak(&g_countdown, &c, -1)) break;
            continue;
        }
        if (atomic_compare_exchange_weak(&g_countdown, &c, c - 1)) return 0;
    }
    atomic_fetch_add(&g_fired, 1);
    return 1;
}

// The HIP runtime arrives as a dependency of a dlopen()ed library (RTLD_LOCAL), so RTLD_NEXT does not see it: find the
// loaded object by name and ask it directly.
static int find_hip(struct dl_phdr_info *info, size_t size, void *out) {
    (void)size;
    if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) {
        *(void **)out = dlopen(info->dlpi_name, RTLD_NOLOAD | RTLD_LAZY);
        return 1;
    }
    return 0;
}
static void *next_symbol(const char *name) {
    void *f = dlsym(RTLD_NEXT, name);
    if (!f) {
        void *h = NULL;
        dl_iterate_phdr(find_hip, &h);
        if (h) f = dlsym(h, name);
    }
    if (!f) {
        fprintf(stderr, "failalloc: %s not found behind the interposer\n", name);
        abort();
    }
    return f;
}

int hipMalloc(void **p, size_t bytes) {
    static int (*real)(void **, size_t);
    if (!real) real = (int (*)(void **, size_t))next_symbol("hipMalloc");
    if (this_one_fails(0)) {
        if (p) *p = NULL;
        return HIP_ERROR_OUT_OF_MEMORY;
    }
    return real(p, bytes);
}

int hipHostMalloc(void **p, size_t bytes, unsigned flags) {
    static int (*real)(void **, size_t, unsigned);
    if (!real) real = (int (*)(void **, size_t, unsigned))next_symbol("hipHostMalloc");
    if (this_one_fails(0)) {
        if (p) *p = NULL;
        return HIP_ERROR_OUT_OF_MEMORY;
    }
    return real(p, bytes, flags);
}

int hipStreamCreateWithFlags(void **stream, unsigned flags) {
    static int (*real)(void **, unsigned);
    if (!real) real = (int (*)(void **, unsigned))next_symbol("hipStreamCreateWithFlags");
    if (this_one_fails(1)) {
        if (stream) *stream = NULL;
        return HIP_ERROR_OUT_OF_MEMORY;
    }
    return real(stream, flags);
}

int hipEventCreateWithFlags(void **event, unsigned flags) {
    static int (*real)(void **, unsigned);
    if (!real) real = (int (*)(void **, unsigned))next_symbol("hipEventCreateWithFlags");
    if (this_one_fails(1)) {
        if (event) *event = NULL;
        return HIP_ERROR_OUT_OF_MEMORY;
    }
    return real(event, flags);
}

int hipEventCreate(void **event) {
    static int (*real)(void **);
    if (!real) real = (int (*)(void **))next_symbol("hipEventCreate");
    if (this_one_fails(1)) {
        if (event) *event = NULL;
        return HIP_ERROR_OUT_OF_MEMORY;
    }
    return real(event);
}

long long failalloc_free_bytes(void) {
    int (*info)(size_t *, size_t *) = (int (*)(size_t *, size_t *))next_symbol("hipMemGetInfo");
    size_t fr = 0, tot = 0;
    if (!info || info(&fr, &tot) != 0) return -1;
    return (long long)fr;
}
