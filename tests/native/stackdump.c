/* TEST INFRASTRUCTURE (never linked into the product): an LD_PRELOAD helper for child processes the tests start.
 * The GPU box has no debugger, so a child that stalls is asked to describe itself instead: tests/watchdog.py sends
 * SIGUSR2 to every thread of the child (tgkill), and each thread prints its own native backtrace as module(+offset)
 * lines -- resolvable afterwards with addr2line / llvm-symbolizer on the same image (tools/debug/resolve_bt.py).
 * The first thread that gets the signal also asks the product for its own view (ckzg_hip_debug_dump: combiner
 * batches, slot pools, outstanding waits), if a build of the library that has it is loaded.
 *   gcc -O1 -g -shared -fPIC tests/native/stackdump.c -o stackdump.so -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <link.h>
#include <signal.h>
#include <stdatomic.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static atomic_int g_lib_dump_done = 0;

static int find_ckzg(struct dl_phdr_info *info, size_t size, void *out) {
    (void)size;
    if (info->dlpi_name && strstr(info->dlpi_name, "libckzg_hip")) {
        *(void **)out = dlopen(info->dlpi_name, RTLD_NOLOAD | RTLD_LAZY);
        return 1;
    }
    return 0;
}

static void on_usr2(int sig) {
    (void)sig;
    void *bt[64];
    char line[160];
    const long tid = syscall(SYS_gettid);
    int n = snprintf(line, sizeof line, "\n== stackdump: thread %ld, native backtrace:\n", tid);
    if (write(2, line, (size_t)n) < 0) return;
    const int depth = backtrace(bt, 64);
    backtrace_symbols_fd(bt, depth, 2);
    n = snprintf(line, sizeof line, "== stackdump: end of thread %ld\n", tid);
    if (write(2, line, (size_t)n) < 0) return;
    /* (not async-signal-safe, and meant that way: the process is about to be killed; the dump only try-locks) */
    if (atomic_exchange(&g_lib_dump_done, 1) == 0) {
        void *h = NULL;
        dl_iterate_phdr(find_ckzg, &h);
        void (*dump)(int) = h ? (void (*)(int))dlsym(h, "ckzg_hip_debug_dump") : NULL;
        if (dump) dump(2);
    }
}

__attribute__((constructor)) static void install(void) {
    void *warm[4];
    (void)backtrace(warm, 4); /* loads libgcc's unwinder now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_usr2;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGUSR2, &sa, 0);
}
