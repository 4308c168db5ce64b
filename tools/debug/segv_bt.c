/* segv_bt.c -- LD_PRELOAD helper for the GPU box (no debugger there that may ptrace): on SIGSEGV / SIGBUS / SIGABRT print
 * the native backtrace of the faulting thread as module(+offset) lines -- resolvable afterwards with addr2line on the
 * same image -- then let the default action take place.  Diagnostic tooling, never loaded by the product.
 *   gcc -O1 -g -shared -fPIC tools/debug/segv_bt.c -o tools/debug/libsegv_bt.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t *si, void *uc) {
    (void)uc;
    void *bt[96];
    char line[128];
    int n = snprintf(line, sizeof line, "\n== segv_bt: signal %d, fault address %p, native backtrace:\n", sig, si ? si->si_addr : 0);
    if (write(2, line, (size_t)n) < 0) _exit(99);
    int depth = backtrace(bt, 96);
    backtrace_symbols_fd(bt, depth, 2);
    n = snprintf(line, sizeof line, "== segv_bt: end\n");
    if (write(2, line, (size_t)n) < 0) _exit(99);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    void *warm[4];
    (void)backtrace(warm, 4); /* loads libgcc's unwinder now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
