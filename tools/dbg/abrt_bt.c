// LD_PRELOAD helper for hunting native crashes in the test suite: prints the native backtrace of the thread that
// raised SIGABRT / SIGSEGV (glibc's "free(): invalid pointer" aborts give no stack of their own).  Debug tool only.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig) {
    void *bt[64];
    const char *msg = sig == SIGABRT ? "\n=== native backtrace (SIGABRT) ===\n" : "\n=== native backtrace (SIGSEGV) ===\n";
    (void)!write(2, msg, strlen(msg));
    int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void) {
    void *bt[4];
    backtrace(bt, 4);  // loads libgcc now, not inside the handler
    signal(SIGABRT, handler);
    signal(SIGSEGV, handler);
}
