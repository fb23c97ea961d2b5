/* tools/segv_trace.c -- LD_PRELOAD helper for diagnosis runs (not part of the library): a native backtrace on
 * SIGSEGV / SIGABRT (module + offset: addr2line -e libquicked_hip.so), then the previous disposition.
 *   LD_PRELOAD=tools/bin/libsegvtrace.so python -m pytest -p no:faulthandler -s ...                                  */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fatal(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "[segv_trace] fatal signal, native backtrace of the raising thread:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    signal(SIGSEGV, on_fatal);
    signal(SIGABRT, on_fatal);
}
