// LD_PRELOAD helper: on SIGSEGV / SIGBUS / SIGABRT prints the faulting address and a backtrace (module+offset per frame;
// resolve with llvm-symbolizer --obj=squarna_amd/libsquarna_hip.so OFFSET), then exits.  The box has no gdb.
//   gcc -O1 -shared -fPIC -o tools/prof/libsegv.so tools/prof/segv.c ;  LD_PRELOAD=tools/prof/libsegv.so python3 ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
static volatile int g_once;
static void on_fault(int sig, siginfo_t *si, void *uc)
{
    (void)uc;
    if (__sync_lock_test_and_set(&g_once, 1)) { for (;;) pause(); }
    char msg[96];
    int n = snprintf(msg, sizeof msg, "\n[segv] signal %d at address %p\n", sig, si ? si->si_addr : 0);
    if (write(2, msg, n) < 0) {}
    void *frames[48];
    const int k = backtrace(frames, 48);
    backtrace_symbols_fd(frames, k, 2);
    _exit(139);
}
__attribute__((constructor)) static void install(void)
{
    void *warm[2]; backtrace(warm, 2);               // loads libgcc now, not inside the handler
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0); sigaction(SIGABRT, &sa, 0);
}
