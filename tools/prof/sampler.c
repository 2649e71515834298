// LD_PRELOAD sampler: every thread gets a CPU-time timer (SIGPROF to that thread every 0.25 ms of ITS cpu time), the
// handler stores 16 return addresses; dumped at exit with /proc/self/maps for tools/prof/symbolize.py.
// build: gcc -O2 -shared -fPIC -o tools/prof/libsampler.so tools/prof/sampler.c -ldl -lpthread -lrt
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
#define DEPTH 16
#define MAXS (1 << 21)
static void *(*buf)[DEPTH];
static volatile long nsamp;
static volatile int enabled;
static void on_prof(int sig)
{
    if (!enabled) return;
    long k = __sync_fetch_and_add(&nsamp, 1);
    if (k >= MAXS) return;
    void *tmp[DEPTH + 2];
    int n = backtrace(tmp, DEPTH + 2);
    for (int i = 0; i < DEPTH; i++) buf[k][i] = i + 2 < n ? tmp[i + 2] : 0;
}
static void arm_thread(void)
{
    struct sigevent sev; memset(&sev, 0, sizeof sev);
    sev.sigev_notify = SIGEV_THREAD_ID; sev.sigev_signo = SIGPROF;
    sev._sigev_un._tid = (pid_t)syscall(SYS_gettid);
    timer_t t;
    if (timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &t)) return;
    struct itimerspec its = {{0, 250000}, {0, 250000}};
    timer_settime(t, 0, &its, 0);
}
struct tramp { void *(*fn)(void *); void *arg; };
static void *trampoline(void *p)
{
    struct tramp t = *(struct tramp *)p; free(p);
    arm_thread();
    return t.fn(t.arg);
}
int pthread_create(pthread_t *th, const pthread_attr_t *attr, void *(*fn)(void *), void *arg)
{
    static int (*real)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
    if (!real) real = dlsym(RTLD_NEXT, "pthread_create");
    struct tramp *t = malloc(sizeof *t); t->fn = fn; t->arg = arg;
    return real(th, attr, trampoline, t);
}
void sampler_start(void) { nsamp = 0; enabled = 1; }
void sampler_stop(void) { enabled = 0; }
__attribute__((constructor)) static void init(void)
{
    buf = calloc(MAXS, sizeof *buf);
    void *w[4]; backtrace(w, 4);                      // (loads libgcc now, not inside the handler)
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_prof; sa.sa_flags = SA_RESTART;
    sigaction(SIGPROF, &sa, 0);
    arm_thread();
    if (!getenv("SAMPLER_MANUAL")) enabled = 1;
}
__attribute__((destructor)) static void fini(void)
{
    enabled = 0;
    const char *path = getenv("SAMPLER_OUT") ? getenv("SAMPLER_OUT") : "sampler.out";
    FILE *f = fopen(path, "w");
    if (!f) return;
    long n = nsamp < MAXS ? nsamp : MAXS;
    fprintf(f, "samples %ld\n", n);
    FILE *m = fopen("/proc/self/maps", "r");
    char line[512];
    while (m && fgets(line, sizeof line, m)) if (strstr(line, "r-xp")) fprintf(f, "map %s", line);
    if (m) fclose(m);
    for (long k = 0; k < n; k++) {
        fprintf(f, "s");
        for (int i = 0; i < DEPTH && buf[k][i]; i++) fprintf(f, " %p", buf[k][i]);
        fprintf(f, "\n");
    }
    fclose(f);
}
