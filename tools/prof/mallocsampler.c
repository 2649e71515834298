// LD_PRELOAD: counts malloc calls and records the call stack of every 256th one (per thread) while enabled.
// build: gcc -O2 -shared -fPIC -o tools/prof/libmallocsampler.so tools/prof/mallocsampler.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define DEPTH 16
#define MAXS (1 << 20)
static void *(*buf)[DEPTH];
static volatile long nsamp, ncalls;
static volatile int enabled;
static __thread int guard, tick;
extern void *__libc_malloc(size_t);
void *malloc(size_t n)
{
    void *p = __libc_malloc(n);
    if (enabled && !guard) {
        __sync_fetch_and_add(&ncalls, 1);
        if ((++tick & 255) == 0) {
            guard = 1;
            long k = __sync_fetch_and_add(&nsamp, 1);
            if (k < MAXS) {
                void *tmp[DEPTH + 1];
                int c = backtrace(tmp, DEPTH + 1);
                for (int i = 0; i < DEPTH; i++) buf[k][i] = i + 1 < c ? tmp[i + 1] : 0;
            }
            guard = 0;
        }
    }
    return p;
}
void sampler_start(void) { nsamp = 0; ncalls = 0; enabled = 1; }
void sampler_stop(void) { enabled = 0; }
__attribute__((constructor)) static void init(void)
{
    guard = 1;
    buf = calloc(MAXS, sizeof *buf);
    void *w[4]; backtrace(w, 4);
    guard = 0;
    if (!getenv("SAMPLER_MANUAL")) enabled = 1;
}
__attribute__((destructor)) static void fini(void)
{
    enabled = 0; guard = 1;
    const char *path = getenv("SAMPLER_OUT") ? getenv("SAMPLER_OUT") : "malloc.out";
    FILE *f = fopen(path, "w");
    if (!f) return;
    long n = nsamp < MAXS ? nsamp : MAXS;
    fprintf(f, "samples %ld calls %ld\n", n, (long)ncalls);
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
