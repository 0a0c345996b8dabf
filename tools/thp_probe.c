// first-touch cost of fresh anonymous memory: plain 4 KiB pages vs madvise(MADV_HUGEPAGE), 1 and T threads
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
typedef struct { char *p; size_t lo, hi; } W;
static void *touch(void *a) { W *w = a; for (size_t o = w->lo; o < w->hi; o += 4096) w->p[o] = 1; return NULL; }
static double run(size_t bytes, int threads, int huge) {
    char *p = mmap(NULL, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(1); }
    char *q = (char *)(((size_t)p + (2 << 20) - 1) & ~(size_t)((2 << 20) - 1));
    if (huge) if (madvise(q, bytes, MADV_HUGEPAGE)) perror("madvise");
    pthread_t th[64]; W w[64];
    double a = now();
    for (int t = 0; t < threads; ++t) { w[t].p = q; w[t].lo = bytes / threads * t; w[t].hi = bytes / threads * (t + 1); pthread_create(&th[t], NULL, touch, &w[t]); }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    double dt = now() - a;
    munmap(p, bytes + (2 << 20));
    return dt;
}
int main(int argc, char **argv) {
    size_t gb = argc > 1 ? atoi(argv[1]) : 2;
    size_t bytes = gb << 30;
    for (int rep = 0; rep < 2; ++rep)
        for (int huge = 0; huge < 2; ++huge)
            for (int threads = 1; threads <= 16; threads *= 4) {
                double dt = run(bytes, threads, huge);
                printf("{\"gb\": %zu, \"huge\": %d, \"threads\": %d, \"seconds\": %.3f, \"GBps\": %.1f, \"us_per_4k\": %.3f}\n", gb, huge, threads, dt, gb / dt, dt / (bytes / 4096) * 1e6);
                fflush(stdout);
            }
    return 0;
}
