// Is the core ours?  A latency-bound loop (one dependent chain of adds) and a throughput-bound one (eight independent
// chains) on one thread: the first gives the clock (1 add per cycle), the second the adds per cycle the thread gets --
// a Zen 5 core issues six; a thread whose SMT sibling is busy with somebody else's work gets about half.
//   gcc -O2 -o /tmp/host_ipc tools/ubench/host_ipc.c && /tmp/host_ipc
#include <stdint.h>
#include <stdio.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(void) {
    const long N = 400000000L;
    uint64_t a = 1, b = 2, c = 3, d = 4, e = 5, f = 6, g = 7, h = 8;
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        for (long i = 0; i < N; i++) __asm__ volatile("add $1, %0" : "+r"(a));
        double t1 = now();
        for (long i = 0; i < N / 8; i++)
            __asm__ volatile("add $1, %0\n add $1, %1\n add $1, %2\n add $1, %3\n add $1, %4\n add $1, %5\n add $1, %6\n add $1, %7\n"
                             "add $1, %0\n add $1, %1\n add $1, %2\n add $1, %3\n add $1, %4\n add $1, %5\n add $1, %6\n add $1, %7\n"
                             "add $1, %0\n add $1, %1\n add $1, %2\n add $1, %3\n add $1, %4\n add $1, %5\n add $1, %6\n add $1, %7\n"
                             "add $1, %0\n add $1, %1\n add $1, %2\n add $1, %3\n add $1, %4\n add $1, %5\n add $1, %6\n add $1, %7\n"
                             : "+r"(a), "+r"(b), "+r"(c), "+r"(d), "+r"(e), "+r"(f), "+r"(g), "+r"(h));
        double t2 = now();
        const double ghz = N / (t1 - t0) * 1e-9; // dependent adds per ns = clock (the loop's own add/cmp/jmp hide behind it)
        printf("dependent chain: %.2f adds/ns (= GHz); 8 chains: %.2f adds/ns = %.2f per cycle\n", ghz, 4.0 * N / (t2 - t1) * 1e-9,
               4.0 * N / (t2 - t1) * 1e-9 / ghz);
    }
    return (int)(a + b + c + d + e + f + g + h) & 0;
}
