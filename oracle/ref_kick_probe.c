/*
 * oracle/ref_kick_probe.c -- what does the REFERENCE's vns_kick (metaheuristic.c:344-409) read at tour[-1] and
 * tour[n] (the unwrapped probes of :372) at large n?  TEST INFRASTRUCTURE ONLY; built by oracle/Makefile into
 * oracle/_ref/ref_kick_probe together with the reference's own sources (compiled where they lie).
 *
 * A process of its own (the answer depends on glibc's heap state, so not a Python process): it replays the allocation
 * history of a `tsp -alg VNS` run in front of the first kick -- the point array, the incumbent (main.c:6), mh_VNS's two
 * solutions (utils.c:137-154), h_greedyutil's `visited` (heuristics.c:230,285) and ref_2opt_once's `prev`
 * (refinment.c:43,89) allocated and freed a few times, in `warm` mode; none of the freed ones in `cold` mode -- and then
 * calls vns_kick `kicks` times on a seeded random cycle with the glibc stream at srand(seed), with a `prev`
 * allocation / free between kicks as the next ref_2opt would do.  calloc is interposed only to REMEMBER the last block
 * of n ints (vns_kick leaks `tour`, :354 -- so it can be inspected afterwards); the allocator itself is glibc's.
 *
 *   ref_kick_probe n kicks seed warm|cold   ->  one JSON line: fnv of the final successor array, the values found
 *                                               at tour[-1] / tour[n] after every kick, whether each block was mmapped
 */
#include <malloc.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsp.h"
#include "algorithms/metaheuristic.h"

extern void *__libc_calloc(size_t, size_t);
static size_t watch_bytes = 0;
static int *last_block = NULL;

void *calloc(size_t nmemb, size_t size)
{
    void *p = __libc_calloc(nmemb, size);
    if (watch_bytes && nmemb * size == watch_bytes) last_block = (int *)p;
    return p;
}

static uint64_t fnv1a(const int *succ, int n)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int i = 0; i < n; i++) { h ^= (unsigned)succ[i]; h *= 0x100000001b3ULL; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s n kicks seed warm|cold\n", argv[0]); return 2; }
    const int n = atoi(argv[1]), kicks = atoi(argv[2]);
    const unsigned seed = (unsigned)atoi(argv[3]);
    const int warm = strcmp(argv[4], "warm") == 0;
    tsp_init();
    err_setverbosity(QUIET);
    tsp_inst.nnodes = n;
    tsp_inst.points = (point *)calloc((size_t)n, sizeof(point));            /* tsp.c:558 */
    tsp_inst.best_solution.path = (int *)calloc((size_t)n, sizeof(int));    /* main.c:6 */
    tsp_solution s, best;
    tsp_init_solution(n, &s);                                               /* metaheuristic.c:256 */
    if (warm) {
        for (int i = 0; i < 3; i++) {                                       /* h_Greedy_iterative: a solution + visited per start */
            tsp_solution t;
            tsp_init_solution(n, &t);
            int *visited = (int *)calloc((size_t)n, sizeof(int));
            free(visited);
            free(t.path);                                                   /* heuristics.c:70 (comp leaks, as there) */
        }
    }
    tsp_init_solution(n, &best);                                            /* :271 */
    /* a seeded random cycle: Fisher-Yates on its own LCG (not rand(): the stream below must be the kicks' alone) */
    int *order = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) order[i] = i;
    uint64_t x = 0x9E3779B97F4A7C15ULL ^ (uint64_t)n;
    for (int i = n - 1; i > 0; i--) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        int j = (int)((x >> 33) % (uint64_t)(i + 1));
        int t = order[i]; order[i] = order[j]; order[j] = t;
    }
    for (int i = 0; i < n; i++) s.path[order[i]] = order[(i + 1) % n];
    free(order);
    const uint64_t fnv0 = fnv1a(s.path, n);
    srand(seed);
    printf("{\"n\": %d, \"kicks\": %d, \"seed\": %u, \"mode\": \"%s\", \"fnv_before\": \"%016llx\", \"probes\": [", n, kicks, seed,
           argv[4], (unsigned long long)fnv0);
    watch_bytes = (size_t)n * sizeof(int);
    for (int k = 0; k < kicks; k++) {
        if (warm) { int *prev = (int *)calloc((size_t)n, sizeof(int)); free(prev); }   /* ref_2opt_once, refinment.c:43,89 */
        last_block = NULL;
        vns_kick(&s);
        /* vns_kick's allocations of n ints, in order: prev (freed at :405), tour (leaked): the last one is tour */
        const int *tour = last_block;
        const size_t head = ((const size_t *)tour)[-1];
        printf("%s{\"before\": %d, \"after\": %d, \"mmapped\": %d}", k ? ", " : "", tour[-1], tour[n], (int)((head & 2) != 0));
    }
    printf("], \"fnv\": \"%016llx\"}\n", (unsigned long long)fnv1a(s.path, n));
    return 0;
}
