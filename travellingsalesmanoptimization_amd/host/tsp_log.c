/*
 * tsp_log.c -- logging, the stdout contract of the binary, clock and small utilities of the
 * host layer (behaviour of src/utils/errors.c and src/utils/utils.c; new code).
 * The -q output ("Cost: %.2f" / "Time: %.2f") is what scripts/compare_algs.py:72,113 scrapes.
 */
#include "tsp_model.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

static VERBOSITY g_verbosity = NORMAL;

static const char *const k_alg_names[12] = {
    "Greedy", "Greedy Iterative", "2-opt Greedy", "Tabu Search", "VNS", "Cplex No SEC", "Cplex Benders Loop",
    "Extra Mileage", "Cplex Benders Loop with Patching", "Cplex Branch&Cut", "Hard Fixing", "Local Branching"};
static const char *const k_level_names[6] = {"TRACE", "DEBUG", "INFO", "WARN", "ERROR", "FATAL"};
static const char *const k_level_colors[6] = {"\x1b[94m", "\x1b[36m", "\x1b[32m", "\x1b[33m", "\x1b[31m", "\x1b[35m"};

bool err_ok(ERROR_CODE e) { return e == T_OK || e == CANCELLED || e == DEADLINE_EXCEEDED; }
void err_setverbosity(VERBOSITY v) { g_verbosity = v; }
bool err_dolog(void) { return g_verbosity >= VERBOSE; }

/* visible levels: QUIET none; NORMAL warn+; VERBOSE info+; VERY_VERBOSE all (errors.c:47-62) */
static bool level_visible(LOGGING_TYPE level)
{
    switch (g_verbosity) {
    case QUIET: return false;
    case NORMAL: return level >= LOG_WARN;
    case VERBOSE: return level >= LOG_INFO;
    default: return true;
    }
}

void err_logging(LOGGING_TYPE level, const char *file, int line, const char *message, ...)
{
    if (!level_visible(level)) return;
    char stamp[16];
    time_t now = time(NULL);
    stamp[strftime(stamp, sizeof stamp, "%H:%M:%S", localtime(&now))] = '\0';
    fprintf(stderr, "%s %s%-5s\x1b[0m \x1b[90m%s:%d:\x1b[0m ", stamp, k_level_colors[level], k_level_names[level], file, line);
    va_list ap;
    va_start(ap, message);
    vfprintf(stderr, message, ap);
    va_end(ap);
    fputc('\n', stderr);
}

static void rule(void) { puts("--------------------------------------------------------------------------------\n"); }

void err_setinfo(int alg, int nnodes, bool random, char *inputfile, double timelimit, int seed, int tabu_policy,
                 int em, bool init_mip, int bc_policy, bool callback_relaxation, double lb_improv, int lb_delta,
                 bool lb_kstar)
{
    (void)em; (void)init_mip; (void)bc_policy; (void)callback_relaxation; (void)lb_improv; (void)lb_delta; (void)lb_kstar;
    if (g_verbosity == QUIET) return;
    static const char *const pol[4] = {"Fixed", "Dependent on size", "Random", "Linear"};
    puts("Travelling Salesman Problem Solver (MI355X 2-opt engine)");
    rule();
    printf("Algorithm:               %s\n", alg >= 0 && alg < 12 ? k_alg_names[alg] : "?");
    printf("Number of nodes:         %d\n", nnodes);
    printf("Random/File:             %s\n", random ? "random" : (inputfile ? inputfile : "?"));
    if (timelimit != -1.0) printf("Timelimit:               %.2f\n", timelimit); else puts("Timelimit:               not set");
    if (seed != -1) printf("Seed:                    %d\n", seed); else puts("Seed:                    not set");
    if (alg == ALG_TABU_SEARCH) printf("Tenure Policy:           %s\n", pol[tabu_policy & 3]);
    rule();
}

void err_printoutput(double cost, double time, int alg)
{
    if (g_verbosity != QUIET) {
        rule();
        printf("algorithm: %s\n", alg >= 0 && alg < 12 ? k_alg_names[alg] : "?");
        printf("cost: %.2f\n", cost);
        printf("execution time: %.2f seconds\n", time);
        rule();
        puts("Program finished, shutting down...");
    } else if (alg == ALG_CX_BENDERS || alg == ALG_CX_BENDERS_PAT || alg == ALG_CX_BRANCH_AND_CUT) {
        printf("Time: %.2f\n", time);   /* exact methods are compared on time (errors.c:158-160) */
    } else {
        printf("Cost: %.2f\n", cost);
    }
}

void utils_safe_memory_free(void **p)
{
    if (p && *p) { free(*p); *p = NULL; }
}

bool utils_file_exists(const char *filename)
{
    struct stat st;
    return stat(filename, &st) == 0;
}

void utils_startclock(struct timespec *c)
{
    if (clock_gettime(CLOCK_MONOTONIC, c) == -1) log_error("monotonic clock not supported");
}

double utils_timeelapsed(struct timespec *c)
{
    struct timespec now;
    if (clock_gettime(CLOCK_MONOTONIC, &now) == -1) { log_error("monotonic clock not supported"); return -1.0; }
    return (double)(now.tv_sec - c->tv_sec) + (double)(now.tv_nsec - c->tv_nsec) / 1e9;
}

void swap(int *a, int *b) { int t = *a; *a = *b; *b = t; }

/* The program's own rand() stream.  The reference draws from glibc's process-wide generator
 * (seed 1 unless -n seeds it: tsp.c:469); here the HIP runtime shares the process
 * and consumes values from that generator while it initialises, so the host layer keeps the
 * stream in a private TYPE_3 state (initstate(1, .., 128) == glibc's default table) and
 * switches to it only around its own draws.  Same numbers as the reference binary sees. */
static char g_rng_table[128];
static bool g_rng_ready = false;

static char *rng_enter(void)
{
    if (!g_rng_ready) {
        char *prev = initstate(1u, g_rng_table, sizeof g_rng_table); /* leaves g_rng_table current */
        g_rng_ready = true;
        return prev;
    }
    return setstate(g_rng_table);
}

/* Values drawn ahead for the device (mh_VNS's kicks run inside the kernel on numbers handed over in a buffer,
 * tspgpu_vns_search) and not consumed there stay queued: tsp_rand() serves them first, so the program's stream continues
 * exactly where the reference's would. */
static int *g_rng_q = NULL;
static long g_rng_q_head = 0, g_rng_q_len = 0, g_rng_q_cap = 0;

static int rng_draw(void)
{
    char *prev = rng_enter();
    const int r = rand();
    setstate(prev);
    return r;
}

int tsp_rand(void)
{
    if (g_rng_q_head < g_rng_q_len) return g_rng_q[g_rng_q_head++];
    return rng_draw();
}

/* the next `count` values of the stream, without consuming them (NULL if the queue cannot grow) */
const int *tsp_rand_peek(long count)
{
    if (g_rng_q_head > 0) {
        memmove(g_rng_q, g_rng_q + g_rng_q_head, (size_t)(g_rng_q_len - g_rng_q_head) * sizeof(int));
        g_rng_q_len -= g_rng_q_head; g_rng_q_head = 0;
    }
    if (count > g_rng_q_cap) {
        int *q = (int *)realloc(g_rng_q, (size_t)count * sizeof(int));
        if (!q) return NULL;
        g_rng_q = q; g_rng_q_cap = count;
    }
    while (g_rng_q_len < count) g_rng_q[g_rng_q_len++] = rng_draw();
    return g_rng_q;
}

/* the first `count` peeked values have been used */
void tsp_rand_consume(long count)
{
    g_rng_q_head += count;
    if (g_rng_q_head > g_rng_q_len) g_rng_q_head = g_rng_q_len;
}

void tsp_srand(unsigned seed)
{
    char *prev = rng_enter();
    srand(seed);
    setstate(prev);
    g_rng_q_head = g_rng_q_len = 0;      /* values drawn ahead belonged to the old seed */
}

ERROR_CODE tsp_init_solution(int nnodes, tsp_solution *s)
{
    s->path = (int *)calloc((size_t)nnodes, sizeof(int));
    s->comp = (int *)calloc((size_t)nnodes, sizeof(int));
    s->cost = __DBL_MAX__;
    s->ncomp = 0;
    if (!s->path || !s->comp) { log_fatal("solution allocation failed"); return UNAVAILABLE; }
    return T_OK;
}
