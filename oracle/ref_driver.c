/*
 * oracle/ref_driver.c -- thin ctypes-friendly driver around the REFERENCE's own
 * heuristic sources, which oracle/Makefile compiles where they lie under
 * /root/reference/src into oracle/_ref/libtspref.so (never copied into this
 * repository, never shipped in git: oracle/_ref/ is git-ignored).
 *
 * TEST INFRASTRUCTURE ONLY.  Purpose: (1) generate the golden vectors in
 * tests/golden/ (oracle/make_golden.py), (2) validate oracle/cpu_ref.c,
 * (3) optionally serve as bench.py's cpu_baseline of kind "reference".
 *
 * It mirrors what src/main.c:158-192 does around the algorithms, minus the
 * gnuplot output (gnuplot is not installed; SIGPIPE is ignored so that the
 * reference's plot_* calls inside mh_TabuSearch / mh_VNS are harmless).
 */
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "tsp.h"
#include "algorithms/heuristics.h"
#include "algorithms/metaheuristic.h"
#include "algorithms/refinment.h"

static int drv_ready = 0;

static void drop_instance(void)
{
    if (!drv_ready) return;
    free(tsp_inst.points); tsp_inst.points = NULL;
    free(tsp_inst.costs); tsp_inst.costs = NULL;
    free(tsp_inst.best_solution.path); tsp_inst.best_solution.path = NULL;
}

int refdrv_init(const char *scratch_dir)
{
    signal(SIGPIPE, SIG_IGN);
    if (scratch_dir && *scratch_dir) {
        /* mh_TabuSearch / mh_VNS fopen("results/...") relative to the cwd
         * (metaheuristic.c:97,275): the Python wrapper chdirs here around them */
        char buf[4096];
        mkdir(scratch_dir, 0777);
        snprintf(buf, sizeof buf, "%s/results", scratch_dir);
        mkdir(buf, 0777);
    }
    drop_instance();
    tsp_init();
    err_setverbosity(QUIET);
    drv_ready = 1;
    return 0;
}

static void fresh_incumbent(void)
{
    free(tsp_inst.best_solution.path);
    tsp_inst.best_solution.path = (int *)calloc(tsp_inst.nnodes, sizeof(int)); /* main.c:6 */
    tsp_inst.best_solution.cost = __DBL_MAX__;
    tsp_inst.starting_node = 0;
    tsp_env.timelimit = -1;
    utils_startclock(&tsp_inst.c);
}

int refdrv_set_points(const double *xy, int n)
{
    drop_instance();
    tsp_inst.nnodes = n;
    tsp_inst.points = (point *)calloc(n, sizeof(point));
    for (int i = 0; i < n; i++) { tsp_inst.points[i].x = xy[2 * i]; tsp_inst.points[i].y = xy[2 * i + 1]; }
    tsp_compute_costs();
    fresh_incumbent();
    return 0;
}

int refdrv_random(int n, int seed)
{
    drop_instance();
    tsp_inst.nnodes = n;
    tsp_env.seed = seed;
    tsp_generate_randompoints();
    fresh_incumbent();
    return 0;
}

int refdrv_read_file(const char *path)
{
    drop_instance();
    free(tsp_env.inputfile);
    tsp_env.inputfile = strdup(path);
    tsp_read_input();   /* exits the process on ATT / CEIL_2D, tsp.c:576-584 */
    fresh_incumbent();
    return 0;
}

int refdrv_n(void) { return tsp_inst.nnodes; }
const double *refdrv_costs(void) { return tsp_inst.costs; }
void refdrv_points(double *xy)
{
    for (int i = 0; i < tsp_inst.nnodes; i++) { xy[2 * i] = tsp_inst.points[i].x; xy[2 * i + 1] = tsp_inst.points[i].y; }
}

int refdrv_nn(int start, int *path, double *cost)
{
    tsp_solution s = { 0, path, 0, NULL };
    int e = h_greedyutil(start, &s, tsp_inst.costs);
    *cost = s.cost;
    return e;
}

double refdrv_two_opt_once(int *path, double *cost)
{
    tsp_solution s = { *cost, path, 0, NULL };
    double d = ref_2opt_once(&s, tsp_inst.costs);
    *cost = s.cost;
    return d;
}

/* the do/while of ref_2opt (refinment.c:15-27) unrolled here so that sweeps can
 * be counted and the first `ntrace` post-sweep costs recorded */
long refdrv_two_opt_counted(int *path, double *cost, long max_sweeps, double *trace, int ntrace)
{
    tsp_solution s = { 0, path, 0, NULL };
    for (int i = 0; i < tsp_inst.nnodes; i++) s.cost += tsp_inst.costs[i * tsp_inst.nnodes + path[i]];
    long sweeps = 0;
    double d;
    do {
        if (max_sweeps >= 0 && sweeps >= max_sweeps) break;
        d = ref_2opt_once(&s, tsp_inst.costs);
        if (sweeps < ntrace) trace[sweeps] = s.cost;
        sweeps++;
    } while (d < EPSILON);
    *cost = s.cost;
    return sweeps;
}

/* ref_2opt itself on a caller-supplied matrix (NULL = the instance's) */
int refdrv_ref_2opt(int *path, double *cost, double *costs)
{
    tsp_solution s = { *cost, path, 0, NULL };
    int e = ref_2opt(&s, costs ? costs : tsp_inst.costs, false);
    *cost = s.cost;
    return e;
}

int refdrv_tabu_move(int *path, double *cost, int *tabu_list, int tenure, int iter)
{
    tabu_search ts;
    memset(&ts, 0, sizeof ts);
    ts.tabu_list = tabu_list;
    ts.tenure = tenure;
    return tabu_best_move(path, cost, &ts, iter);
}

int refdrv_vns_kick(int *path)
{
    tsp_solution s = { 0, path, 0, NULL };
    return vns_kick(&s);
}

void refdrv_srand(unsigned seed) { srand(seed); }

int refdrv_mod_costs(double *costs, int *path, double *cost)
{
    tsp_solution s = { 0, path, 0, NULL };
    utils_startclock(&tsp_inst.c);
    int e = h_Greedy_2opt_mod_costs(&s, costs);
    *cost = s.cost;
    return e;
}

/* alg: 0 GREEDY, 1 GREEDY_ITER, 2 2OPT_GREEDY, 3 TABU_SEARCH, 4 VNS
 * (tsp.h:40-53, dispatch main.c:8-80) */
int refdrv_run(int alg, int k, double *cost, int *path, int *starting_node)
{
    int e = -1;
    fresh_incumbent();
    tsp_inst.alg = (algorithms)alg;
    tsp_env.k = k;
    switch (alg) {
    case 0: e = h_Greedy(); break;
    case 1: e = h_Greedy_iterative(); break;
    case 2: e = h_greedy_2opt(); break;
    case 3: e = mh_TabuSearch(); break;
    case 4: e = mh_VNS(); break;
    default: return -1;
    }
    *cost = tsp_inst.best_solution.cost;
    memcpy(path, tsp_inst.best_solution.path, sizeof(int) * tsp_inst.nnodes);
    *starting_node = tsp_inst.starting_node;
    return e;
}
