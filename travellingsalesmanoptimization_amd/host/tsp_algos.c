/*
 * tsp_algos.c -- the heuristic path of the reference (src/algorithms/refinment.c,
 * heuristics.c, metaheuristic.c) with its own signatures, running on the MI355X through
 * include/tspgpu.h.  Control flow, incumbent rules, error codes and deadline semantics follow
 * the reference (cited per function); the O(n^2) loops are device kernels.
 *
 * Which matrix a call runs on: `costs == tsp_inst.costs` selects the matrix that
 * tsp_compute_costs left resident on the device; any other pointer (the xstar-weighted
 * matrix of cplex_model.c:1176-1258) is uploaded into a context private to the calling
 * thread, so concurrent CPLEX callback threads never share device state.
 */
#include "tsp_model.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "tspgpu.h"

/* Per-thread contexts for caller-supplied matrices.  A context holds a stream, an n x n matrix and tour slots
 * on the device, so its lifetime is managed: a pthread key destructor destroys it when its thread exits (CPLEX
 * tears its worker threads down after the solve), and whatever is still alive when the instance is freed goes in
 * tsp_gpu_release_threads().  The registry (mutex-protected list) is what makes the second path possible. */
#include <pthread.h>

typedef struct thread_ctx {
    tspgpu_ctx *gpu;
    struct thread_ctx *next;
    pthread_mutex_t use;        /* held by the owning thread for the duration of a device call, and by whoever destroys
                                   the context: tsp_gpu_release_threads() waits for a call in flight instead of freeing
                                   the context under it */
} thread_ctx;

static pthread_mutex_t reg_lock = PTHREAD_MUTEX_INITIALIZER;
static thread_ctx *reg_head = NULL;
static int reg_live = 0;
static pthread_key_t reg_key;
static pthread_once_t reg_once = PTHREAD_ONCE_INIT;

static void reg_unlink(thread_ctx *t)
{
    for (thread_ctx **pp = &reg_head; *pp; pp = &(*pp)->next)
        if (*pp == t) { *pp = t->next; reg_live--; break; }
}

static void thread_ctx_exit(void *arg)          /* runs in the exiting thread */
{
    thread_ctx *t = (thread_ctx *)arg;
    pthread_mutex_lock(&reg_lock);              /* (tsp_gpu_release_threads walks the list under this lock) */
    reg_unlink(t);
    pthread_mutex_unlock(&reg_lock);
    pthread_mutex_lock(&t->use);
    tspgpu_destroy(t->gpu);
    t->gpu = NULL;
    pthread_mutex_unlock(&t->use);
    pthread_mutex_destroy(&t->use);
    free(t);
}

static void reg_make_key(void) { pthread_key_create(&reg_key, thread_ctx_exit); }

int tsp_gpu_thread_contexts(void)
{
    pthread_mutex_lock(&reg_lock);
    const int live = reg_live;
    pthread_mutex_unlock(&reg_lock);
    return live;
}

/* destroys the contexts of threads that are still alive (waiting for a device call that one of them may be in); they
 * re-create theirs on the next call */
void tsp_gpu_release_threads(void)
{
    pthread_mutex_lock(&reg_lock);
    for (thread_ctx *t = reg_head; t; t = t->next) {
        pthread_mutex_lock(&t->use);
        if (t->gpu) tspgpu_destroy(t->gpu);
        t->gpu = NULL;
        pthread_mutex_unlock(&t->use);
    }
    pthread_mutex_unlock(&reg_lock);
}

/* the calling thread's context, LOCKED (thread_done() when the device call is over); NULL without a device */
static thread_ctx *thread_lease(void)
{
    pthread_once(&reg_once, reg_make_key);
    thread_ctx *t = (thread_ctx *)pthread_getspecific(reg_key);
    if (!t) {
        t = (thread_ctx *)calloc(1, sizeof *t);
        if (!t) return NULL;
        pthread_mutex_init(&t->use, NULL);
        pthread_setspecific(reg_key, t);
        pthread_mutex_lock(&reg_lock);
        t->next = reg_head; reg_head = t; reg_live++;
        pthread_mutex_unlock(&reg_lock);
    }
    pthread_mutex_lock(&t->use);
    if (!t->gpu) {
        const char *dev = getenv("TSP_GPU_DEVICE");
        if (tspgpu_create(dev ? atoi(dev) : 0, &t->gpu) != 0) { t->gpu = NULL; pthread_mutex_unlock(&t->use); return NULL; }
    }
    return t;
}

static void thread_done(thread_ctx *t) { if (t) pthread_mutex_unlock(&t->use); }

static double time_left(void)
{
    if (tsp_env.timelimit == -1.0) return -1.0;
    const double left = tsp_env.timelimit - utils_timeelapsed(&tsp_inst.c);
    return left > 0 ? left : 0.0;
}

static bool past_deadline(void)
{
    return tsp_env.timelimit != -1.0 && utils_timeelapsed(&tsp_inst.c) > tsp_env.timelimit;
}

/* Context holding `costs`.  The instance's own matrix lives in the process-wide context; any other pointer is a
 * caller matrix and is uploaded into the calling thread's context on EVERY call: the pointer says nothing about
 * the contents (cplex_model.c:1176-1258 refills one buffer per callback), so nothing is cached across calls. */
static tspgpu_ctx *ctx_for(double *costs, thread_ctx **lease)
{
    *lease = NULL;
    if (costs == tsp_inst.costs || (!costs && tsp_lazy_costs)) return tsp_gpu();
    thread_ctx *t = thread_lease();
    if (!t) return NULL;
    if (tspgpu_set_costs(t->gpu, costs, tsp_inst.nnodes) != 0) {
        log_error("tspgpu_set_costs: %s", tspgpu_last_error(t->gpu));
        thread_done(t);
        return NULL;
    }
    *lease = t;                                 /* held until thread_done(): the context cannot be destroyed under the call */
    return t->gpu;
}

static ERROR_CODE from_rc(int rc) { return (ERROR_CODE)rc; }

/* TSP_GPU_STATS=1: one JSON line per multi-start call on stderr (what bench.py's C-path leg parses) */
static void multi_stats(const char *what, struct tspgpu_multi *m, int starts, long sweeps, double seconds, double cost)
{
    const char *on = getenv("TSP_GPU_STATS");
    if (!on || !atoi(on)) return;
    static const char *kinds[] = {"none", "host", "rccl"};
    const int kind = m ? (int)tspgpu_multi_info(m, 2) : 0;
    fprintf(stderr, "tspgpu-stats: {\"call\": \"%s\", \"devices\": %d, \"exchange\": \"%s\", \"starts\": %d, \"sweeps\": %ld, "
                    "\"seconds\": %.6f, \"solve_s\": %.6f, \"exchange_s\": %.6f, \"rccl_init_s\": %.6f, \"best_cost\": %.2f}\n",
            what, m ? tspgpu_multi_devices(m) : 1, kinds[kind < 0 || kind > 2 ? 0 : kind], starts, sweeps, seconds,
            m ? tspgpu_multi_info(m, 5) : seconds, m ? tspgpu_multi_info(m, 4) : 0.0, m ? tspgpu_multi_info(m, 3) : 0.0, cost);
}

/* ===================================================================== refinment.c */

/* refinment.c:3-37.  Cost recompute, sweep loop and stop rule run on the device
 * (tspgpu_two_opt); the deadline is polled once per batch of sweeps instead of per sweep. */
ERROR_CODE ref_2opt(tsp_solution *solution, double *costs, bool update_incumbent)
{
    thread_ctx *lease;
    tspgpu_ctx *g = ctx_for(costs, &lease);
    if (!g) return UNAVAILABLE;
    ERROR_CODE e = T_OK;
    if (past_deadline()) {
        /* the reference recomputes the cost and leaves at the first poll */
        double c = 0;
        for (int i = 0; i < tsp_inst.nnodes; i++)
            c += costs ? costs[(size_t)i * tsp_inst.nnodes + solution->path[i]] : tsp_get_cost(i, solution->path[i]);   /* (a lazy host copy: tsp_get_cost) */
        solution->cost = c;
        log_debug("time limit exceeded in 2opt");
        e = DEADLINE_EXCEEDED;
    } else {
        int rc = tspgpu_two_opt(g, solution->path, &solution->cost, time_left(), NULL);
        if (rc != 0 && rc != DEADLINE_EXCEEDED) {
            log_error("tspgpu_two_opt: %s", tspgpu_last_error(g));
            thread_done(lease);
            return from_rc(rc);
        }
        e = from_rc(rc);
    }
    thread_done(lease);
    if (update_incumbent) {
        ERROR_CODE u = tsp_update_best_solution(solution);
        if (!err_ok(u)) log_error("code %d : Error in 2opt solution update", u);
    }
    return e;
}

/* refinment.c:39-93.  The return value is the best delta and has no room for an error: a device failure here is fatal
 * (log_fatal + tsp_handlefatal, what the reference does with errors it cannot return, tsp.c:738-742) -- returning 0
 * would make a caller that loops on the delta stop as if it had reached a local optimum. */
double ref_2opt_once(tsp_solution *solution, double *costs)
{
    thread_ctx *lease;
    tspgpu_ctx *g = ctx_for(costs, &lease);
    double delta = 0;
    if (!g || tspgpu_two_opt_once(g, solution->path, &solution->cost, &delta) != 0) {
        log_fatal("tspgpu_two_opt_once failed: %s", g ? tspgpu_last_error(g) : "no device");
        thread_done(lease);
        tsp_handlefatal();
    }
    thread_done(lease);
    return delta;
}

/* refinment.c:95-114: host utility kept for callers that own a prev[] array (vns / 3-opt moves) */
void ref_reverse_path(int a, int succ_a, int b, int succ_b, int *prev, int *path)
{
    path[a] = b;
    path[succ_a] = succ_b;
    for (int v = b; ; ) {
        const int back = prev[v];
        path[v] = back;
        if (back == succ_a) break;
        v = back;
    }
    for (int k = 0; k < tsp_inst.nnodes; k++) prev[path[k]] = k;
}

/* ===================================================================== heuristics.c */

/* heuristics.c:216-288 */
ERROR_CODE h_greedyutil(int starting_node, tsp_solution *solution, double *costs)
{
    if (!costs && !tsp_matrix_free && !tsp_lazy_costs) { log_error("matrix of costs not found"); return INTERNAL; }
    if (starting_node >= tsp_inst.nnodes || starting_node < 0) { log_error("starting node not correct"); return UNAVAILABLE; }
    if (past_deadline()) { log_warn("time limit exceeded in greedy util"); return DEADLINE_EXCEEDED; }
    thread_ctx *lease;
    tspgpu_ctx *g = ctx_for(costs, &lease);
    if (!g) return UNAVAILABLE;
    int rc = tspgpu_nn_tour(g, starting_node, solution->path, &solution->cost);
    if (rc) log_error("tspgpu_nn_tour: %s", tspgpu_last_error(g));
    thread_done(lease);
    return from_rc(rc);
}

/* heuristics.c:12-32 */
ERROR_CODE h_Greedy(void)
{
    log_info("running Nearest Neighbour");
    tsp_solution s;
    tsp_init_solution(tsp_inst.nnodes, &s);
    ERROR_CODE e = h_greedyutil(tsp_inst.starting_node, &s, tsp_inst.costs);
    if (!err_ok(e)) log_error("code %d : greedy did not finish correctly", e);
    else {
        e = tsp_update_best_solution(&s);
        if (!err_ok(e)) log_error("code %d : error in updating solution for greedy", e);
    }
    free(s.path); free(s.comp);
    return e;
}

/* heuristics.c:34-72: every start on the device in one batch; the first strictly best wins */
ERROR_CODE h_Greedy_iterative(void)
{
    log_info("running All Nearest Neighbour");
    if (past_deadline()) return DEADLINE_EXCEEDED;
    tspgpu_ctx *g = tsp_gpu();
    if (!g) return UNAVAILABLE;
    struct tspgpu_multi *m = tsp_gpu_multi();
    tsp_solution s;
    tsp_init_solution(tsp_inst.nnodes, &s);
    int start = -1, done = 0;
    const double t0 = utils_timeelapsed(&tsp_inst.c);
    int rc = m ? tspgpu_multi_nn_all(m, NULL, tsp_inst.nnodes, time_left(), s.path, &s.cost, &start, &done)
               : tspgpu_nn_all_timed(g, NULL, tsp_inst.nnodes, time_left(), s.path, &s.cost, &start, &done);
    multi_stats("h_Greedy_iterative", m, tsp_inst.nnodes, 0, utils_timeelapsed(&tsp_inst.c) - t0, s.cost);
    ERROR_CODE e = from_rc(rc);
    if (rc && rc != DEADLINE_EXCEEDED) log_error("tspgpu_nn_all: %s", m ? tspgpu_multi_last_error(m) : tspgpu_last_error(g));
    else if (start >= 0 && s.cost < tsp_inst.best_solution.cost) {
        log_info("found new best, node %d", start);
        tsp_inst.starting_node = start;
        ERROR_CODE u = tsp_update_best_solution(&s);
        if (!err_ok(u)) log_error("code %d : error in updating best solution of greedy iterative", u);
    }
    free(s.path); free(s.comp);
    return e;
}

/* heuristics.c:74-116: NN + 2-opt from every start, batched on the device.  Without a time
 * limit the result equals the sequential loop exactly (minimum over the same set, ties to the
 * lowest start) and starting_node ends as the LAST start processed (:105-109). */
ERROR_CODE h_greedy_2opt(void)
{
    log_info("running All Nearest Neighbour + 2OPT");
    if (past_deadline()) { log_warn("time limit exceeded in greedy 2opt"); return DEADLINE_EXCEEDED; }
    tspgpu_ctx *g = tsp_gpu();
    if (!g) return UNAVAILABLE;
    tsp_solution s;
    tsp_init_solution(tsp_inst.nnodes, &s);
    int start = -1;
    long sweeps = 0;
    /* several devices (TSP_GPU_DEVICES): start i on device i mod G, one RCCL MIN all-reduce + one broadcast */
    struct tspgpu_multi *m = tsp_gpu_multi();
    const double t0 = utils_timeelapsed(&tsp_inst.c);
    int rc = m ? tspgpu_multi_multistart_nn_2opt(m, NULL, tsp_inst.nnodes, time_left(), s.path, &s.cost, &start, &sweeps)
               : tspgpu_multistart_nn_2opt(g, NULL, tsp_inst.nnodes, time_left(), s.path, &s.cost, &start, &sweeps, NULL, NULL);
    multi_stats("h_greedy_2opt", m, tsp_inst.nnodes, sweeps, utils_timeelapsed(&tsp_inst.c) - t0, s.cost);
    ERROR_CODE e = from_rc(rc);
    if (rc != 0 && rc != DEADLINE_EXCEEDED) {
        log_error("tspgpu_multistart_nn_2opt: %s", m ? tspgpu_multi_last_error(m) : tspgpu_last_error(g));
    } else {
        ERROR_CODE u = tsp_update_best_solution(&s);
        if (!err_ok(u)) log_error("code %d : Error in 2opt solution update", u);
        if (rc == 0) tsp_inst.starting_node = tsp_inst.nnodes - 1;
        else log_warn("time limit exceeded in greedy 2opt");
        log_debug("best start %d, cost %f, %ld sweeps", start, s.cost, sweeps);
    }
    free(s.path); free(s.comp);
    return e;
}

/* heuristics.c:118-149: caller's matrix and solution, no incumbent update; what is left in
 * *solution is the tour of the LAST start, as in the reference's loop */
ERROR_CODE h_Greedy_2opt_mod_costs(tsp_solution *solution, double *costs)
{
    if (past_deadline()) return DEADLINE_EXCEEDED;
    thread_ctx *lease;
    tspgpu_ctx *g = ctx_for(costs, &lease);
    if (!g) return UNAVAILABLE;
    const int n = tsp_inst.nnodes;
    int *best = (int *)malloc((size_t)n * sizeof(int));
    double best_cost = 0;
    int start = -1;
    int rc = tspgpu_multistart_nn_2opt(g, NULL, n, time_left(), best, &best_cost, &start, NULL, solution->path, &solution->cost);
    free(best);
    if (rc != 0 && rc != DEADLINE_EXCEEDED) log_error("tspgpu_multistart_nn_2opt: %s", tspgpu_last_error(g));
    thread_done(lease);
    return from_rc(rc);
}

ERROR_CODE h_ExtraMileage(void)
{
    log_error("Extra Mileage is not part of the accelerated heuristic path");
    return UNIMPLEMENTED;
}

/* ===================================================================== metaheuristic.c */

ERROR_CODE tabu_fixed_policy(tabu_search *t, int value)
{
    if (tsp_env.policy != POL_FIXED) { log_warn("policy has already been set"); return ALREADY_EXISTS; }
    t->tenure = value;
    return T_OK;
}

ERROR_CODE tabu_dependent_policy(tabu_search *t)
{
    if (tsp_env.policy != POL_SIZE) { log_warn("policy has already been set"); return ALREADY_EXISTS; }
    t->tenure = (int)ceil((t->max_tenure + t->min_tenure) / 2);
    return T_OK;
}

ERROR_CODE tabu_random_policy(tabu_search *t)
{
    if (tsp_env.policy != POL_RANDOM) { log_warn("policy has already been set"); return ALREADY_EXISTS; }
    t->tenure = (int)(tsp_rand() / RAND_MAX) * (t->max_tenure - t->min_tenure) + t->min_tenure;
    return T_OK;
}

ERROR_CODE tabu_linear_policy(tabu_search *ts)
{
    if (tsp_env.policy != POL_LINEAR) { log_warn("policy has already been set"); return ALREADY_EXISTS; }
    if (ts->tenure == ts->max_tenure || ts->tenure == ts->min_tenure) ts->increment = !ts->increment;
    ts->tenure += ts->increment ? 1 : -1;
    return T_OK;
}

ERROR_CODE tabu_init(tabu_search *ts, int nnodes)
{
    log_info("running Tabu Search");
    ts->tabu_list = (int *)malloc((size_t)nnodes * sizeof(int));
    for (int i = 0; i < nnodes; i++) ts->tabu_list[i] = -1;
    ts->increment = true;
    ts->tenure = MIN_FRACTION * nnodes + 1;
    ts->max_tenure = MAX_FRACTION * nnodes;
    ts->min_tenure = MIN_FRACTION * nnodes;
    return T_OK;
}

bool is_in_tabu_list(tabu_search *ts, int node, int it)
{
    return it - ts->tabu_list[node] < ts->tenure && ts->tabu_list[node] != -1;
}

void tabu_free(tabu_search *ts) { utils_safe_free(ts->tabu_list); }

/* metaheuristic.c:188-245: one tabu move on the device, host arrays in and out */
ERROR_CODE tabu_best_move(int *path, double *cost, tabu_search *ts, int it)
{
    tspgpu_ctx *g = tsp_gpu();
    if (!g) return UNAVAILABLE;
    int rc = tspgpu_tabu_move(g, path, cost, ts->tabu_list, ts->tenure, it);
    if (rc) log_error("tspgpu_tabu_move: %s", tspgpu_last_error(g));
    return from_rc(rc);
}

/* metaheuristic.c:86-186.  With the only reachable policy (linear: tsp.c:15, no flag selects
 * another) the whole k-iteration walk runs resident on the device; other policies step
 * through tabu_best_move.  As in the reference the loop does not stop at the deadline
 * (:118-123): it needs an explicit -k. */
ERROR_CODE mh_TabuSearch(void)
{
    tabu_search ts;
    if (!err_ok(tabu_init(&ts, tsp_inst.nnodes))) { log_fatal("Error in init tabu search"); tsp_handlefatal(); }
    ERROR_CODE e = h_greedy_2opt();
    if (!err_ok(e)) { log_fatal("code %d : Error in greedy solution computation", e); tsp_handlefatal(); }

    const int n = tsp_inst.nnodes;
    tsp_solution s;
    tsp_init_solution(n, &s);
    s.cost = tsp_inst.best_solution.cost;
    memcpy(s.path, tsp_inst.best_solution.path, (size_t)n * sizeof(int));

    const int k = tsp_env.k;
    if (k > 50000000) { log_error("Tabu Search needs an explicit -k (at most 5e7 iterations)"); free(s.path); free(s.comp); tabu_free(&ts); return INVALID_ARGUMENT; }
    FILE *f = fopen("results/TabuResults.dat", "w+");
    e = past_deadline() ? DEADLINE_EXCEEDED : T_OK;

    if (tsp_env.policy == POL_LINEAR) {
        tspgpu_ctx *g = tsp_gpu();
        tsp_solution best;
        tsp_init_solution(n, &best);
        double *trace = (double *)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
        int rc = tspgpu_tabu_search(g, s.path, &s.cost, k, best.path, &best.cost, trace);
        if (rc) { log_fatal("code %d : Error in tabu best move: %s", rc, tspgpu_last_error(g)); tsp_handlefatal(); }
        if (f) for (int i = 0; i < k; i++) fprintf(f, "%d,%f\n", i, trace[i]);
        ERROR_CODE u = tsp_update_best_solution(&best);
        if (!err_ok(u)) { log_fatal("code %d : Error in updating best solution", u); tsp_handlefatal(); }
        free(trace); free(best.path); free(best.comp);
    } else {
        for (int it = 0; it < k; it++) {
            ERROR_CODE p = tsp_env.policy == POL_FIXED ? tabu_fixed_policy(&ts, 30)
                         : tsp_env.policy == POL_RANDOM ? tabu_random_policy(&ts) : tabu_dependent_policy(&ts);
            if (!err_ok(p)) log_warn("using already set policy %d", tsp_env.policy);
            ERROR_CODE m = tabu_best_move(s.path, &s.cost, &ts, it);
            if (!err_ok(m)) { log_fatal("code %d : Error in tabu best move", m); tsp_handlefatal(); }
            ERROR_CODE u = tsp_update_best_solution(&s);
            if (!err_ok(u)) { log_fatal("code %d : Error in updating best solution", u); tsp_handlefatal(); }
            if (f) fprintf(f, "%d,%f\n", it, s.cost);
        }
    }
    if (f) fclose(f);
    free(s.path); free(s.comp);
    tabu_free(&ts);
    return e;
}

/* metaheuristic.c:344-409 + :490-500 (case 7).  Three tour positions from rand(); the
 * reference compares candidates with tour[idx-1] / tour[idx+1] without wrapping (:372).  On
 * glibc those out-of-range ints read 0 (chunk-size high word / calloc padding) except the
 * upper one when 4n+8 is a multiple of 16; the guards below hold those values so that the
 * rand() stream is consumed exactly as by the reference binary. */
ERROR_CODE vns_kick(tsp_solution *solution)
{
    log_debug("KICK");
    const int n = tsp_inst.nnodes;
    int *store = (int *)malloc((size_t)(n + 2) * sizeof(int));
    int *tour = store + 1;
    tour[-1] = 0;
    tour[n] = (n % 4 == 2) ? -2 : 0;
    for (int p = 0, v = 0; p < n; p++, v = solution->path[v]) tour[p] = v;

    int pick[3];
    for (int i = 0; i < 3; i++) {
        int r;
        do {
            r = tsp_rand() % n;
            for (int j = 0; j < i; j++)
                if (r == pick[j] || r == tour[pick[j] - 1] || r == tour[pick[j] + 1]) { r = -1; break; }
        } while (r == -1);
        pick[i] = r;
        for (int j = i; j > 0 && pick[j] < pick[j - 1]; j--) swap(&pick[j], &pick[j - 1]);
    }
    const int A = tour[pick[0]], sA = tour[(pick[0] + 1) % n];
    const int B = tour[pick[1]], sB = tour[(pick[1] + 1) % n];
    const int C = tour[pick[2]], sC = tour[(pick[2] + 1) % n];
    free(store);
    return tabu_make_move(NULL, solution, 7, A, sA, B, sB, C, sC);
}

/* metaheuristic.c:425-507.  Case 7 is the one the VNS kick uses; 1-3 are single reversals;
 * 4-6 reproduce the reference's two-step sequences (including their variable shuffles). */
ERROR_CODE tabu_make_move(int *prev, tsp_solution *solution, int bestCase, int i, int succ_i, int j, int succ_j,
                          int k, int succ_k)
{
    int *own = NULL;
    if (!prev && bestCase != 7) {
        own = (int *)malloc((size_t)tsp_inst.nnodes * sizeof(int));
        for (int v = 0; v < tsp_inst.nnodes; v++) own[solution->path[v]] = v;
        prev = own;
    }
    int t;
    switch (bestCase) {
    case 1: ref_reverse_path(k, succ_k, i, succ_i, prev, solution->path); break;
    case 2: ref_reverse_path(j, succ_j, k, succ_k, prev, solution->path); break;
    case 3: ref_reverse_path(i, succ_i, j, succ_j, prev, solution->path); break;
    case 4:
        ref_reverse_path(i, succ_i, j, succ_j, prev, solution->path);
        j = succ_i; succ_i = j;
        ref_reverse_path(j, succ_j, k, succ_k, prev, solution->path);
        break;
    case 5:
        ref_reverse_path(k, succ_k, i, succ_i, prev, solution->path);
        t = i; i = succ_k; succ_k = t;
        ref_reverse_path(i, succ_i, j, succ_j, prev, solution->path);
        break;
    case 6:
        ref_reverse_path(j, succ_j, k, succ_k, prev, solution->path);
        t = k; k = succ_j; succ_j = t;
        ref_reverse_path(k, succ_k, i, succ_i, prev, solution->path);
        break;
    case 7:
        solution->path[i] = succ_j;
        solution->path[k] = succ_i;
        solution->path[j] = succ_k;
        break;
    default: break;
    }
    free(own);
    return T_OK;
}

/* metaheuristic.c:251-341.  The loop -- local search, incumbent, r = rand() % 9 - 2 kicks -- runs on the device
 * (tspgpu_vns_search: inside the LDS-resident kernel where the instance allows it).  The random numbers stay the
 * program's: they are drawn here from the glibc stream (tsp_rand_peek), handed over in order, and what the device did
 * not use stays queued for the next draw, so the stream is consumed exactly as by the reference binary.  TSP_VNS_HOST=1
 * keeps the loop on the host (one ref_2opt call per iteration, vns_kick above). */
ERROR_CODE mh_VNS(void)
{
    log_info("running Variable Neighborhood Search");
    const int n = tsp_inst.nnodes;
    tsp_solution s, best;
    tsp_init_solution(n, &s);
    ERROR_CODE e = h_Greedy_iterative();
    if (!err_ok(e)) { log_fatal("code %d : Error in greedy", e); tsp_handlefatal(); }
    memcpy(s.path, tsp_inst.best_solution.path, (size_t)n * sizeof(int));
    s.cost = tsp_inst.best_solution.cost;
    tsp_init_solution(n, &best);
    best.cost = s.cost;
    memcpy(best.path, s.path, (size_t)n * sizeof(int));

    FILE *f = fopen("results/VNSResults.dat", "w+");
    e = T_OK;
    const char *on_host = getenv("TSP_VNS_HOST");
    if (on_host && atoi(on_host)) {
        for (int it = 0; it < tsp_env.k; it++) {
            if (past_deadline()) { e = DEADLINE_EXCEEDED; break; }
            e = ref_2opt(&s, tsp_inst.costs, true);
            if (!err_ok(e)) { log_fatal("code %d : Error in local search", e); tsp_handlefatal(); }
            if (s.cost < best.cost) {
                log_info("found new best: %f ", s.cost);
                best.cost = s.cost;
                memcpy(best.path, s.path, (size_t)n * sizeof(int));
            }
            if (f) fprintf(f, "%d,%f\n", it, s.cost);
            const int kicks = tsp_rand() % (UPPER - LOWER + 1) - LOWER;
            for (int j = 0; j < kicks; j++) vns_kick(&s);
        }
    } else {
        tspgpu_ctx *g = tsp_gpu();
        if (!g) return UNAVAILABLE;
        enum { CHUNK = 16384 };                     /* iterations per call: bounds the trace buffer and the numbers drawn ahead */
        double *trace = f ? (double *)malloc(CHUNK * sizeof(double)) : NULL;
        int it = 0, pending = 0;
        while (it < tsp_env.k) {
            if (past_deadline()) { e = DEADLINE_EXCEEDED; break; }
            const int upto = tsp_env.k - it > CHUNK ? it + CHUNK : tsp_env.k;
            const long want = 32L * (upto - it) + 1024;
            const int *rv = tsp_rand_peek(want);
            if (!rv) { log_fatal("out of memory for %ld random numbers", want); tsp_handlefatal(); }
            long used = 0;
            const int it0 = it;
            const double best0 = best.cost;
            const int rc = tspgpu_vns_search(g, s.path, &s.cost, upto, time_left(), rv, want, &used, &it, &pending, best.path, &best.cost, trace);
            tsp_rand_consume(used);
            /* RESOURCE_EXHAUSTED means "the numbers ran out in front of a kick phase: come back with more" only when the call
             * got somewhere (iterations done or numbers consumed: `want` covers far more than one kick phase).  The same code
             * with no progress is a real failure (the resident loop forced where it does not apply, a plan that does not
             * fit) and calling again would spin for ever */
            const int refill = rc == RESOURCE_EXHAUSTED && (it > it0 || used > 0);
            if (rc != 0 && rc != DEADLINE_EXCEEDED && !refill) {
                log_fatal("code %d : Error in local search: %s", rc, tspgpu_last_error(g));
                tsp_handlefatal();
            }
            if (best.cost < best0) log_info("found new best: %f ", best.cost);
            if (f && trace) for (int i = it0; i < it; i++) fprintf(f, "%d,%f\n", i, trace[i - it0]);
            if (rc == DEADLINE_EXCEEDED) { e = DEADLINE_EXCEEDED; break; }
        }
        free(trace);
    }
    if (f) fclose(f);
    ERROR_CODE u = tsp_update_best_solution(&best);
    if (!err_ok(u)) log_error("code %d : error in updating best solution of VNS", u);
    free(best.path); free(best.comp); free(s.path); free(s.comp);
    return e;
}

/* ===================================================================== main.c:4-87 */
ERROR_CODE tsp_run_algorithm(void)
{
    free(tsp_inst.best_solution.path);
    tsp_inst.best_solution.path = (int *)calloc((size_t)tsp_inst.nnodes, sizeof(int));
    switch (tsp_inst.alg) {
    case ALG_GREEDY: return h_Greedy();
    case ALG_GREEDY_ITER: return h_Greedy_iterative();
    case ALG_2OPT_GREEDY: return h_greedy_2opt();
    case ALG_TABU_SEARCH: return mh_TabuSearch();
    case ALG_VNS: return mh_VNS();
    case ALG_EXTRAMILEAGE: return h_ExtraMileage();
    default:
        log_error("algorithm %d belongs to the CPLEX path, which this library leaves to the reference build", tsp_inst.alg);
        return UNIMPLEMENTED;
    }
}
