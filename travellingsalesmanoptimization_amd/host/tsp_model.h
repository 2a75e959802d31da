/*
 * tsp_model.h -- host-side data model and entry points of the heuristic path, kept
 * source-compatible with enricobolzonello/TravellingSalesmanOptimization so that code
 * written against the reference's headers (src/tsp.h, src/utils/utils.h,
 * src/utils/errors.h, src/algorithms/{heuristics,refinment,metaheuristic}.h) compiles
 * and links against this library unchanged -- the hot loops run on an MI355X through
 * include/tspgpu.h instead of on the CPU.
 *
 * Struct layouts, enum values and function signatures are the reference's (they are
 * the drop-in contract: SURVEY 8b); everything behind them is new.  Each prototype
 * names the reference definition it stands in for.
 */
#ifndef TSP_MODEL_H
#define TSP_MODEL_H

#include <stdbool.h>
#include <stdio.h>
#include <time.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/utils/errors.h:33-51, :56-59, :69-74 --------------------------------- */
typedef enum {
    T_OK = 0, CANCELLED = 1, UNKNOWN = 2, INVALID_ARGUMENT = 3, DEADLINE_EXCEEDED = 4,
    NOT_FOUND = 5, ALREADY_EXISTS = 6, PERMISSION_DENIED = 7, UNAUTHENTICATED = 16,
    RESOURCE_EXHAUSTED = 8, FAILED_PRECONDITION = 9, ABORTED = 10, OUT_OF_RANGE = 11,
    UNIMPLEMENTED = 12, INTERNAL = 13, UNAVAILABLE = 14, DATA_LOSS = 15
} ERROR_CODE;

typedef enum { LOG_TRACE, LOG_DEBUG, LOG_INFO, LOG_WARN, LOG_ERROR, LOG_FATAL } LOGGING_TYPE;
typedef enum { QUIET = 0, NORMAL = 1, VERBOSE = 2, VERY_VERBOSE = 3 } VERBOSITY;

bool err_ok(ERROR_CODE error);                       /* errors.c:31-37 */
void err_setverbosity(VERBOSITY verbosity);          /* errors.c:39-41 */
bool err_dolog(void);                                /* errors.c:43-45 */
void err_logging(LOGGING_TYPE level, const char *file, int line, const char *message, ...); /* errors.c:47-62 */
void err_printoutput(double cost, double time, int alg); /* errors.c:145-166: the -q stdout contract */
void err_setinfo(int alg, int nnodes, bool random, char *inputfile, double timelimit, int seed, int tabu_policy,
                 int em_init, bool init_mip, int bc_policy, bool callback_relaxation, double lb_improv,
                 int lb_delta, bool lb_kstar);       /* errors.c:83-143 */

#define log_trace(...) err_logging(LOG_TRACE, __FILE__, __LINE__, __VA_ARGS__)
#define log_debug(...) err_logging(LOG_DEBUG, __FILE__, __LINE__, __VA_ARGS__)
#define log_info(...)  err_logging(LOG_INFO, __FILE__, __LINE__, __VA_ARGS__)
#define log_warn(...)  err_logging(LOG_WARN, __FILE__, __LINE__, __VA_ARGS__)
#define log_error(...) err_logging(LOG_ERROR, __FILE__, __LINE__, __VA_ARGS__)
#define log_fatal(...) err_logging(LOG_FATAL, __FILE__, __LINE__, __VA_ARGS__)

/* ---- src/utils/utils.h ---------------------------------------------------------- */
#define MAX_COORDINATE 5000
#define MIN_COORDINATE -5000
#define TSP_RAND() (((double)tsp_rand() / RAND_MAX) * (MAX_COORDINATE - MIN_COORDINATE) + MIN_COORDINATE)
int tsp_rand(void);            /* glibc rand() on the program's private stream (see tsp_log.c) */
void tsp_srand(unsigned seed);
const int *tsp_rand_peek(long count);   /* the next `count` values of that stream, not consumed (for tspgpu_vns_search) */
void tsp_rand_consume(long count);      /* ... of which the first `count` have now been used */
#define NOT_CONNECTED -1.0f
#define utils_safe_free(pointer) utils_safe_memory_free((void **)&(pointer))

typedef struct { double x; double y; } point;                           /* utils.h:37-40 */
typedef struct { double cost; int *path; int ncomp; int *comp; } tsp_solution; /* utils.h:42-47 */

void utils_safe_memory_free(void **pointer_address);
bool utils_file_exists(const char *filename);
void utils_startclock(struct timespec *c);            /* utils.c:31-35 */
double utils_timeelapsed(struct timespec *c);         /* utils.c:37-44 */
void swap(int *a, int *b);
ERROR_CODE tsp_init_solution(int nnodes, tsp_solution *solution); /* utils.c:137-154 */

/* ---- src/tsp.h ------------------------------------------------------------------ */
#define EPSILON -1.0E-7

typedef enum { POL_FIXED = 0, POL_SIZE = 1, POL_RANDOM = 2, POL_LINEAR = 3 } ts_policies;
typedef enum { EM_MAX = 0, EM_RANDOM = 1 } em_init;
typedef enum { BC_PROB = 0, BC_NODES = 1, BC_DEPTH = 2 } bc_skip;
typedef enum {
    ALG_GREEDY = 0, ALG_GREEDY_ITER = 1, ALG_2OPT_GREEDY = 2, ALG_TABU_SEARCH = 3, ALG_VNS = 4,
    ALG_CX_NOSEC = 5, ALG_CX_BENDERS = 6, ALG_EXTRAMILEAGE = 7, ALG_CX_BENDERS_PAT = 8,
    ALG_CX_BRANCH_AND_CUT = 9, ALG_HARD_FIXING = 10, ALG_LOCAL_BRANCHING = 11
} algorithms;

typedef struct { double cost; int *path; point *points; int nnodes; double execution_time; } return_struct;

typedef struct {                /* tsp.h:66-103 */
    double timelimit; int seed; bool graph_random; bool graph_input; char *inputfile; bool tofile; int k;
    ts_policies policy;
    em_init mileage_init;
    bool bl_patching;
    bool init_mip; bc_skip skip_policy; bool callback_relaxation; bool modified_costs;
    double hf_prob;
    bool lb_dynk; int lb_initk; double lb_improv; int lb_delta; bool lb_kstar;
} options;

typedef struct { int tenure; int max_tenure; int min_tenure; bool increment; int *tabu_list; } tabu_search; /* tsp.h:105-113 */

typedef struct {                /* tsp.h:115-134 */
    algorithms alg;
    int nnodes;
    struct timespec c;
    point *points;
    double *costs;
    tsp_solution best_solution;
    int starting_node;
    int *threads_seeds;
    int ncols;
    int cplex_terminate;
} instance;

extern instance tsp_inst;       /* tsp.h:235 */
extern options tsp_env;         /* tsp.h:236 */

void tsp_init(void);                                         /* tsp.c:6-44 */
ERROR_CODE tsp_parse_commandline(int argc, char **argv);     /* tsp.c:46-466 */
ERROR_CODE tsp_generate_randompoints(void);                  /* tsp.c:468-481 */
void tsp_read_input(void);                                   /* tsp.c:527-606 */
ERROR_CODE tsp_compute_costs(void);                          /* tsp.c:608-636  -> k_build_costs on the device */
double tsp_get_cost(int i, int j);                           /* tsp.c:638-640 */
bool tsp_validate_solution(int nnodes, int *current_solution_path); /* tsp.c:642-667 */
ERROR_CODE tsp_update_best_solution(tsp_solution *current_solution); /* tsp.c:669-684 */
bool tsp_is_tour(int path[], int n);                         /* tsp.c:687-728 */
double tsp_solution_cost(int path[]);                        /* tsp.c:730-736 */
void tsp_handlefatal(void);                                  /* tsp.c:738-742 */
void tsp_free_instance(void);                                /* tsp.c:744-751 */

/* ---- src/algorithms/refinment.h -------------------------------------------------- */
ERROR_CODE ref_2opt(tsp_solution *solution, double *costs, bool update_incumbent); /* refinment.c:3-37 -> device */
double ref_2opt_once(tsp_solution *solution, double *costs);                       /* refinment.c:39-93 -> device */
void ref_reverse_path(int a, int succ_a, int b, int succ_b, int *prev, int *solution_path); /* refinment.c:95-114 */

/* ---- src/algorithms/heuristics.h ------------------------------------------------- */
ERROR_CODE h_Greedy(void);                                   /* heuristics.c:12-32 */
ERROR_CODE h_Greedy_iterative(void);                         /* heuristics.c:34-72  -> device, all starts batched */
ERROR_CODE h_greedy_2opt(void);                              /* heuristics.c:74-116 -> device, all starts batched */
ERROR_CODE h_Greedy_2opt_mod_costs(tsp_solution *solution, double *costs); /* heuristics.c:118-149 -> device */
ERROR_CODE h_greedyutil(int starting_node, tsp_solution *solution, double *costs); /* heuristics.c:216-288 -> device */
ERROR_CODE h_ExtraMileage(void);                             /* heuristics.c:156-210: not on the hot path, UNIMPLEMENTED */

/* ---- src/algorithms/metaheuristic.h ---------------------------------------------- */
#define UPPER 10
#define LOWER 2
#define MAX_FRACTION 0.25
#define MIN_FRACTION 0.125

ERROR_CODE mh_TabuSearch(void);                              /* metaheuristic.c:86-186 -> device */
ERROR_CODE mh_VNS(void);                                     /* metaheuristic.c:251-341: 2-opt on device, kicks on host */
ERROR_CODE vns_kick(tsp_solution *solution);                 /* metaheuristic.c:344-409 */
ERROR_CODE tabu_fixed_policy(tabu_search *t, int value);     /* metaheuristic.c:7-16 */
ERROR_CODE tabu_dependent_policy(tabu_search *t);            /* metaheuristic.c:18-27 */
ERROR_CODE tabu_random_policy(tabu_search *t);               /* metaheuristic.c:29-38 */
ERROR_CODE tabu_linear_policy(tabu_search *ts);              /* metaheuristic.c:40-59 */
ERROR_CODE tabu_init(tabu_search *ts, int nnodes);           /* metaheuristic.c:65-84 */
ERROR_CODE tabu_best_move(int *solution_path, double *solution_cost, tabu_search *ts, int current_iteration); /* :188-245 -> device */
bool is_in_tabu_list(tabu_search *ts, int b, int current_iteration); /* metaheuristic.c:416-418 */
void tabu_free(tabu_search *ts);
ERROR_CODE tabu_make_move(int *prev, tsp_solution *solution, int bestCase, int i, int succ_i, int j, int succ_j,
                          int k, int succ_k);                /* metaheuristic.c:425-507 */

/* ---- src/main.c ------------------------------------------------------------------ */
ERROR_CODE tsp_run_algorithm(void);                          /* main.c:4-87 (heuristic algorithms only) */

/* ---- additions of this library (not in the reference) ---------------------------- */
/* edge-weight kind read from the TSPLIB header: 0 EUC_2D, 1 ATT, 2 CEIL_2D.  The reference
 * accepts EUC_2D only (tsp.c:576-584); the others are enabled by TSP_ALLOW_EXT=1. */
extern int tsp_edge_weight_kind;
/* true when no n x n matrix exists (n > 32 768 or TSP_MATRIX_FREE=1): tsp_inst.costs == NULL,
 * tsp_get_cost recomputes, the device runs its matrix-free kernels */
extern bool tsp_matrix_free;
extern bool tsp_lazy_costs;       /* tsp_inst.costs materialised on first use (tsp_core.c); the `tsp` executable sets it */
double *tsp_host_costs(void);     /* tsp_inst.costs, downloaded from the device if it is not on the host yet */
/* the device context behind tsp_inst.costs (created on first use; NULL if no MI355X) */
struct tspgpu_ctx;
struct tspgpu_ctx *tsp_gpu(void);
void tsp_gpu_release(void);
/* TSP_GPU_DEVICES="0,1,...": the multi-start loops (h_greedy_2opt, h_Greedy_iterative) shard their start nodes over
 * these devices (include/tspgpu.h, tspgpu_multi_*: one RCCL MIN all-reduce + one broadcast per call); NULL otherwise.
 * TSP_GPU_EXCHANGE=rccl|host overrides the automatic choice; TSP_GPU_STATS=1 prints one JSON line per call on stderr. */
struct tspgpu_multi;
struct tspgpu_multi *tsp_gpu_multi(void);
/* caller-matrix calls (h_Greedy_2opt_mod_costs from CPLEX callback threads, cplex_model.c:1176-1258) run in a
 * device context private to the calling thread; it is destroyed when the thread exits.  Live count / early release: */
int tsp_gpu_thread_contexts(void);
void tsp_gpu_release_threads(void);

#ifdef __cplusplus
}
#endif
#endif
