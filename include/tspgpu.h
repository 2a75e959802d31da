/*
 * include/tspgpu.h -- C ABI of the MI355X (gfx950) 2-opt local-search engine.
 *
 * This is the drop-in boundary for the heuristic path of
 * enricobolzonello/TravellingSalesmanOptimization.  The reference has no
 * plugin/FFI layer: its boundary is a set of plain C functions over two
 * process-wide globals (src/tsp.h:235-236).  Each entry point below names the
 * reference function (file:line under the reference checkout) whose work it
 * takes over; the host-side C layer in travellingsalesmanoptimization_amd/host/
 * keeps the reference's own signatures on top of these.
 *
 * Conventions
 *   - plain pointers and sizes only; no global state; one opaque context per
 *     caller thread (contexts are independent, so the CPLEX callback threads
 *     of src/algorithms/cplex_model.c:1176-1258 can each own one).
 *   - return value: the reference's ERROR_CODE numbering
 *     (src/utils/errors.h:33-51): 0 T_OK, 3 INVALID_ARGUMENT,
 *     4 DEADLINE_EXCEEDED (a success, src/utils/errors.c:31-37),
 *     8 RESOURCE_EXHAUSTED, 9 FAILED_PRECONDITION, 12 UNIMPLEMENTED,
 *     13 INTERNAL (HIP runtime failure), 14 UNAVAILABLE (no device).
 *   - tours are SUCCESSOR arrays, path[i] = node visited after node i
 *     (src/algorithms/refinment.c:51-52), exactly as tsp_solution.path.
 *   - cost matrices are row-major n x n doubles, as tsp_inst.costs.
 *   - there is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef TSPGPU_H
#define TSPGPU_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tspgpu_ctx tspgpu_ctx;

/* edge-weight kinds for tspgpu_set_points.  EUC_2D reproduces src/tsp.c:629
 * bit for bit (float sqrt of a double sum).  ATT / CEIL_2D are TSPLIB 95
 * definitions in double; the reference rejects them (src/tsp.c:576-584). */
enum { TSPGPU_EUC_2D = 0, TSPGPU_ATT = 1, TSPGPU_CEIL_2D = 2 };

/* storage of the device-resident cost matrix.  AUTO keeps the narrowest EXACT copy:
 * uint16 when every entry is an integer in [0, 65534] (diagonal -1), int32 when every
 * entry is an integer in [-1, 2^27) (true for every EUC_2D / ATT / CEIL_2D matrix),
 * else doubles, the reference's own format. */
enum { TSPGPU_ELEM_AUTO = 0, TSPGPU_ELEM_F64 = 1, TSPGPU_ELEM_I32 = 2, TSPGPU_ELEM_U16 = 3 };

/* tunables (tspgpu_set_option) */
enum {
    TSPGPU_OPT_ELEM = 1,        /* TSPGPU_ELEM_*; takes effect at the next build/set_costs */
    TSPGPU_OPT_KERNEL = 2,      /* 0 auto, 1 "simple", 2 "pipelined", 3 "resident" sweep kernel (info reports 4 = matrix-free) */
    TSPGPU_OPT_BATCH = 3,       /* sweeps enqueued between host polls (default 32) */
    TSPGPU_OPT_WGS_PER_TOUR = 4,/* workgroups per tour in the sweep (0 = auto) */
    TSPGPU_OPT_HISTORY = 5,     /* record (a,b,delta) of the first N sweeps of slot 0 */
    TSPGPU_OPT_GRAPH = 6,       /* 1 = replay sweep batches as a hipGraph (default 1) */
    TSPGPU_OPT_TIMING = 7,      /* 1 = bracket every sweep kernel with HIP events */
    TSPGPU_OPT_BLOCK = 8,       /* threads per sweep workgroup (0 = auto) */
    TSPGPU_OPT_MAX_TOURS = 9,   /* tours kept in flight by the multi-start driver */
    TSPGPU_OPT_DEPTH = 10,      /* matrix rows in flight per workgroup in the pipelined sweep (0 = auto) */
    TSPGPU_OPT_FUSED = 12,      /* one launch per sweep (resident kernel): 1 (default) when <= 4 tours are in flight,
                                   2 always, 0 never (separate sweep + apply launches) */
    TSPGPU_OPT_MATRIX_FREE = 11,/* 0 auto (matrix-free when a matrix row cannot sit in LDS), 1 always, 2 never;
                                   takes effect at the next tspgpu_build_costs */
    TSPGPU_OPT_PIPE2 = 15,      /* one-launch-per-sweep kernel over streamed rows: 1 (default) two tour edges per barrier
                                   interval where four rows fit LDS, 0 one edge per barrier over three row buffers */
    TSPGPU_OPT_NN_KERNEL = 14,  /* nearest-neighbour construction: 0 auto (the grid kernel whenever the weights come from
                                   the uploaded points -- with every point's 3 nearest neighbours in LDS where they fit --,
                                   else the matrix kernel), 1 matrix / strided kernels always, 3 the grid kernel without
                                   the neighbour lists */
    TSPGPU_OPT_SWEEP_CAP = 13,  /* sweeps per start in tspgpu_multistart_nn_2opt (-1 = to the local optimum, the
                                   reference's behaviour; >= 0 caps every local search: tests and bounded runs) */
    TSPGPU_OPT_PERSIST = 16,    /* single-tour descent with the uint16 matrix resident in LDS, one launch per descent
                                   (whole rows for n <= 4096, half-window rows up to n of about 5400, one workgroup per CU),
                                   and tspgpu_tabu_search's walk the same way (n up to about 3800): 0 never, 1 (default)
                                   where it applies -- falls back to one launch per sweep when the grid cannot be
                                   co-resident --, 2 or fail with code 8 */
    TSPGPU_OPT_PERSIST_EDGES = 17, /* tour edges per workgroup of that kernel (0 = auto: ceil(n / CUs); at most 16 with whole
                                   rows, 24 with half-window rows) */
    TSPGPU_OPT_BUILD_KERNEL = 19,  /* tspgpu_build_costs with uint16 cells: 0 (default) the upper triangle computed once, every
                                   64 x 64 tile stored twice (as it is and transposed through LDS), 1 every cell computed
                                   (what int32 / f64 cells always do) */
    TSPGPU_OPT_STREAM_PERSIST = 20,/* single-tour descent past the LDS-resident sizes (uint16 cells, n from about 5400 to 16383)
                                   in ONE launch with the rows streamed and the tour state kept on the chip (k_str2opt, one
                                   workgroup per CU, one grid-wide exchange per sweep): 0 never, 1 (default) where it applies
                                   -- falls back to one launch per sweep when the grid cannot be co-resident --, 2 or fail with
                                   code 8 (and used from n = 1024 up) */
    TSPGPU_OPT_PERSIST_WINDOW = 18 /* rows of that kernel: 0 auto (whole rows where they fit the chip's LDS, else the half
                                   window of n/2 cells ahead of the workgroup's own edges), 1 half-window rows wherever they
                                   apply, 2 whole rows only */
};

int  tspgpu_device_count(void);
int  tspgpu_create(int device, tspgpu_ctx **out);
void tspgpu_destroy(tspgpu_ctx *ctx);
const char *tspgpu_last_error(const tspgpu_ctx *ctx);
int  tspgpu_set_option(tspgpu_ctx *ctx, int option, long value);
/* info: 0 n, 1 row stride, 2 element kind in use, 3 sweep kernel in use,
 * 4 workgroups per tour, 5 LDS bytes per workgroup, 6 threads per workgroup,
 * 7 matrix is symmetric, 8 compute units, 9 rows in flight per workgroup,
 * 10 matrix-free mode in use, 11 one-launch-per-sweep path in use, 12 cells per side of the NN grid (0: the
 * grid kernel is not in use), 13 most points in one grid cell, 14 the fused streaming kernel takes two edges per
 * barrier interval, 15 the last single-tour descent ran LDS-resident (TSPGPU_OPT_PERSIST), 16 / 17 / 18 workgroups, tour
 * edges per workgroup and LDS bytes per workgroup of that kernel on this instance (0: it does not apply), 19 window cells per
 * row of its half-window form (0: whole rows), 20 the last single-tour descent ran in the half-window form, 21 the last
 * single-tour descent began LDS-resident and was finished one launch per sweep (the grid lost its co-residency), 22 sweeps run by the last
 * LDS-resident descent / tabu walk / VNS walk, 23 how the last tspgpu_vns_search ran (1 resident throughout, 2 one device local
 * search per iteration with the kicks on the host, 3 resident launches first, then -- the grid lost its co-residency -- host kicks),
 * 24 the last single-tour descent ran in the streamed persistent kernel (TSPGPU_OPT_STREAM_PERSIST) */
long tspgpu_info(const tspgpu_ctx *ctx, int what);

/* ---- instance / cost matrix ------------------------------------------- */

/* Upload n points ({x,y} doubles = the reference's `point`, src/utils/utils.h:37-40). */
int tspgpu_set_points(tspgpu_ctx *ctx, const double *xy, int n, int edge_weight_type);

/* Replaces tsp_compute_costs (src/tsp.c:608-636): builds the n x n matrix on
 * the device from the uploaded points.  host_out may be NULL; otherwise it
 * receives the row-major n x n doubles (what tsp_inst.costs holds). */
int tspgpu_build_costs(tspgpu_ctx *ctx, double *host_out);

/* Caller-supplied matrix, the h_Greedy_2opt_mod_costs case
 * (src/algorithms/heuristics.c:118-149): row-major n x n doubles. */
int tspgpu_set_costs(tspgpu_ctx *ctx, const double *host_costs, int n);

/* Read the device matrix back (row-major n x n doubles). */
int tspgpu_get_costs(tspgpu_ctx *ctx, double *host_out);

/* ---- single-tour entry points (host arrays in/out) --------------------- */

/* h_greedyutil (src/algorithms/heuristics.c:216-288): nearest-neighbour tour
 * from `start`, ties to the lowest index.  14 if start is out of range. */
int tspgpu_nn_tour(tspgpu_ctx *ctx, int start, int *path, double *cost);

/* ref_2opt_once (src/algorithms/refinment.c:39-93): one best-improvement sweep
 * over all pairs; applies the move when delta < -1e-7 and adds delta to *cost.
 * *delta receives the best delta (0 when no improving pair exists). */
int tspgpu_two_opt_once(tspgpu_ctx *ctx, int *path, double *cost, double *delta);

/* ref_2opt (src/algorithms/refinment.c:3-37): recomputes *cost from the
 * matrix, then sweeps to the local optimum.  time_left_s < 0 = no deadline
 * (tsp_env.timelimit == -1); otherwise 4 is returned once it is exceeded,
 * polled once per batch of sweeps.  *sweeps (may be NULL) counts sweeps, the
 * final non-improving one included. */
int tspgpu_two_opt(tspgpu_ctx *ctx, int *path, double *cost, double time_left_s, long *sweeps);

/* tabu_best_move (src/algorithms/metaheuristic.c:188-245): best non-tabu move,
 * always applied; stamps tabu_list[a,b,succ a,succ b] = iter. */
int tspgpu_tabu_move(tspgpu_ctx *ctx, int *path, double *cost, int *tabu_list, int tenure, int iter);

/* the k-iteration loop of mh_TabuSearch (src/algorithms/metaheuristic.c:115-166)
 * with tabu_init (:65-84) and the linear tenure policy (:40-59), resident on
 * the device.  In: path/cost = the seed.  Out: path/cost = the walk's final
 * tour, best_path/best_cost = the incumbent (strict <, src/tsp.c:669-676).
 * trace (may be NULL) receives the k per-iteration costs that the reference
 * prints to results/TabuResults.dat. */
int tspgpu_tabu_search(tspgpu_ctx *ctx, int *path, double *cost, int k,
                       int *best_path, double *best_cost, double *trace);

/* the loop of mh_VNS (src/algorithms/metaheuristic.c:279-318): k iterations of { ref_2opt (:290), incumbent (:298-302),
 * r = rand() % 9 - 2 kicks (:308-318; vns_kick :344-409 = three tour positions under the reference's rejection rule, then
 * tabu_make_move case 7, :490-500) }, resident on the device where the instance allows it (uint16 cells, n up to about
 * 5400, an idle chip: the whole loop inside the LDS-resident kernel -- every workgroup applies the same kicks to its own
 * copy of the tour, no exchange --, else one device local search per iteration with the kicks on the host).
 * The random numbers are the CALLER's: rand_values[0 .. nrand) are rand() outputs drawn from the program's stream in
 * order; *consumed says how many the call used, so that the caller's stream can continue exactly where the reference's
 * would (host/tsp_algos.c keeps the rest queued).  In / out: path (the current tour; *cost is recomputed, refinment.c:6-9),
 * *iterations (completed so far), *kick_pending (1: the local search of iteration *iterations is done, its kicks are not),
 * best_path / *best_cost (the incumbent, strict <).  trace (may be NULL, else room for k - *iterations doubles): the cost
 * of every local optimum reached by this call, trace[0] = iteration *iterations at entry (what the reference prints to
 * results/VNSResults.dat).  Returns 0 when *iterations == k, 4 when the deadline passed,
 * 8 when the numbers ran out in front of a kick phase -- state consistent, call again with more. */
int tspgpu_vns_search(tspgpu_ctx *ctx, int *path, double *cost, int k, double time_left_s,
                      const int *rand_values, long nrand, long *consumed, int *iterations, int *kick_pending,
                      int *best_path, double *best_cost, double *trace);

/* ---- multi-start entry points ------------------------------------------ */

/* h_Greedy_iterative (src/algorithms/heuristics.c:34-72): NN from every listed
 * start (starts == NULL: 0..nstarts-1), first strictly-best kept. */
int tspgpu_nn_all(tspgpu_ctx *ctx, const int *starts, int nstarts,
                  int *best_path, double *best_cost, int *best_start);
/* The same under the reference's cooperative deadline (heuristics.c:43-49: checked before every
 * start): starts are processed in ascending batches sized to the time left (time_left_s < 0:
 * no limit); returns DEADLINE_EXCEEDED (4) with the best of the *done_starts processed so far
 * (best_start = -1 if none). */
int tspgpu_nn_all_timed(tspgpu_ctx *ctx, const int *starts, int nstarts, double time_left_s,
                        int *best_path, double *best_cost, int *best_start, int *done_starts);

/* h_greedy_2opt (src/algorithms/heuristics.c:74-116): NN + 2-opt from every
 * listed start, all on the device; the winner is the lowest cost, ties to the
 * earliest entry of `starts`.  last_path/last_cost (may be NULL) receive the
 * tour of the LAST start, which is what h_Greedy_2opt_mod_costs leaves in
 * *solution (src/algorithms/heuristics.c:118-149). */
int tspgpu_multistart_nn_2opt(tspgpu_ctx *ctx, const int *starts, int nstarts,
                              double time_left_s, int *best_path, double *best_cost,
                              int *best_start, long *total_sweeps,
                              int *last_path, double *last_cost);

/* ---- multi-device multi-start (csrc/tspgpu_multi.cpp) ------------------------
 * The reference's multi-start loops (h_greedy_2opt, src/algorithms/heuristics.c:82-111; h_Greedy_iterative, :43-66)
 * are sequential C; their iterations are independent except for the incumbent minimum (src/tsp.c:669-676, strict <).
 * Here one process drives several MI355X: one engine context and one host thread per device, entry p of the start
 * list on device p mod G, every device building its own matrix from the coordinates, and ONE exchange per call:
 * ncclAllReduce(ncclMin) over xGMI of one packed int64 per device (cost:31 | list position:24 | device rank:8: lowest
 * cost, ties to the earliest start -- the sequential strict-< result -- and the low byte names the owner), then
 * ncclBroadcast of the owner's successor array (4n bytes).  RCCL is dlopen'ed at the first exchange that needs it.
 * A device id may be listed more than once (several contexts on one GPU, how the sharding is exercised on a
 * one-GPU box); an RCCL communicator needs distinct devices, so that list exchanges on the host. */
typedef struct tspgpu_multi tspgpu_multi;
enum { TSPGPU_MOPT_EXCHANGE = 1000 };   /* tspgpu_multi_set_option: 0 auto (RCCL when G > 1 distinct devices, none for
                                           G = 1, host for a repeated device), 1 host, 2 RCCL (also with G = 1: a
                                           one-rank communicator; refused for a repeated device).  Every other option
                                           is a TSPGPU_OPT_* applied to each device's context. */
/* the winner among G per-device results by the host exchange's order (by_keys = 0) or by the keys the RCCL exchange reduces
 * (by_keys = 1); pos[i] < 0 = device i found nothing; returns the device rank, -1 if nobody, -2 on a bad argument.  Pure
 * host arithmetic (no device needed): what pins both selection orders in the CPU tests. */
int  tspgpu_multi_select(const double *cost, const long *pos, int ndev, int by_keys);
int  tspgpu_multi_create(const int *device_ids, int ndev, tspgpu_multi **out);
void tspgpu_multi_destroy(tspgpu_multi *m);
const char *tspgpu_multi_last_error(const tspgpu_multi *m);
int  tspgpu_multi_devices(const tspgpu_multi *m);
tspgpu_ctx *tspgpu_multi_ctx(tspgpu_multi *m, int i);          /* the i-th device's context (owned by m) */
/* info: 0 devices, 1 exchange the next call will use (0 none, 1 host, 2 RCCL), 2 exchange used by the last call,
 * 3 seconds spent in ncclCommInitAll, 4 seconds of the last exchange, 5 seconds of the last per-device solve,
 * 6 exchanges so far, 7 the device ids are distinct */
double tspgpu_multi_info(const tspgpu_multi *m, int what);
int  tspgpu_multi_set_option(tspgpu_multi *m, int option, long value);
/* create the RCCL communicator now if the next exchange will use one (ncclCommInitAll takes seconds on 8 devices;
 * the reference starts its clock after tsp_compute_costs, src/main.c:177 -- the host layer calls this there) */
int  tspgpu_multi_prepare(tspgpu_multi *m);
int  tspgpu_multi_set_points(tspgpu_multi *m, const double *xy, int n, int edge_weight_type);
int  tspgpu_multi_build_costs(tspgpu_multi *m);                 /* tsp_compute_costs on every device */
/* h_greedy_2opt (src/algorithms/heuristics.c:74-116) sharded over the devices; same results as
 * tspgpu_multistart_nn_2opt over the whole list when no deadline is set */
int  tspgpu_multi_multistart_nn_2opt(tspgpu_multi *m, const int *starts, int nstarts, double time_left_s,
                                     int *best_path, double *best_cost, int *best_start, long *total_sweeps);
/* h_Greedy_iterative (src/algorithms/heuristics.c:34-72) sharded the same way */
int  tspgpu_multi_nn_all(tspgpu_multi *m, const int *starts, int nstarts, double time_left_s,
                         int *best_path, double *best_cost, int *best_start, int *done_starts);

/* ---- device-resident variants (inputs already in HBM; used by bench.py) --- */

/* Tour slots: the slot array grows on demand and keeps what the existing slots hold; a slot holds a tour once
 * something was loaded / built / copied into it, and until the next tspgpu_build_costs / tspgpu_set_costs (its edge
 * costs belong to the matrix).  Slot entry points answer FAILED_PRECONDITION (9) for a slot that holds none.
 * The host-array entry points above and the multi-start entry points use slots from 0 upwards as scratch. */
/* upload a successor array into tour slot `slot` */
int tspgpu_tour_load(tspgpu_ctx *ctx, int slot, const int *path);
/* NN tour built on the device straight into a slot */
int tspgpu_tour_nn(tspgpu_ctx *ctx, int slot, int start);
/* copy slot src to slot dst on the device */
int tspgpu_tour_copy(tspgpu_ctx *ctx, int dst, int src);
/* sweep slot to its local optimum (max_sweeps < 0: no cap) */
int tspgpu_tour_two_opt(tspgpu_ctx *ctx, int slot, long max_sweeps, double time_left_s, long *sweeps);
/* Intra-sweep sharding ("optional, config 5": one sweep of a large instance split over
 * the GPUs of a node).  Every rank holds the same tour in `slot`; tspgpu_tour_sweep_part evaluates
 * the runs [part*G/nparts, (part+1)*G/nparts) of ONE sweep (refinment.c:49-69) and returns the best
 * pair found there (delta 0, a = b = 0: nothing improving in this part); after the ranks have
 * agreed on the minimum of (delta, a, b) -- one MIN all-reduce -- each applies it with
 * tspgpu_tour_apply_move (refinment.c:74-86,95-114), which also counts the sweep; a delta >= 0
 * marks the slot as locally optimal.  Symmetric matrices only. */
int tspgpu_tour_sweep_part(tspgpu_ctx *ctx, int slot, int part, int nparts, double *delta, int *a, int *b);
int tspgpu_tour_apply_move(tspgpu_ctx *ctx, int slot, int a, int b, double delta);
/* fetch slot's successor array / cost / last delta */
int tspgpu_tour_store(tspgpu_ctx *ctx, int slot, int *path, double *cost, double *last_delta);
/* launch the sweep kernel alone `reps` times on slot (no move applied) and
 * return its mean duration in ms from HIP events on the engine's stream */
int tspgpu_time_sweep(tspgpu_ctx *ctx, int slot, int reps, float *ms_mean);
/* same for the matrix build kernel */
int tspgpu_time_build(tspgpu_ctx *ctx, int reps, float *ms_mean);
/* with TSPGPU_OPT_TIMING: sum of sweep-kernel ms and launch count since reset */
int tspgpu_timing_read(tspgpu_ctx *ctx, double *sweep_ms_total, long *sweep_launches, int reset);
/* diagnostics: 64 wall-clock stamps (10 ns ticks) per sweep workgroup of the last launch
 * made while stamping was enabled (tools/stamps.py); not part of the reference's surface */
int tspgpu_debug_stamps(tspgpu_ctx *ctx, unsigned long long *out, int capacity_words);
/* with TSPGPU_OPT_HISTORY: the recorded moves of slot 0; returns count in *count */
int tspgpu_history(tspgpu_ctx *ctx, int *a, int *b, double *delta, int capacity, int *count);

#ifdef __cplusplus
}
#endif
#endif /* TSPGPU_H */
