/*
 * tsp_core.c -- instance core of the host layer: globals, defaults, TSPLIB reader, random
 * instances, validation, incumbent, and tsp_compute_costs running on the MI355X.
 * Behaviour follows src/tsp.c of the reference (cited per function); the code is new.
 */
#include "tsp_model.h"

#include <libgen.h>
#include <pthread.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "tspgpu.h"

instance tsp_inst;
options tsp_env;
int tsp_edge_weight_kind = TSPGPU_EUC_2D;
bool tsp_matrix_free = false;
/* tsp_inst.costs on the host only when somebody reads it: the heuristic path never does (every O(n^2) loop runs on the
 * device), and downloading n x n doubles (2.74 GB for d18512) costs seconds before the clock even starts.  Off by
 * default -- code that indexes tsp_inst.costs directly (the reference's CPLEX path) finds it filled as before --;
 * the `tsp` executable, which only runs the heuristic path, turns it on (TSP_EAGER_COSTS=1 turns it off again).
 * With it on, tsp_inst.costs stays NULL until tsp_get_cost / tsp_host_costs() materialise it. */
bool tsp_lazy_costs = false;

static struct tspgpu_ctx *g_ctx = NULL;
static struct tspgpu_multi *g_multi = NULL;    /* set when TSP_GPU_DEVICES names the devices; owns g_ctx then */

/* TSP_GPU_DEVICES="0,1,2,3": the devices the multi-start loops (h_greedy_2opt, h_Greedy_iterative) shard over.
 * Unset: one device, TSP_GPU_DEVICE (default 0).  A device may be listed twice (two contexts on one GPU). */
static int device_list(int *out, int cap)
{
    const char *s = getenv("TSP_GPU_DEVICES");
    int k = 0;
    if (!s) return 0;
    while (*s == ' ') s++;
    for (;;) {
        char *end;
        const long v = strtol(s, &end, 10);
        if (end == s || v < 0 || v > 1023 || k >= cap) return -1;      /* "", "a,b", "0,,1", "-1": malformed */
        out[k++] = (int)v;
        s = end;
        while (*s == ' ') s++;
        if (!*s) return k;
        if (*s != ',') return -1;
        s++;
        while (*s == ' ') s++;
    }
}

struct tspgpu_ctx *tsp_gpu(void)
{
    if (!g_ctx) {
        int devs[64];
        const int nd = device_list(devs, 64);
        if (nd < 0) {      /* a typo must not silently turn an 8-GPU run into a 1-GPU run */
            log_fatal("TSP_GPU_DEVICES=\"%s\" is not a comma-separated list of device ids", getenv("TSP_GPU_DEVICES"));
            return NULL;
        }
        if (nd > 0) {
            int rc = tspgpu_multi_create(devs, nd, &g_multi);
            if (rc != 0) {
                log_fatal("TSP_GPU_DEVICES: tspgpu_multi_create over %d device(s) -> %d; there is no CPU fallback", nd, rc);
                g_multi = NULL;
                return NULL;
            }
            const char *ex = getenv("TSP_GPU_EXCHANGE");     /* "rccl" | "host" | "auto"; unset = automatic */
            if (ex && strcmp(ex, "rccl") && strcmp(ex, "host") && strcmp(ex, "auto")) {
                log_fatal("TSP_GPU_EXCHANGE=\"%s\": expected rccl, host or auto", ex);
                tspgpu_multi_destroy(g_multi); g_multi = NULL;
                return NULL;
            }
            if (ex && tspgpu_multi_set_option(g_multi, TSPGPU_MOPT_EXCHANGE, !strcmp(ex, "rccl") ? 2 : !strcmp(ex, "host") ? 1 : 0) != 0) {
                log_fatal("TSP_GPU_EXCHANGE=%s: %s", ex, tspgpu_multi_last_error(g_multi));
                tspgpu_multi_destroy(g_multi); g_multi = NULL;
                return NULL;
            }
            g_ctx = tspgpu_multi_ctx(g_multi, 0);
            return g_ctx;
        }
        const char *dev = getenv("TSP_GPU_DEVICE");
        tspgpu_ctx *c = NULL;
        int rc = tspgpu_create(dev ? atoi(dev) : 0, &c);
        if (rc != 0) {
            log_fatal("no MI355X (gfx950) device available: tspgpu_create -> %d; there is no CPU fallback", rc);
            return NULL;
        }
        g_ctx = c;
    }
    return g_ctx;
}

/* the multi-device handle, or NULL when the run is on one device */
struct tspgpu_multi *tsp_gpu_multi(void)
{
    tsp_gpu();
    return g_multi;
}

void tsp_gpu_release(void)
{
    tsp_gpu_release_threads();          /* contexts of caller-matrix threads that are still alive (tsp_algos.c) */
    if (g_multi) tspgpu_multi_destroy(g_multi);     /* owns every device's context, g_ctx included */
    else if (g_ctx) tspgpu_destroy(g_ctx);
    g_multi = NULL;
    g_ctx = NULL;
}

/* defaults: tsp.c:6-44 */
void tsp_init(void)
{
    memset(&tsp_env, 0, sizeof tsp_env);
    tsp_env.timelimit = -1;
    tsp_env.seed = -1;
    tsp_env.k = __INT_MAX__;
    tsp_env.policy = POL_LINEAR;
    tsp_env.mileage_init = EM_MAX;
    tsp_env.bl_patching = true;
    tsp_env.init_mip = true;
    tsp_env.skip_policy = BC_PROB;
    tsp_env.callback_relaxation = true;
    tsp_env.hf_prob = 0.7;
    tsp_env.lb_initk = 10;
    tsp_env.lb_improv = 0.02;
    tsp_env.lb_delta = 10;

    memset(&tsp_inst, 0, sizeof tsp_inst);
    tsp_inst.nnodes = -1;
    tsp_inst.best_solution.cost = __DBL_MAX__;
    tsp_inst.alg = ALG_GREEDY;
    tsp_inst.ncols = -1;
    tsp_edge_weight_kind = TSPGPU_EUC_2D;

    err_setverbosity(NORMAL);
}

/* tsp.c:468-481: glibc stream, x then y, node order */
ERROR_CODE tsp_generate_randompoints(void)
{
    tsp_srand((unsigned)tsp_env.seed);
    tsp_inst.points = (point *)calloc((size_t)tsp_inst.nnodes, sizeof(point));
    for (int i = 0; i < tsp_inst.nnodes; i++) {
        tsp_inst.points[i].x = TSP_RAND();
        tsp_inst.points[i].y = TSP_RAND();
    }
    tsp_edge_weight_kind = TSPGPU_EUC_2D;
    return tsp_compute_costs();
}

/* tsp.c:527-606.  Same keyword handling (prefix match, tokens split on " :"), same fatal
 * messages.  EDGE_WEIGHT_TYPE other than EUC_2D is fatal unless TSP_ALLOW_EXT=1. */
void tsp_read_input(void)
{
    FILE *in = fopen(tsp_env.inputfile, "r");
    if (!in) { log_fatal(" input file not found!"); tsp_handlefatal(); }

    tsp_inst.nnodes = -1;
    tsp_edge_weight_kind = TSPGPU_EUC_2D;
    const char *ext = getenv("TSP_ALLOW_EXT");
    const bool allow_ext = ext && atoi(ext) != 0;
    bool coords = false;
    char line[300];
    while (fgets(line, sizeof line, in)) {
        if (strlen(line) <= 1) continue;
        char *key = strtok(line, " :");
        if (!key) continue;
        if (!strncmp(key, "DIMENSION", 9)) {
            if (tsp_inst.nnodes >= 0) { log_fatal("two DIMENSION parameters in the file"); tsp_handlefatal(); }
            tsp_inst.nnodes = atoi(strtok(NULL, " :"));
            tsp_inst.points = (point *)calloc((size_t)tsp_inst.nnodes, sizeof(point));
        } else if (!strncmp(key, "NODE_COORD_SECTION", 18)) {
            if (tsp_inst.nnodes <= 0) { log_fatal("DIMENSION not found"); tsp_handlefatal(); }
            coords = true;
        } else if (!strncmp(key, "TYPE", 4)) {
            if (strncmp(strtok(NULL, " :"), "TSP", 3)) { log_fatal(" format error:  only TSP file type accepted"); tsp_handlefatal(); }
        } else if (!strncmp(key, "EDGE_WEIGHT_TYPE", 16)) {
            const char *w = strtok(NULL, " :\n");
            if (!strncmp(w, "EUC_2D", 6)) tsp_edge_weight_kind = TSPGPU_EUC_2D;
            else if (allow_ext && !strncmp(w, "ATT", 3)) tsp_edge_weight_kind = TSPGPU_ATT;
            else if (allow_ext && !strncmp(w, "CEIL_2D", 7)) tsp_edge_weight_kind = TSPGPU_CEIL_2D;
            else { log_fatal(" format error:  only EDGE_WEIGHT_TYPE == EUC_2D managed"); tsp_handlefatal(); }
        } else if (!strncmp(key, "EOF", 3)) {
            break;
        } else if (coords) {
            const int idx = atoi(key) - 1;
            const char *sx = strtok(NULL, " :,"), *sy = strtok(NULL, " :,");
            if (idx >= 0 && idx < tsp_inst.nnodes && sx && sy) {
                tsp_inst.points[idx].x = atof(sx);
                tsp_inst.points[idx].y = atof(sy);
            }
        }
    }
    fclose(in);
    ERROR_CODE e = tsp_compute_costs();
    if (!err_ok(e)) log_error("code error: %d", e);
}

/* tsp.c:608-636 -> tspgpu_set_points + tspgpu_build_costs (k_build_costs).  tsp_inst.costs is
 * still filled (row-major n x n doubles): the untouched CPLEX path and tsp_get_cost read it. */
ERROR_CODE tsp_compute_costs(void)
{
    if (tsp_inst.nnodes <= 0) { log_fatal("computing costs of empty graph"); tsp_handlefatal(); }
    struct tspgpu_ctx *g = tsp_gpu();
    if (!g) return UNAVAILABLE;
    struct tspgpu_multi *m = tsp_gpu_multi();
    const size_t n = (size_t)tsp_inst.nnodes;
    free(tsp_inst.costs);
    tsp_inst.costs = NULL;
    /* Matrix-free above 32 768 nodes (8 GB of doubles on the host; the reference overflows int
     * past 46 340 and needs 59 GB for pla85900) or on request: no n x n array anywhere, the
     * device recomputes weights from the coordinates and tsp_inst.costs stays NULL. */
    const char *mf = getenv("TSP_MATRIX_FREE");
    tsp_matrix_free = mf ? atoi(mf) != 0 : n > 32768;   /* an explicit 0 keeps the matrix at any size */
    int rc;
    if (m) {
        /* several devices: each builds its own matrix from the 16n-byte coordinate array (never shipped) */
        if ((rc = tspgpu_multi_set_option(m, TSPGPU_OPT_MATRIX_FREE, tsp_matrix_free ? 1 : 0)) ||
            (rc = tspgpu_multi_set_points(m, (const double *)tsp_inst.points, (int)n, tsp_edge_weight_kind)) ||
            (rc = tspgpu_multi_build_costs(m))) {
            log_error("multi-device cost build: %s", tspgpu_multi_last_error(m));
            return (ERROR_CODE)rc;
        }
        /* the RCCL communicator, if the exchange will use one: before the clock starts (main.c:177) */
        if ((rc = tspgpu_multi_prepare(m))) { log_error("tspgpu_multi_prepare: %s", tspgpu_multi_last_error(m)); return (ERROR_CODE)rc; }
        if (tsp_matrix_free || tsp_lazy_costs) return T_OK;
        return tsp_host_costs() ? T_OK : RESOURCE_EXHAUSTED;
    }
    rc = tspgpu_set_points(g, (const double *)tsp_inst.points, (int)n, tsp_edge_weight_kind);
    if (rc) { log_error("tspgpu_set_points: %s", tspgpu_last_error(g)); return (ERROR_CODE)rc; }
    if (tsp_matrix_free) {
        tspgpu_set_option(g, TSPGPU_OPT_MATRIX_FREE, 1);
        rc = tspgpu_build_costs(g, NULL);
        if (rc) { log_error("tspgpu_build_costs: %s", tspgpu_last_error(g)); return (ERROR_CODE)rc; }
        return T_OK;
    }
    tspgpu_set_option(g, TSPGPU_OPT_MATRIX_FREE, 0);
    if (tsp_lazy_costs) {
        rc = tspgpu_build_costs(g, NULL);
        if (rc) { log_error("tspgpu_build_costs: %s", tspgpu_last_error(g)); return (ERROR_CODE)rc; }
        return T_OK;
    }
    tsp_inst.costs = (double *)malloc(n * n * sizeof(double)); /* size_t: no int overflow past n = 46340 */
    if (!tsp_inst.costs) return RESOURCE_EXHAUSTED;
    rc = tspgpu_build_costs(g, tsp_inst.costs);
    if (rc) { log_error("tspgpu_build_costs: %s", tspgpu_last_error(g)); return (ERROR_CODE)rc; }
    return T_OK;
}

/* the host copy of the matrix (row-major n x n doubles, what the reference's tsp_inst.costs holds), downloaded from the
 * device by an EXPLICIT call when tsp_lazy_costs kept it away (2.74 GB for d18512: nothing on the heuristic path reads it);
 * NULL in matrix-free mode or when the allocation fails.  Serialised: two threads asking at once get one download. */
static pthread_mutex_t host_costs_mu = PTHREAD_MUTEX_INITIALIZER;
double *tsp_host_costs(void)
{
    if (tsp_inst.costs || tsp_matrix_free || tsp_inst.nnodes <= 0) return tsp_inst.costs;
    pthread_mutex_lock(&host_costs_mu);
    if (!tsp_inst.costs) {
        struct tspgpu_ctx *g = tsp_gpu();
        const size_t n = (size_t)tsp_inst.nnodes;
        double *c = g ? (double *)malloc(n * n * sizeof(double)) : NULL;
        if (g && !c) log_error("host copy of the cost matrix: out of memory (%zu bytes)", n * n * sizeof(double));
        if (c) {
            const int rc = tspgpu_get_costs(g, c);
            if (rc) { log_error("tspgpu_get_costs: %s", tspgpu_last_error(g)); free(c); c = NULL; }
        }
        if (c) __atomic_store_n(&tsp_inst.costs, c, __ATOMIC_RELEASE);
    }
    pthread_mutex_unlock(&host_costs_mu);
    return tsp_inst.costs;
}

/* one weight on the host with the arithmetic of tsp.c:629 (and TSPLIB's for ATT / CEIL_2D) */
static double host_weight(int i, int j)
{
    if (i == j) return -1.0;
    const double dx = tsp_inst.points[j].x - tsp_inst.points[i].x, dy = tsp_inst.points[j].y - tsp_inst.points[i].y;
    const double sq = dx * dx + dy * dy;
    if (tsp_edge_weight_kind == TSPGPU_EUC_2D) return (double)((int)(sqrtf((float)sq) + 0.5));
    if (tsp_edge_weight_kind == TSPGPU_ATT) { const double r = sqrt(sq / 10.0), t = (double)(long)(r + 0.5); return t < r ? t + 1.0 : t; }
    return ceil(sqrt(sq));
}

/* tsp.c:638-640.  While no host copy exists (matrix-free instances; the lazy mode of the `tsp` executable) the weight is
 * computed on demand with the reference's arithmetic -- bit-identical to the device matrix (tests) -- instead of downloading
 * n x n doubles to serve an O(n) lookup (ref_2opt's past-deadline branch, tsp_solution_cost: ADVICE r3). */
double tsp_get_cost(int i, int j)
{
    const double *c = __atomic_load_n(&tsp_inst.costs, __ATOMIC_ACQUIRE);
    return c ? c[(size_t)i * tsp_inst.nnodes + j] : host_weight(i, j);
}

/* tsp.c:687-728 */
bool tsp_is_tour(int path[], int n)
{
    if (n == 0) return false;
    unsigned char *seen = (unsigned char *)calloc((size_t)n, 1);
    int v = 0, steps = 0;
    bool ok = true;
    seen[0] = 1;
    for (;;) {
        const int nx = path[v];
        if (nx < 0 || nx >= n) { ok = false; break; }
        if (seen[nx]) { seen[nx] = 1; break; }
        seen[nx] = 1;
        v = nx;
        if (++steps > n) { ok = false; break; }
    }
    for (int i = 0; ok && i < n; i++) if (!seen[i]) ok = false;
    free(seen);
    return ok;
}

/* tsp.c:642-667 */
bool tsp_validate_solution(int nnodes, int *path)
{
    int *hits = (int *)calloc((size_t)nnodes, sizeof(int));
    bool ok = true;
    for (int i = 0; i < nnodes && ok; i++) {
        if (path[i] < 0 || path[i] >= nnodes) ok = false; else hits[path[i]]++;
    }
    for (int i = 0; i < nnodes && ok; i++) if (hits[i] != 1) ok = false;
    free(hits);
    return ok && tsp_is_tour(path, nnodes);
}

/* tsp.c:669-684: strict <, T_OK / CANCELLED / INVALID_ARGUMENT */
ERROR_CODE tsp_update_best_solution(tsp_solution *cur)
{
    if (!tsp_validate_solution(tsp_inst.nnodes, cur->path)) {
        log_error("You tried to update best_solution with an unvalid solution");
        return INVALID_ARGUMENT;
    }
    if (cur->cost < tsp_inst.best_solution.cost) {
        memcpy(tsp_inst.best_solution.path, cur->path, (size_t)tsp_inst.nnodes * sizeof(int));
        tsp_inst.best_solution.cost = cur->cost;
        log_info("new best solution: %f", cur->cost);
        return T_OK;
    }
    log_debug("discarded cost: %.2f", cur->cost);
    return CANCELLED;
}

/* tsp.c:730-736 */
double tsp_solution_cost(int path[])
{
    double c = 0;
    for (int i = 0; i < tsp_inst.nnodes; i++) c += tsp_get_cost(i, path[i]);
    return c;
}

void tsp_free_instance(void)
{
    utils_safe_free(tsp_env.inputfile);
    utils_safe_free(tsp_inst.points);
    utils_safe_free(tsp_inst.costs);
    utils_safe_free(tsp_inst.best_solution.path);
    utils_safe_free(tsp_inst.best_solution.comp);
    utils_safe_free(tsp_inst.threads_seeds);
    tsp_gpu_release();
}

void tsp_handlefatal(void)
{
    log_warn("fatal error detected, shutting down application");
    tsp_free_instance();
    exit(EXIT_FAILURE);
}
