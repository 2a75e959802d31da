/*
 * oracle/cpu_ref.c -- CPU restatement of the reference's heuristic hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cpu_ref.h).  Plain C, single-threaded, the
 * same arithmetic, candidate order and tie-breaks as the reference so that it
 * can serve both as the bit-exact checker for the HIP path and as the timed
 * "reference CPU 2-opt" baseline.  Build: oracle/Makefile (gcc -O3 -std=gnu99
 * -ffp-contract=off, the reference's own optimisation level, Makefile:65,79-88).
 *
 * Parity: pinned against oracle/_ref (the reference sources compiled as they
 * lie) through tests/golden/*.json -- see tests/test_oracle_golden.py.
 */
#include "cpu_ref.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS (-1.0E-7)      /* src/tsp.h:19   EPSILON */
#define ORC_UNLINKED (-1.0f)   /* src/utils/utils.h:35 NOT_CONNECTED */

/* ------------------------------------------------------------------ K1 --- */

/* One matrix entry.  EUC_2D follows src/tsp.c:629 to the letter: the squared
 * length is a double, it is narrowed to float for sqrtf, the root is widened
 * again before the +0.5 and the truncation.  ATT / CEIL_2D are not in the
 * reference (src/tsp.c:576-584 rejects them); they follow TSPLIB 95 in double. */
static double edge_weight(double ax, double ay, double bx, double by, int kind)
{
    double dx = bx - ax, dy = by - ay;
    double sq = dx * dx + dy * dy;
    if (kind == ORC_EUC_2D) {
        float root = sqrtf((float)sq);
        return (double)((int)((double)root + 0.5));
    }
    if (kind == ORC_ATT) {
        double r = sqrt(sq / 10.0);
        double t = (double)(long)(r + 0.5);
        return t < r ? t + 1.0 : t;
    }
    return ceil(sqrt(sq)); /* CEIL_2D */
}

int orc_cost_matrix(const double *xy, int n, int kind, double *out)
{
    if (n <= 0 || !xy || !out) return 3; /* INVALID_ARGUMENT */
    for (long i = 0; i < n; i++) {
        double *row = out + i * (long)n;
        for (int j = 0; j < n; j++)
            row[j] = (j == i) ? -1.0
                              : edge_weight(xy[2 * i], xy[2 * i + 1],
                                            xy[2 * j], xy[2 * j + 1], kind);
    }
    return 0;
}

/* selected rows of the same matrix (for instances whose full matrix is GBs) */
int orc_cost_rows(const double *xy, int n, int kind, const int *rows, int nrows, double *out)
{
    if (n <= 0 || !xy || !out || !rows) return 3;
    for (int r = 0; r < nrows; r++) {
        int i = rows[r];
        if (i < 0 || i >= n) return 3;
        for (int j = 0; j < n; j++)
            out[(long)r * n + j] = (j == i) ? -1.0
                                            : edge_weight(xy[2 * i], xy[2 * i + 1],
                                                          xy[2 * j], xy[2 * j + 1], kind);
    }
    return 0;
}

void orc_random_points(int n, int seed, double *xy)
{
    srand((unsigned)seed);
    for (int i = 0; i < n; i++) {
        /* TSP_RAND, utils.h:26: x first, then y, node order */
        xy[2 * i] = ((double)rand() / RAND_MAX) * (5000 - (-5000)) + (-5000);
        xy[2 * i + 1] = ((double)rand() / RAND_MAX) * (5000 - (-5000)) + (-5000);
    }
}

/* ------------------------------------------------------------- K2/K4/K5 --- */

double orc_tour_cost(const double *c, int n, const int *succ)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += c[(long)i * n + succ[i]];
    return s;
}

void orc_reverse_path(int a, int sa, int b, int sb, int *prev, int *succ, int n)
{
    succ[a] = b;
    succ[sa] = sb;
    for (int cur = b;;) {           /* walk back from b until succ_a */
        int p = prev[cur];
        succ[cur] = p;
        cur = p;
        if (p == sa) break;
    }
    for (int k = 0; k < n; k++) prev[succ[k]] = k;
}

static void fill_prev(const int *succ, int n, int *prev)
{
    for (int i = 0; i < n; i++) prev[succ[i]] = i;
}

double orc_two_opt_once(const double *c, int n, int *succ, double *cost,
                        int *move_ab)
{
    double best = 0;
    int ba = -1, bb = -1;
    for (int a = 0; a < n - 1; a++) {
        const double *row_a = c + (long)a * n;
        for (int b = a + 1; b < n; b++) {
            int sa = succ[a], sb = succ[b];
            if (sa == sb || a == sb || b == sa) continue;
            double kept = row_a[sa] + c[(long)b * n + sb];
            double made = row_a[b] + c[(long)sa * n + sb];
            double d = made - kept;
            if (d < best) { best = d; ba = a; bb = b; }
        }
    }
    if (move_ab) { move_ab[0] = ba; move_ab[1] = bb; }
    if (best < ORC_EPS) {
        int *prev = (int *)malloc(sizeof(int) * (size_t)n);
        fill_prev(succ, n, prev);
        orc_reverse_path(ba, succ[ba], bb, succ[bb], prev, succ, n);
        *cost += best;
        free(prev);
    }
    return best;
}

long orc_two_opt(const double *c, int n, int *succ, double *cost, long max_sweeps)
{
    long sweeps = 0;
    double d;
    *cost = orc_tour_cost(c, n, succ);
    do {
        if (max_sweeps >= 0 && sweeps >= max_sweeps) break;
        d = orc_two_opt_once(c, n, succ, cost, NULL);
        sweeps++;
    } while (d < ORC_EPS);
    return sweeps;
}

/* ------------------------------------------------------------------ K6 --- */

int orc_nn_tour(const double *c, int n, int start, int *succ, double *cost)
{
    if (!c) return 13;                      /* INTERNAL, heuristics.c:218-221 */
    if (start < 0 || start >= n) return 14; /* UNAVAILABLE, :223-226 */
    unsigned char *seen = (unsigned char *)calloc((size_t)n, 1);
    int cur = start;
    double total = 0;
    seen[cur] = 1;
    for (;;) {
        const double *row = c + (long)cur * n;
        int arg = -1;
        double lo = DBL_MAX;
        for (int i = 0; i < n; i++) {
            if (i == cur || seen[i]) continue;
            double w = row[i];
            if (w != ORC_UNLINKED && w < lo) { lo = w; arg = i; }
        }
        if (arg < 0) { succ[cur] = start; break; }
        succ[cur] = arg;
        seen[arg] = 1;
        total += lo;
        cur = arg;
    }
    total += c[(long)cur * n + start];
    *cost = total;
    free(seen);
    return 0;
}

int orc_nn_all(const double *c, int n, const int *starts, int nstarts,
               int *best_succ, double *best_cost, int *best_start)
{
    int *succ = (int *)malloc(sizeof(int) * (size_t)n);
    double best = DBL_MAX;
    int arg = -1;
    for (int k = 0; k < nstarts; k++) {
        int s = starts ? starts[k] : k;
        double cost;
        if (orc_nn_tour(c, n, s, succ, &cost)) continue;
        if (cost < best) {
            best = cost; arg = s;
            memcpy(best_succ, succ, sizeof(int) * (size_t)n);
        }
    }
    free(succ);
    *best_cost = best;
    *best_start = arg;
    return 0;
}

int orc_multistart_nn_2opt(const double *c, int n, const int *starts,
                           int nstarts, int *best_succ, double *best_cost,
                           int *best_start, long *total_sweeps)
{
    int *succ = (int *)malloc(sizeof(int) * (size_t)n);
    double best = DBL_MAX;
    int arg = -1;
    long sweeps = 0;
    for (int k = 0; k < nstarts; k++) {
        int s = starts ? starts[k] : k;
        double cost;
        if (orc_nn_tour(c, n, s, succ, &cost)) break;
        sweeps += orc_two_opt(c, n, succ, &cost, -1);
        if (orc_valid_tour(succ, n) && cost < best) {
            best = cost; arg = s;
            memcpy(best_succ, succ, sizeof(int) * (size_t)n);
        }
    }
    free(succ);
    *best_cost = best;
    *best_start = arg;
    if (total_sweeps) *total_sweeps = sweeps;
    return 0;
}

/* ------------------------------------------------------------------ K3 --- */

static int is_tabu(const int *tl, int node, int iter, int tenure)
{
    return iter - tl[node] < tenure && tl[node] != -1; /* metaheuristic.c:416-418 */
}

int orc_tabu_move(const double *c, int n, int *succ, double *cost,
                  int *tabu_list, int tenure, int iter, int *move_ab)
{
    double best = DBL_MAX;
    int ba = -1, bb = -1;
    for (int a = 0; a < n - 1; a++) {
        for (int b = a + 1; b < n; b++) {
            int sa = succ[a], sb = succ[b];
            if (sa == sb || a == sb || b == sa) continue;
            if (is_tabu(tabu_list, a, iter, tenure) || is_tabu(tabu_list, b, iter, tenure) ||
                is_tabu(tabu_list, sa, iter, tenure) || is_tabu(tabu_list, sb, iter, tenure))
                continue;
            double kept = c[(long)a * n + sa] + c[(long)b * n + sb];
            double made = c[(long)a * n + b] + c[(long)sa * n + sb];
            double d = made - kept;
            if (d < best) { best = d; ba = a; bb = b; }
        }
    }
    if (move_ab) { move_ab[0] = ba; move_ab[1] = bb; }
    if (best < DBL_MAX) {
        int sa = succ[ba], sb = succ[bb];
        int *prev = (int *)malloc(sizeof(int) * (size_t)n);
        fill_prev(succ, n, prev);
        orc_reverse_path(ba, sa, bb, sb, prev, succ, n);
        free(prev);
        *cost += best;
        tabu_list[ba] = iter; tabu_list[bb] = iter;
        tabu_list[sa] = iter; tabu_list[sb] = iter;
    }
    return 0;
}

int orc_tabu_search(const double *c, int n, int *succ, double *cost, int k,
                    int *best_succ, double *best_cost, double *trace)
{
    /* tabu_init, metaheuristic.c:65-84 (MIN_FRACTION .125, MAX_FRACTION .25) */
    int *tl = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) tl[i] = -1;
    int up = 1;
    int tenure = (int)(0.125 * n + 1);
    int t_max = (int)(0.25 * n), t_min = (int)(0.125 * n);

    double best = *cost;
    memcpy(best_succ, succ, sizeof(int) * (size_t)n);

    for (int it = 0; it < k; it++) {
        /* tabu_linear_policy, metaheuristic.c:40-59 */
        if (tenure == t_max || tenure == t_min) up = !up;
        tenure += up ? 1 : -1;
        orc_tabu_move(c, n, succ, cost, tl, tenure, it, NULL);
        if (orc_valid_tour(succ, n) && *cost < best) {
            best = *cost;
            memcpy(best_succ, succ, sizeof(int) * (size_t)n);
        }
        if (trace) trace[it] = *cost;
    }
    free(tl);
    *best_cost = best;
    return 0;
}

/* ----------------------------------------------------------------- VNS --- */

/* The reference probes tour[idx-1] and tour[idx+1] without wrapping
 * (metaheuristic.c:372).  On glibc the int before a calloc'd block is the high
 * half of the chunk size (0) and the int after n ints is calloc-zeroed padding
 * unless 4n+8 is a multiple of 16; 0 is what the compiled reference therefore
 * sees, and what this restatement feeds the comparison.  For n % 4 == 2 the
 * probe lands in the next chunk header: treated as "no match". */
static int probe(const int *tour, int n, int idx)
{
    if (idx < 0) return 0;
    if (idx >= n) return (n % 4 == 2) ? -2 : 0;
    return tour[idx];
}

int orc_vns_kick(int n, int *succ)
{
    int *tour = (int *)malloc(sizeof(int) * (size_t)n);
    for (int p = 0, node = 0; p < n; p++, node = succ[node]) tour[p] = node;

    int pick[3];
    for (int i = 0; i < 3; i++) {
        int r;
        do {
            r = rand() % n;
            for (int j = 0; j < i; j++) {
                if (r == pick[j] || r == probe(tour, n, pick[j] - 1) ||
                    r == probe(tour, n, pick[j] + 1)) { r = -1; break; }
            }
        } while (r == -1);
        pick[i] = r;
        for (int j = i; j > 0; j--)
            if (pick[j] < pick[j - 1]) { int t = pick[j]; pick[j] = pick[j - 1]; pick[j - 1] = t; }
    }
    int A = tour[pick[0]], sA = tour[(pick[0] + 1) % n];
    int B = tour[pick[1]], sB = tour[(pick[1] + 1) % n];
    int C = tour[pick[2]], sC = tour[(pick[2] + 1) % n];
    /* tabu_make_move case 7, metaheuristic.c:490-500 */
    succ[A] = sB;
    succ[C] = sA;
    succ[B] = sC;
    free(tour);
    return 0;
}

int orc_vns(const double *c, int n, int *succ, double *cost, int k,
            int *best_succ, double *best_cost)
{
    double best = *cost;
    memcpy(best_succ, succ, sizeof(int) * (size_t)n);
    for (int it = 0; it < k; it++) {
        orc_two_opt(c, n, succ, cost, -1);
        if (*cost < best) {
            best = *cost;
            memcpy(best_succ, succ, sizeof(int) * (size_t)n);
        }
        int r = rand() % (10 - 2 + 1) - 2;   /* UPPER 10, LOWER 2, :308 */
        for (int j = 0; j < r; j++) orc_vns_kick(n, succ);
    }
    *best_cost = best;
    return 0;
}

/* ------------------------------------------------ matrix-free variants --- */
/* Same algorithms with every c[i][j] recomputed from the coordinates (edge_weight above):
 * for instances whose n x n matrix does not fit (pla85900: 59 GB of doubles).  Results are
 * identical to the matrix versions by construction; tests check that on small instances. */

int orc_nn_tour_xy(const double *xy, int n, int kind, int start, int *succ, double *cost)
{
    if (start < 0 || start >= n) return 14;
    unsigned char *seen = (unsigned char *)calloc((size_t)n, 1);
    int cur = start;
    double total = 0;
    seen[cur] = 1;
    for (;;) {
        int arg = -1;
        double lo = DBL_MAX;
        const double cx = xy[2 * cur], cy = xy[2 * cur + 1];
        for (int i = 0; i < n; i++) {
            if (i == cur || seen[i]) continue;
            double w = edge_weight(cx, cy, xy[2 * i], xy[2 * i + 1], kind);
            if (w < lo) { lo = w; arg = i; }
        }
        if (arg < 0) { succ[cur] = start; break; }
        succ[cur] = arg;
        seen[arg] = 1;
        total += lo;
        cur = arg;
    }
    total += edge_weight(xy[2 * cur], xy[2 * cur + 1], xy[2 * start], xy[2 * start + 1], kind);
    *cost = total;
    free(seen);
    return 0;
}

double orc_tour_cost_xy(const double *xy, int n, int kind, const int *succ)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += edge_weight(xy[2 * i], xy[2 * i + 1], xy[2 * succ[i]], xy[2 * succ[i] + 1], kind);
    return s;
}

/* the scan of refinment.c:49-69 restricted to a in [a_lo, a_hi) -- read-only, so that a test can spread one sweep of a
 * large matrix-free instance over host threads (pla85900: 3.7e9 pairs); the first strictly smallest delta in (a asc, b asc)
 * order of the slice; the caller combines slices in ascending a with a strict < (= the sequential result) */
double orc_two_opt_scan_xy(const double *xy, int n, int kind, const int *succ, int a_lo, int a_hi, int *move_ab)
{
    double best = 0;
    int ba = -1, bb = -1;
    double *dn = (double *)malloc(sizeof(double) * (size_t)n);   /* c[b][succ b] */
    for (int b = 0; b < n; b++) dn[b] = edge_weight(xy[2 * b], xy[2 * b + 1], xy[2 * succ[b]], xy[2 * succ[b] + 1], kind);
    if (a_hi > n - 1) a_hi = n - 1;
    for (int a = a_lo < 0 ? 0 : a_lo; a < a_hi; a++) {
        const int sa = succ[a];
        const double ax = xy[2 * a], ay = xy[2 * a + 1], sx = xy[2 * sa], sy = xy[2 * sa + 1];
        for (int b = a + 1; b < n; b++) {
            const int sb = succ[b];
            if (sa == sb || a == sb || b == sa) continue;
            double kept = dn[a] + dn[b];
            double made = edge_weight(ax, ay, xy[2 * b], xy[2 * b + 1], kind) +
                          edge_weight(sx, sy, xy[2 * sb], xy[2 * sb + 1], kind);
            double d = made - kept;
            if (d < best) { best = d; ba = a; bb = b; }
        }
    }
    free(dn);
    if (move_ab) { move_ab[0] = ba; move_ab[1] = bb; }
    return best;
}

double orc_two_opt_once_xy(const double *xy, int n, int kind, int *succ, double *cost, int *move_ab)
{
    int ab[2];
    const double best = orc_two_opt_scan_xy(xy, n, kind, succ, 0, n - 1, ab);
    const int ba = ab[0], bb = ab[1];
    if (move_ab) { move_ab[0] = ba; move_ab[1] = bb; }
    if (best < ORC_EPS) {
        int *prev = (int *)malloc(sizeof(int) * (size_t)n);
        fill_prev(succ, n, prev);
        orc_reverse_path(ba, succ[ba], bb, succ[bb], prev, succ, n);
        *cost += best;
        free(prev);
    }
    return best;
}

/* ---------------------------------------------------------- validation --- */

int orc_valid_tour(const int *succ, int n)
{
    if (n <= 0) return 0;
    unsigned char *hit = (unsigned char *)calloc((size_t)n, 1);
    int ok = 1;
    for (int i = 0; i < n && ok; i++) {
        int v = succ[i];
        if (v < 0 || v >= n || hit[v]) ok = 0; else hit[v] = 1;
    }
    if (ok) { /* single cycle through node 0 */
        int cnt = 0, v = 0;
        do { v = succ[v]; cnt++; } while (v != 0 && cnt <= n);
        ok = (cnt == n);
    }
    free(hit);
    return ok;
}

uint64_t orc_fnv1a(const int *succ, int n)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int i = 0; i < n; i++) { h ^= (unsigned)succ[i]; h *= 0x100000001b3ULL; }
    return h;
}
