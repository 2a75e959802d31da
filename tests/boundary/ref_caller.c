/* A caller written against the REFERENCE's own headers (-I/root/reference/src: tsp.h, algorithms/refinment.h,
 * heuristics.h, metaheuristic.h, utils/utils.h, utils/errors.h) and linked with libtsphost.so -- what a maintainer of
 * the reference gets when the heuristic path is swapped for the MI355X engine.
 *   - _Static_assert: sizeof / offsetof of every struct of src/tsp.h:58-134 and src/utils/utils.h:37-47, the value of
 *     every enum constant of src/tsp.h:25-56 and src/utils/errors.h:33-51, against the host layer's (host_layout.h is
 *     generated from host/tsp_model.h by host_layout_gen.c);
 *   - every prototype of the path as a typed pointer initialised from the function the library exports;
 *   - at run time (no GPU needed): the library's globals are written and read through the reference's struct
 *     definitions -- tsp_init's defaults (src/tsp.c:6-44), a command line (src/tsp.c:46-466), validation and the
 *     incumbent rule (src/tsp.c:642-728, :669-684).
 * Test infrastructure; nothing here is copied from the reference: it only #includes its headers at build time. */
#include <limits.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsp.h"
#include "algorithms/refinment.h"
#include "algorithms/heuristics.h"
#include "algorithms/metaheuristic.h"

#include "host_layout.h"

#define S(t) _Static_assert(sizeof(t) == H_SIZEOF_##t, "sizeof(" #t ") differs from the host layer's");
#define F(t, f) _Static_assert(offsetof(t, f) == H_OFF_##t##_##f, "offsetof(" #t ", " #f ") differs from the host layer's");
#define V(x) _Static_assert((long)(x) == H_VAL_##x, #x " differs from the host layer's");
#include "layout_items.inc"
#undef S
#undef F
#undef V

#define P(ret, name, args) static ret (*const p_##name) args = name;
#include "protos.inc"
#undef P

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "ref_caller: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void)
{
    int nprotos = 0;
#define P(ret, name, args) nprotos += p_##name != 0;
#include "protos.inc"
#undef P
    /* tsp_init: src/tsp.c:6-44, read back through the reference's struct definitions */
    p_tsp_init();
    CHECK(tsp_env.timelimit == -1 && tsp_env.seed == -1 && tsp_env.k == INT_MAX && tsp_env.policy == POL_LINEAR);
    CHECK(tsp_env.bl_patching && tsp_env.init_mip && tsp_env.callback_relaxation && tsp_env.hf_prob == 0.7);
    CHECK(tsp_env.lb_initk == 10 && tsp_env.lb_improv == 0.02 && tsp_env.lb_delta == 10 && !tsp_env.lb_kstar);
    CHECK(tsp_inst.nnodes == -1 && tsp_inst.alg == ALG_GREEDY && tsp_inst.ncols == -1 && tsp_inst.costs == NULL);
    /* a command line: src/tsp.c:57-409 */
    char *argv[] = {"tsp", "-n", "12", "-seed", "7", "-alg", "VNS", "-t", "3.5", "-k", "40", "-q", NULL};
    CHECK(err_ok(p_tsp_parse_commandline(12, argv)));
    CHECK(tsp_inst.nnodes == 12 && tsp_env.seed == 7 && tsp_inst.alg == ALG_VNS && tsp_env.timelimit == 3.5 && tsp_env.k == 40);
    CHECK(tsp_env.graph_random && !tsp_env.graph_input);
    /* validation and the incumbent rule on host arrays: src/tsp.c:642-728, :669-684 */
    tsp_solution s;
    CHECK(err_ok(p_tsp_init_solution(12, &s)));
    for (int i = 0; i < 12; i++) s.path[i] = (i + 1) % 12;
    CHECK(p_tsp_validate_solution(12, s.path) && p_tsp_is_tour(s.path, 12));
    s.path[3] = 3;
    CHECK(!p_tsp_validate_solution(12, s.path));
    s.path[3] = 4;
    tsp_inst.best_solution.path = (int *)calloc(12, sizeof(int));
    tsp_inst.best_solution.cost = 100.0;
    s.cost = 100.0;
    CHECK(p_tsp_update_best_solution(&s) == CANCELLED);              /* strict < */
    s.cost = 99.0;
    CHECK(p_tsp_update_best_solution(&s) == T_OK && tsp_inst.best_solution.cost == 99.0);
    CHECK(memcmp(tsp_inst.best_solution.path, s.path, 12 * sizeof(int)) == 0);
    /* the host-side tabu helpers: metaheuristic.c:65-84, :416-418, :40-59 */
    tabu_search ts;
    CHECK(err_ok(p_tabu_init(&ts, 1000)));
    CHECK(ts.tenure == 126 && ts.max_tenure == 250 && ts.min_tenure == 125 && ts.increment && ts.tabu_list[999] == -1);
    ts.tabu_list[5] = 10;
    CHECK(p_is_in_tabu_list(&ts, 5, 100) && !p_is_in_tabu_list(&ts, 5, 136) && !p_is_in_tabu_list(&ts, 6, 11));
    tsp_env.policy = POL_LINEAR;
    CHECK(err_ok(p_tabu_linear_policy(&ts)) && ts.tenure == 127);
    p_tabu_free(&ts);
    /* ref_reverse_path on host arrays: refinment.c:95-114 */
    {
        int path[6] = {1, 2, 3, 4, 5, 0}, prev[6] = {5, 0, 1, 2, 3, 4};
        p_ref_reverse_path(0, 1, 3, 4, prev, path);                 /* 0 -> 3 -> 2 -> 1 -> 4 -> 5 -> 0 */
        const int want[6] = {3, 4, 1, 2, 5, 0};
        CHECK(memcmp(path, want, sizeof want) == 0 && prev[3] == 0 && prev[4] == 1);
    }
    printf("ref_caller ok: %d prototypes, layouts and enums equal\n", nprotos);
    return 0;
}
