/* tsp_main.c -- the `tsp` executable: same flow as src/main.c:158-192 (parse, read or
 * generate, banner, start the clock AFTER the cost matrix, run, print, free). */
#include "tsp_model.h"

#include <stdlib.h>

int main(int argc, char *argv[])
{
    /* this executable runs the heuristic path only: nothing on the host reads the n x n matrix unless asked to */
    const char *eager = getenv("TSP_EAGER_COSTS");
    tsp_lazy_costs = !(eager && atoi(eager) != 0);
    ERROR_CODE e = tsp_parse_commandline(argc, argv);
    if (!err_ok(e)) {
        log_error("error in command line parsing, error code: %d", e);
        tsp_free_instance();
        return EXIT_FAILURE;
    }
    if (tsp_env.graph_input) tsp_read_input();
    if (tsp_env.graph_random) tsp_generate_randompoints();
    if (!tsp_env.graph_input && !tsp_env.graph_random) {
        log_error("no instance: use -f <file> or -n <nodes>");
        return EXIT_FAILURE;
    }
    err_setinfo(tsp_inst.alg, tsp_inst.nnodes, tsp_env.graph_random, tsp_env.inputfile, tsp_env.timelimit, tsp_env.seed,
                tsp_env.policy, tsp_env.mileage_init, tsp_env.init_mip, tsp_env.skip_policy,
                tsp_env.callback_relaxation, tsp_env.lb_improv, tsp_env.lb_delta, tsp_env.lb_kstar);

    utils_startclock(&tsp_inst.c);  /* measures only algorithm time */
    e = tsp_run_algorithm();
    if (!err_ok(e)) {
        log_warn("error detected, shutting down application");
        tsp_free_instance();
        return EXIT_FAILURE;
    }
    const double elapsed = utils_timeelapsed(&tsp_inst.c);
    err_printoutput(tsp_inst.best_solution.cost, elapsed, tsp_inst.alg);
    tsp_free_instance();
    return EXIT_SUCCESS;
}
