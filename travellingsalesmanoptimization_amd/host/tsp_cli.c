/*
 * tsp_cli.c -- command line of the `tsp` binary.  Grammar, defaults, accepted values and
 * quirks are those of the reference (src/tsp.c:46-466) -- every flag is kept, none is added
 * that changes behaviour; the parser itself is table driven.
 *
 * Quirks kept on purpose ("Config / flags"): -skip stores into tsp_env.k
 * (tsp.c:237); -lb_initk/-lb_delta validate with the other's message; an unknown token
 * turns --help on; -n after -f is ignored and vice versa.
 */
#include "tsp_model.h"

#include <stdlib.h>
#include <string.h>

enum opt_id {
    O_FILE, O_TIME, O_SEED, O_ALG, O_N, O_TOFILE, O_NOPATCH, O_INITMIP, O_SKIP, O_NORELAX, O_MODCOSTS, O_HFPROB,
    O_LBDYNK, O_LBINITK, O_LBDELTA, O_LBIMPROV, O_LBKSTAR, O_K, O_EM, O_Q, O_V, O_VV, O_HELP, O_ALLALGS
};

static const struct { const char *name; enum opt_id id; bool has_value; } k_opts[] = {
    {"-f", O_FILE, true}, {"-file", O_FILE, true}, {"-t", O_TIME, true}, {"-time", O_TIME, true},
    {"-seed", O_SEED, true}, {"-alg", O_ALG, true}, {"-n", O_N, true}, {"--to_file", O_TOFILE, false},
    {"--no_patching", O_NOPATCH, false}, {"--init_mip", O_INITMIP, false}, {"-skip", O_SKIP, true},
    {"--no_relax", O_NORELAX, false}, {"--modify_costs", O_MODCOSTS, false}, {"-hf_prob", O_HFPROB, true},
    {"--lb_dynk", O_LBDYNK, false}, {"-lb_initk", O_LBINITK, true}, {"-lb_delta", O_LBDELTA, true},
    {"-lb_improv", O_LBIMPROV, true}, {"--lb_kstar", O_LBKSTAR, false}, {"-k", O_K, true}, {"-em", O_EM, true},
    {"-q", O_Q, false}, {"-v", O_V, false}, {"-vv", O_VV, false}, {"-h", O_HELP, false}, {"-help", O_HELP, false},
    {"--help", O_HELP, false}, {"--all_algs", O_ALLALGS, false}};

static const struct { const char *name; algorithms alg; } k_algs[] = {
    {"GREEDY", ALG_GREEDY}, {"GREEDY_ITER", ALG_GREEDY_ITER}, {"2OPT_GREEDY", ALG_2OPT_GREEDY},
    {"TABU_SEARCH", ALG_TABU_SEARCH}, {"VNS", ALG_VNS}, {"CPLEX_NOSEC", ALG_CX_NOSEC},
    {"CPLEX_BENDERS", ALG_CX_BENDERS}, {"EXTRA_MILEAGE", ALG_EXTRAMILEAGE},
    {"CPLEX_BRANCH_CUT", ALG_CX_BRANCH_AND_CUT}, {"HARD_FIXING", ALG_HARD_FIXING},
    {"LOCAL_BRANCHING", ALG_LOCAL_BRANCHING}};

static void usage(void)
{
    puts("tsp - Traveling Salesman Solver (MI355X 2-opt engine behind the heuristic path)\n");
    puts("USAGE:");
    puts("tsp [--help, -help, -h] [--all_algs] [-file, -f <path>] [-time, -t <value>] [-seed <value>] [-alg <option>] [-n <value>] [--to_file]");
    puts("    [-k <value>] [-em <option>] [--init_mip] [-skip <option>] [--no_relax] [-q, (DEFAULT), -v, -vv]\n");
    puts("OPTIONS:");
    puts("    --help, -help, -h       prints this text");
    puts("    --all_algs              prints all possible algorithms");
    puts("    -file, -f <path>        input a TSPLIB file format");
    puts("    -time, -t <value>       execution time limit in seconds");
    puts("    -seed <value>           seed for random generation");
    puts("    -alg <option>           selects the algorithm (see --all_algs)");
    puts("    -n <value>              number of nodes of a random instance");
    puts("    -k <value>              iterations of Tabu Search / VNS");
    puts("    --to_file, -em, --no_patching, --init_mip, -skip, --no_relax, --modify_costs, -hf_prob,");
    puts("    --lb_dynk, -lb_initk, -lb_delta, -lb_improv, --lb_kstar   accepted as in the reference");
    puts("    -q / -v / -vv           quiet (only the result line) / verbose / very verbose");
}

static void list_algs(void)
{
    puts("Available algorithms:");
    for (size_t i = 0; i < sizeof k_algs / sizeof k_algs[0]; i++) printf("    - %s\n", k_algs[i].name);
}

ERROR_CODE tsp_parse_commandline(int argc, char **argv)
{
    if (argc < 2) {
        printf("Type %s --help to see the full list of commands\n", argv[0]);
        exit(EXIT_FAILURE);
    }
    tsp_init();
    bool help = false, algs = false;

    for (int i = 1; i < argc; i++) {
        int hit = -1;
        for (size_t o = 0; o < sizeof k_opts / sizeof k_opts[0]; o++)
            if (!strcmp(k_opts[o].name, argv[i])) { hit = (int)o; break; }
        if (hit < 0) { help = true; continue; }
        const char *val = NULL;
        if (k_opts[hit].has_value) {
            if (i + 1 >= argc) { help = true; log_warn("invalid input"); continue; }
            val = argv[++i];
        }
        switch (k_opts[hit].id) {
        case O_FILE:
            if (tsp_env.graph_random) { log_error("ignoring input file, random graphs will be used"); break; }
            if (!utils_file_exists(val)) { log_fatal("file does not exist"); tsp_handlefatal(); }
            free(tsp_env.inputfile);
            tsp_env.inputfile = strdup(val);
            tsp_env.graph_input = true;
            break;
        case O_TIME: {
            const double t = atof(val);
            if (t < 0) log_warn("time cannot be negative, ignoring time limit"); else tsp_env.timelimit = t;
            break;
        }
        case O_SEED: tsp_env.seed = atoi(val); break;
        case O_ALG: {
            bool known = false;
            for (size_t a = 0; a < sizeof k_algs / sizeof k_algs[0]; a++)
                if (!strcmp(k_algs[a].name, val)) { tsp_inst.alg = k_algs[a].alg; known = true; }
            if (!known) log_warn("algorithm not recognized, using greedy as default");
            break;
        }
        case O_N: {
            const int n = atoi(val);
            if (n <= 0) { log_fatal("number of nodes should be greater than 0"); tsp_handlefatal(); }
            if (tsp_env.graph_input) { log_warn("ignoring number of nodes, graph from input file will be used"); break; }
            tsp_inst.nnodes = n;
            free(tsp_env.inputfile);
            tsp_env.inputfile = strdup("random");
            tsp_env.graph_random = true;
            break;
        }
        case O_TOFILE: tsp_env.tofile = true; break;
        case O_NOPATCH: tsp_env.bl_patching = false; break;
        case O_INITMIP: tsp_env.init_mip = true; break;
        case O_SKIP: {
            const int v = atoi(val);
            if (v < 0 || v > 2) log_info("supported options are 0 (thread seeds), 1 (number of nodes), 2 (depth>3)");
            else tsp_env.k = v; /* sic: tsp.c:237 */
            break;
        }
        case O_NORELAX: tsp_env.callback_relaxation = false; break;
        case O_MODCOSTS: tsp_env.modified_costs = true; break;
        case O_HFPROB: {
            const double p = atof(val);
            if (p <= 0.0 || p > 1.0) log_warn("hf_prob must be (0,1]"); else tsp_env.hf_prob = p;
            break;
        }
        case O_LBDYNK: tsp_env.lb_dynk = true; break;
        case O_LBINITK: {
            const int p = atoi(val);
            if (p < 10) log_warn("lb_delta must be >5"); else tsp_env.lb_initk = p;
            break;
        }
        case O_LBDELTA: {
            const int p = atoi(val);
            if (p < 5) log_warn("lb_delta must be >5"); else tsp_env.lb_delta = p;
            break;
        }
        case O_LBIMPROV: {
            const double p = atof(val);
            if (p <= 0.0 || p > 1.0) log_warn("lb_improv must be (0,1]"); else tsp_env.lb_improv = p;
            break;
        }
        case O_LBKSTAR: tsp_env.lb_kstar = true; break;
        case O_K: tsp_env.k = atoi(val); break;
        case O_EM:
            if (!strcmp(val, "RANDOM")) tsp_env.mileage_init = EM_RANDOM;
            else { if (strcmp(val, "MAX")) log_warn("initialization method not recognized, using MAX as default"); tsp_env.mileage_init = EM_MAX; }
            break;
        case O_Q: err_setverbosity(QUIET); break;
        case O_V: err_setverbosity(VERBOSE); break;
        case O_VV: err_setverbosity(VERY_VERBOSE); break;
        case O_HELP: help = true; break;
        case O_ALLALGS: algs = true; break;
        }
    }
    if (help) { usage(); exit(EXIT_SUCCESS); }
    if (algs) { list_algs(); exit(EXIT_SUCCESS); }
    return T_OK;
}
