#!/usr/bin/env python3
"""Slow golden vectors from the REFERENCE's own code (oracle/_ref/libtspref.so), one JSON file
per case under tests/golden/ (authoring container only; needs /root/reference for the build):

    python oracle/make_golden_slow.py d18512_multistart   # ~4 min: BASELINE config 4's instance,
        NN + ref_2opt_once x 5 from starts 0..7 (the batched multi-start path at n = 18 512)
    python oracle/make_golden_slow.py n16384              # ~45 min: -n 16384 -seed 123, NN(0) to the
        2-opt local optimum (the large size of the throughput table; bench.py's parity gate)
    python oracle/make_golden_slow.py mod_costs_threads   # seconds: h_Greedy_2opt_mod_costs on four caller
        matrices per instance (the concurrency test of the host layer)

    python oracle/make_golden_slow.py n4096_multistart [count [procs]]   # ~10 min per 64 starts on 6 cores: -n 4096
        -seed 123 (the headline instance), NN + ref_2opt to the local optimum from starts 0..count-1 (gate of the
        batched multi-start legs of bench.py: 64 starts on one GPU, the 512-start job sharded over N); incremental

The outputs are DATA (costs, sweep counts, FNV-1a of successor arrays).
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O  # noqa: E402

ROOT = os.path.dirname(HERE)
DATA = os.path.join(ROOT, "tests", "golden", "data")


def d18512_multistart():
    ref = O.Reference()
    t0 = time.time()
    ref.read_file(os.path.join(DATA, "d18512.tsp"))
    print("matrix", time.time() - t0, flush=True)
    cap = 5
    starts = []
    for s in range(8):
        succ, nn_cost, _ = ref.nn(s)
        nn_fnv = O.fnv1a(succ)
        sweeps, cost, trace = ref.two_opt_counted(succ, cap, cap)
        starts.append({"start": s, "nn_cost": nn_cost, "nn_fnv": f"{nn_fnv:016x}", "sweeps": sweeps,
                       "cost": cost, "fnv": f"{O.fnv1a(succ):016x}", "trace": [float(x) for x in trace]})
        print(s, nn_cost, cost, time.time() - t0, flush=True)
    best = min(starts, key=lambda e: (e["cost"], e["start"]))   # strict <, ascending starts (tsp.c:671)
    out = {"_generator": "oracle/make_golden_slow.py d18512_multistart (reference compiled by oracle/Makefile)",
           "instance": "d18512", "n": ref.n, "max_sweeps": cap, "starts": starts,
           "best": {"start": best["start"], "cost": best["cost"], "fnv": best["fnv"]}}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "golden_d18512_multistart.json"), "w"), indent=1)


def n16384():
    ref = O.Reference()
    n, seed = 16384, 123
    t0 = time.time()
    ref.random(n, seed)
    print("matrix", time.time() - t0, flush=True)
    succ, nn_cost, _ = ref.nn(0)
    nn_fnv = O.fnv1a(succ)
    sweeps, cost, trace = ref.two_opt_counted(succ, -1, 16)
    out = {"_generator": "oracle/make_golden_slow.py n16384 (reference compiled by oracle/Makefile)",
           "n": n, "seed": seed, "two_opt": {"nn_cost": nn_cost, "nn_fnv": f"{nn_fnv:016x}", "sweeps": sweeps,
                                             "final_cost": cost, "final_fnv": f"{O.fnv1a(succ):016x}",
                                             "trace": [float(x) for x in trace]},
           "seconds": time.time() - t0}
    print(out, flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "golden_n16384_s123.json"), "w"), indent=1)


def _n4096_one(s):
    ref = O.Reference(scratch=os.path.join(HERE, "_ref", f"scratch_{s}"))
    ref.random(4096, 123)
    succ, nn_cost, _ = ref.nn(s)
    nn_fnv = O.fnv1a(succ)
    sweeps, cost, _ = ref.two_opt_counted(succ, -1, 0)
    return {"start": s, "nn_cost": nn_cost, "nn_fnv": f"{nn_fnv:016x}", "sweeps": sweeps, "cost": cost,
            "fnv": f"{O.fnv1a(succ):016x}"}


def n4096_multistart(count=64, procs=6):
    """h_greedy_2opt's loop body (heuristics.c:82-111) for starts 0..count-1 of the headline instance, one
    reference process per start (the reference keeps its instance in process-wide globals); starts already in
    the file are kept (the runs are independent), so the fixture can be extended"""
    import multiprocessing as mp
    t0 = time.time()
    path = os.path.join(ROOT, "tests", "golden", "golden_n4096_multistart.json")
    have = {e["start"]: e for e in json.load(open(path))["starts"]} if os.path.exists(path) else {}
    todo = [s for s in range(count) if s not in have]
    with mp.get_context("spawn").Pool(procs) as pool:
        for e in pool.imap_unordered(_n4096_one, todo, chunksize=1):
            have[e["start"]] = e
            print(e["start"], e["cost"], round(time.time() - t0), flush=True)
    starts = [have[s] for s in sorted(have)]
    best = min(starts, key=lambda e: (e["cost"], e["start"]))   # strict <, ascending starts (tsp.c:671)
    out = {"_generator": "oracle/make_golden_slow.py n4096_multistart (reference compiled by oracle/Makefile)",
           "n": 4096, "seed": 123, "starts": starts,
           "best": {"start": best["start"], "cost": best["cost"], "fnv": best["fnv"]},
           "total_sweeps": sum(e["sweeps"] for e in starts)}
    json.dump(out, open(path, "w"), indent=1)


def mod_costs_matrix(c, seed):
    """the shape cplex_model.c:1176-1258 feeds h_Greedy_2opt_mod_costs: c[i][j] * (1 - x*_ij), symmetric,
    non-integer, diagonal 0 (same construction as oracle/make_golden.py)"""
    n = c.shape[0]
    r = np.random.default_rng(seed)
    x = np.triu(r.random((n, n)), 1)
    x = x + x.T
    mc = c * (1.0 - x)
    np.fill_diagonal(mc, 0.0)
    return np.ascontiguousarray(mc)


def mod_costs_threads():
    """four different caller matrices on ONE instance (tsp_inst.nnodes is process-wide, so concurrent
    CPLEX callback threads always share n): the concurrency test of the host layer compares each
    thread's result with these (seconds of CPU)."""
    ref = O.Reference()
    out = {"_generator": "oracle/make_golden_slow.py mod_costs_threads (reference compiled by oracle/Makefile)", "cases": []}
    for name, seeds in [("kroA100", [21, 22, 23, 24]), ("n200_s3", [31, 32, 33, 34])]:
        if name.startswith("n"):
            ref.random(200, 3)
        else:
            ref.read_file(os.path.join(DATA, name + ".tsp"))
        c = ref.costs()
        for seed in seeds:
            succ, cost = ref.mod_costs(mod_costs_matrix(c, seed))
            out["cases"].append({"instance": name, "seed": seed, "cost_hex": float(cost).hex(), "cost": cost,
                                 "fnv": f"{O.fnv1a(succ):016x}"})
            print(name, seed, cost, flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "golden_mod_costs_threads.json"), "w"), indent=1)


if __name__ == "__main__":
    fn = {"d18512_multistart": d18512_multistart, "n16384": n16384, "mod_costs_threads": mod_costs_threads,
          "n4096_multistart": n4096_multistart}[sys.argv[1]]
    fn(*[int(a) for a in sys.argv[2:]])
