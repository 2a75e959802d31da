#!/usr/bin/env python3
"""Short descents, the pattern of mh_VNS (metaheuristic.c:251-341: kick, then ref_2opt to the next local optimum, ~5 sweeps):
time per tspgpu_two_opt call (path upload, descent, path download) with the LDS-resident kernel and with one launch per
sweep.  The kick here is a random segment insertion on the host (a timing probe: it does not have to be the
reference's vns_kick).  usage: python tools/vns_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import travellingsalesmanoptimization_amd as T
from bench import read_tsplib, reference_points


def kick(succ, rng):
    """three random tour edges (A,sA) (B,sB) (C,sC) in tour order -> A->sB ... C->sA ... B->sC (an or-3opt segment move)"""
    n = len(succ)
    order = np.empty(n, dtype=np.int64)
    v = 0
    for p in range(n):
        order[p] = v; v = succ[v]
    i, j, k = sorted(rng.choice(n - 1, size=3, replace=False))
    if j == i + 1 or k == j + 1:
        return
    A, sA, B, sB, C, sC = order[i], order[i + 1], order[j], order[j + 1], order[k], order[k + 1]
    succ[A] = sB; succ[C] = sA; succ[B] = sC


for name, xy in (("pr1002", read_tsplib(os.path.join(ROOT, "tests", "golden", "data", "pr1002.tsp"))[0]), ("n4096", reference_points(4096, 123))):
    eng = T.Engine(0)
    eng.set_points(xy); eng.build_costs()
    for mode in (0, 1):
        eng.set_option(T.OPT_PERSIST, mode)
        succ, cost = eng.nn_tour(0)
        cost, sw, rc = eng.two_opt(succ)
        rng = np.random.default_rng(7)
        tt = 0.0; tot = 0
        for it in range(60):
            kick(succ, rng)
            t0 = time.perf_counter()
            cost, sw, rc = eng.two_opt(succ)
            tt += time.perf_counter() - t0; tot += sw
        print(f"{name} persist={mode}: 60 kicks, {tot} sweeps, {tt*1e3:.2f} ms in two_opt = {tt/60*1e6:.1f} us per call, {tt/tot*1e6:.1f} us per sweep, final {cost:.0f}, used={eng.info()['persist']}", flush=True)
    eng.close()
