#!/usr/bin/env python3
"""Short descents, the pattern of mh_VNS (metaheuristic.c:251-341: kick, then ref_2opt to the next local optimum, ~5 sweeps):
time per tspgpu_two_opt call (path upload, descent, path download) with the LDS-resident kernel and with one launch per
sweep.  usage: python tools/vns_probe.py   (needs the oracle's vns_kick: a tool, not part of the product)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import travellingsalesmanoptimization_amd as T
import oracle as O
from bench import read_tsplib, reference_points
for name, xy in (("pr1002", read_tsplib(os.path.join(ROOT, "tests", "golden", "data", "pr1002.tsp"))[0]), ("n4096", reference_points(4096, 123))):
    eng = T.Engine(0)
    eng.set_points(xy); eng.build_costs()
    for mode in (0, 1):
        eng.set_option(T.OPT_PERSIST, mode)
        succ, cost = eng.nn_tour(0)
        cost, sw, rc = eng.two_opt(succ)
        O.libc_srand(7)
        tt = 0.0; tot = 0
        for it in range(60):
            O.vns_kick(succ)
            t0 = time.perf_counter()
            cost, sw, rc = eng.two_opt(succ)
            tt += time.perf_counter() - t0; tot += sw
        print(f"{name} persist={mode}: 60 kicks, {tot} sweeps, {tt*1e3:.2f} ms in two_opt = {tt/60*1e6:.1f} us per call, {tt/tot*1e6:.1f} us per sweep, final {cost:.0f}, used={eng.info()['persist']}", flush=True)
    eng.close()
