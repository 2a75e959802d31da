#!/usr/bin/env python3
"""Tabu walk rate (mh_TabuSearch's loop, tspgpu_tabu_search): microseconds per iteration with the LDS-resident kernel
(k_lds2opt<., true>, TSPGPU_OPT_PERSIST = 1) and with the sweep + apply kernels, from the 2-opt local optimum of NN(0).
usage: python tools/tabu_probe.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from bench import reference_points

ns = [int(a) for a in sys.argv[1:]] or [1024, 2048, 3584, 4096]
eng = T.Engine(0)
for n in ns:
    eng.set_option(T.OPT_ELEM, T.ELEM_U16)
    eng.set_points(reference_points(n, 123)); eng.build_costs()
    seed, cost = eng.nn_tour(0)
    cost, _, _ = eng.two_opt(seed)
    k = 3000
    for mode in (0, 1):
        eng.set_option(T.OPT_PERSIST, mode)
        ts = []
        for rep in range(3):
            s = seed.copy()
            t0 = time.perf_counter()
            best, bc, final, _ = eng.tabu_search(s, cost, k)
            ts.append(time.perf_counter() - t0)
        dt = min(ts)
        print(f"n={n} persist={mode} used={eng.info()['persist']}: {k} iterations in {dt*1e3:.2f} ms = {dt/k*1e6:.2f} us per iteration; "
              f"best {bc:.0f} final {final:.0f}", flush=True)
eng.close()
