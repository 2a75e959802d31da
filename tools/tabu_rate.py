#!/usr/bin/env python3
"""Tabu search iteration rate (device-resident walk): python tools/tabu_rate.py [n] [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from bench import reference_points
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
eng = T.Engine(0)
eng.set_points(reference_points(n, 123)); eng.build_costs()
succ, cost = eng.nn_tour(0)
eng.tabu_search(succ.copy(), cost, 32)
t0 = time.perf_counter()
out = eng.tabu_search(succ.copy(), cost, k)
dt = time.perf_counter() - t0
i = eng.info()
print(f"n={n} k={k}: {dt*1e3:.1f} ms, {dt/k*1e6:.1f} us / iteration, {k*T.evals_per_sweep(n)/dt/1e9:.1f} Gevals/s; kernel={i['kernel']} elem={i['elem']} best={out[0] if isinstance(out, tuple) else out}")
eng.close()
