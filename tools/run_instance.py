#!/usr/bin/env python3
"""Runs the single-tour descent NN(0) -> 2-opt local optimum `steps` times on a TSPLIB instance of tests/golden/data (or a
uniform-random instance: a number) with the engine's defaults, prints one JSON line; the program the rocprofv3 passes of
tools/collect_r03.sh profile.    python3 tools/run_instance.py fnl4461 5"""
import json, os, sys, time
sys.path.insert(0, ".")
from bench import read_tsplib, reference_points, draw_points, DATA
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if name.isdigit():
    draw_points([(int(name), 123)])
    xy = reference_points(int(name), 123)
else:
    xy = read_tsplib(os.path.join(DATA, name + ".tsp"))[0]
import travellingsalesmanoptimization_amd as T
eng = T.Engine(0)
eng.set_points(xy); eng.build_costs()
eng.tour_nn(0, 0)
eng.tour_copy(1, 0); eng.tour_two_opt(1)
ts = []
for _ in range(steps):
    eng.tour_copy(1, 0); eng.tour_store(1, want_path=False)
    t0 = time.perf_counter(); sw, rc = eng.tour_two_opt(1); ts.append(time.perf_counter() - t0)
_, cost, _ = eng.tour_store(1, want_path=False)
i = eng.info()
n = len(xy)
print(json.dumps({"instance": name, "n": n, "steps": steps, "sweeps_per_step": int(sw), "final_cost": cost, "ms_per_step": 1e3 * min(ts),
                  "us_per_sweep": 1e6 * min(ts) / sw, "evals_per_s": T.evals_per_sweep(n) * sw / min(ts), "persist": i["persist"],
                  "persist_window": i["persist_window"], "descent_launches_incl_warmup": steps + 1}))
eng.close()
