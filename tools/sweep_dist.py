#!/usr/bin/env python3
"""Sweeps to the local optimum per start node (how uneven a multi-start batch is): python tools/sweep_dist.py [instance]
pr1002: mean 176.4, max 202, min 154 -> a batch that waits for its slowest tour runs at 87 % of a perfectly refilled one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from travellingsalesmanoptimization_amd import tsplib
what = sys.argv[1] if len(sys.argv) > 1 else "pr1002"
xy, kind = tsplib.read(os.path.join(ROOT, "tests", "golden", "data", what + ".tsp"))
eng = T.Engine(0); eng.set_points(xy, kind); eng.build_costs()
sw = []
for s in range(len(xy)):
    eng.tour_nn(0, s)
    k, _ = eng.tour_two_opt(0)
    sw.append(k)
sw = np.array(sw)
print(f"{what} sweeps per start: mean {sw.mean():.1f} max {sw.max()} min {sw.min()} p90 {np.percentile(sw, 90):.0f} mean/max {sw.mean()/sw.max():.3f}")
eng.close()
