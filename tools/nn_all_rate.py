import sys, time
sys.path.insert(0, ".")
import numpy as np
from bench import read_tsplib, DATA
import os
import travellingsalesmanoptimization_amd as T
xy, ewt = read_tsplib(os.path.join(DATA, "pla85900.tsp"))
e = T.Engine(0)
e.set_points(xy, T.CEIL_2D); e.build_costs()
t0 = time.perf_counter()
best, c, s, done, rc = e.nn_all_timed(None, 20.0)
dt = time.perf_counter() - t0
print(f"nn_all_timed 20 s: done {done} starts, best {c} from {s}, rc {rc}, {dt:.1f} s -> {done/dt:.0f} starts/s -> all 85900 in {85900/(done/dt):.0f} s")
