#!/usr/bin/env python3
"""Matrix-free sweep rate: python tools/otf_rate.py [instance|nNNNN] [kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import travellingsalesmanoptimization_amd as T
from travellingsalesmanoptimization_amd import tsplib
from bench import reference_points
what = sys.argv[1] if len(sys.argv) > 1 else "pla85900"
if what.startswith("n") and what[1:].isdigit():
    xy, kind = reference_points(int(what[1:]), 123), T.EUC_2D
elif what.startswith("r") and what[1:].isdigit():          # uniform-random INTEGER points, CEIL_2D: pla85900's weight kind without its node order
    import numpy as np
    xy, kind = np.random.RandomState(5).randint(0, 1400000, size=(int(what[1:]), 2)).astype(np.float64), T.CEIL_2D
else:
    xy, kind = tsplib.read(os.path.join(ROOT, "tests", "golden", "data", what + ".tsp"))
n = len(xy)
for mf in (1, 0):
    eng = T.Engine(0)
    eng.set_option(T.OPT_MATRIX_FREE, mf)
    eng.set_points(xy, kind)
    try:
        eng.build_costs()
        eng.tour_nn(0, 0)
        reps = 20 if n < 20000 else 3
        if os.environ.get("OTF_FULL"):
            eng.set_option(99, 3)                      # every pair evaluated in full (the early-out off)
        ms = eng.time_sweep(0, reps)
        abl = {}
        if mf == 1 and os.environ.get("OTF_ABLATE"):
            for a in (3, 4, 5, 6):
                eng.set_option(99, a); abl[a] = round(eng.time_sweep(0, reps) * 1e3, 1)
            eng.set_option(99, 0)
            print("  option 99 (3: early-out off, 4: nothing behind the box tests, 5: nothing behind the pair tests, 6: nothing behind the run-level box test) us/sweep:", abl, flush=True)
        i = eng.info()
        print(f"{what} n={n} matrix_free={i['matrix_free']} kernel={i['kernel']} elem={i['elem']} wgs={i['wgs_per_tour']}: "
              f"{ms*1e3:10.1f} us / sweep  {T.evals_per_sweep(n)/ms/1e6:8.1f} Gevals/s", flush=True)
    except T.TspGpuError as e:
        print(f"{what} n={n} matrix_free={mf}: {e}", flush=True)
    eng.close()
