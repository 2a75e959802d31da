#!/usr/bin/env python3
"""All-NN + 2-opt (h_greedy_2opt) throughput on one GPU: starts batched on the device.
usage: python tools/multistart_rate.py [instance|nNNNN] [nstarts] [max_tours]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from travellingsalesmanoptimization_amd import tsplib
from bench import reference_points

what = sys.argv[1] if len(sys.argv) > 1 else "pr1002"
if what.startswith("n") and what[1:].isdigit():
    xy, kind = reference_points(int(what[1:]), 123), T.EUC_2D
else:
    xy, kind = tsplib.read(os.path.join(ROOT, "tests", "golden", "data", what + ".tsp"))
n = len(xy)
nstarts = int(sys.argv[2]) if len(sys.argv) > 2 else n
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
eng = T.Engine(0)
eng.set_points(xy, kind); eng.build_costs()
eng.set_option(T.OPT_MAX_TOURS, cap)
starts = np.arange(nstarts, dtype=np.int32)
eng.multistart_nn_2opt(starts[:min(8, nstarts)])          # warm (plans, graphs)
t0 = time.perf_counter()
res = eng.multistart_nn_2opt(starts)
dt = time.perf_counter() - t0
ev = res["sweeps"] * T.evals_per_sweep(n)
i = eng.info()
print(f"{what}: n={n} starts={nstarts} tours_in_flight<={cap} elem={i['elem']} kernel={i['kernel']} wgs/tour={i['wgs_per_tour']} block={i['block']} "
      f"best={res['cost']:.0f} (start {res['start']}) sweeps={res['sweeps']} time={dt:.3f}s  {ev/dt/1e9:.1f} Gevals/s", flush=True)
eng.close()
