#!/usr/bin/env python3
"""Batched multi-start with the one-launch-per-sweep path forced on / off.
usage: python tools/batch_fused_probe.py [instance|nNNNN] [nstarts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from travellingsalesmanoptimization_amd import tsplib
from bench import reference_points
what = sys.argv[1] if len(sys.argv) > 1 else "n4096"
if what.startswith("n") and what[1:].isdigit():
    xy, kind = reference_points(int(what[1:]), 123), T.EUC_2D
else:
    xy, kind = tsplib.read(os.path.join(ROOT, "tests", "golden", "data", what + ".tsp"))
n = len(xy)
nstarts = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for fused in (0, 2):
    eng = T.Engine(0)
    eng.set_points(xy, kind); eng.build_costs()
    eng.set_option(T.OPT_FUSED, fused)
    starts = np.arange(nstarts, dtype=np.int32)
    eng.multistart_nn_2opt(starts[:min(8, nstarts)])
    t0 = time.perf_counter()
    res = eng.multistart_nn_2opt(starts)
    dt = time.perf_counter() - t0
    ev = res["sweeps"] * T.evals_per_sweep(n)
    print(f"{what} starts={nstarts} fused={fused}: best={res['cost']:.0f} sweeps={res['sweeps']} {dt*1e3:.1f} ms {ev/dt/1e9:.1f} Gevals/s", flush=True)
    eng.close()
