#!/usr/bin/env python3
"""All-NN + 2-opt on one GPU (h_greedy_2opt): time of the whole batch against the sweep kernel, the workgroups per tour
and the block size the plan is given.  usage: python tools/tune_multistart.py [instance] [nstarts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from travellingsalesmanoptimization_amd import tsplib

what = sys.argv[1] if len(sys.argv) > 1 else "pr1002"
if what.startswith("n") and what[1:].isdigit():
    from bench import reference_points
    xy, kind = reference_points(int(what[1:]), 123), T.EUC_2D
else:
    xy, kind = tsplib.read(os.path.join(ROOT, "tests", "golden", "data", what + ".tsp"))
n = len(xy)
nstarts = int(sys.argv[2]) if len(sys.argv) > 2 else n
starts = np.arange(nstarts, dtype=np.int32)
eng = T.Engine(0)
eng.set_option(T.OPT_ELEM, int(os.environ.get("ELEM", "0")))
if os.environ.get("DEPTH"):
    eng.set_option(T.OPT_DEPTH, int(os.environ["DEPTH"]))      # rows in flight per thread in the streamed kernel
eng.set_points(xy, kind); eng.build_costs()
combos = [(0, 0, 0), (3, 0, 0), (2, 0, 0)] + [(2, max(1, n // p), 0) for p in (8, 12, 16, 24, 32, 48, 64, 128)]
if len(sys.argv) > 3:
    combos = [tuple(int(v) for v in c.split(",")) for c in sys.argv[3:]]
for kernel, wgs, block in combos:
    try:
        eng.set_option(T.OPT_KERNEL, kernel); eng.set_option(T.OPT_WGS_PER_TOUR, wgs); eng.set_option(T.OPT_BLOCK, block)
        eng.multistart_nn_2opt(starts[:8])
        ts = []
        for rep in range(2):
            t0 = time.perf_counter()
            res = eng.multistart_nn_2opt(starts)
            ts.append(time.perf_counter() - t0)
        i = eng.info()
        dt = min(ts)
        print(f"{what} elem={i['elem']} kernel={kernel} wgs={wgs} block={block}: used kernel={i['kernel']} wgs/tour={i['wgs_per_tour']} block={i['block']} fused={i.get('fused')} "
              f"best={res['cost']:.0f} sweeps={res['sweeps']} {dt*1e3:8.2f} ms  {res['sweeps']*T.evals_per_sweep(n)/dt/1e9:7.1f} Gevals/s", flush=True)
    except T.TspGpuError as e:
        print(f"{what} kernel={kernel} wgs={wgs} block={block}: {e}", flush=True)
eng.close()
