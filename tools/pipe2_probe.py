#!/usr/bin/env python3
"""Two-edge streaming form (pipe_stream2) on / off: batched multi-start rate on d18512 / n16384 / n4096 and the single
search at n=4096 with the pipelined kernel forced.  usage: python tools/pipe2_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
from travellingsalesmanoptimization_amd import tsplib
from bench import reference_points

def batch(what, nstarts, elem=T.ELEM_AUTO):
    if what.startswith("n") and what[1:].isdigit():
        xy, kind = reference_points(int(what[1:]), 123), T.EUC_2D
    else:
        xy, kind = tsplib.read(os.path.join(ROOT, "tests", "golden", "data", what + ".tsp"))
    n = len(xy)
    for pipe2 in (1, 0):
        eng = T.Engine(0)
        eng.set_option(T.OPT_ELEM, elem); eng.set_option(T.OPT_PIPE2, pipe2)
        eng.set_points(xy, kind); eng.build_costs()
        starts = np.arange(nstarts, dtype=np.int32)
        eng.set_option(T.OPT_SWEEP_CAP, 40)
        eng.multistart_nn_2opt(starts)
        eng.set_option(T.OPT_SWEEP_CAP, -1)
        t0 = time.perf_counter()
        res = eng.multistart_nn_2opt(starts)
        dt = time.perf_counter() - t0
        i = eng.info()
        ev = res["sweeps"] * T.evals_per_sweep(n)
        print(f"{what} elem={i['elem']} starts={nstarts} pipe2={i['pipe2']} kernel={i['kernel']}: best={res['cost']:.0f} sweeps={res['sweeps']} "
              f"{dt*1e3:.1f} ms {ev/dt/1e9:.1f} Gevals/s", flush=True)
        eng.close()

def single(n, elem, kernel):
    xy = reference_points(n, 123)
    for pipe2 in (1, 0):
        eng = T.Engine(0)
        eng.set_option(T.OPT_ELEM, elem); eng.set_option(T.OPT_KERNEL, kernel); eng.set_option(T.OPT_PIPE2, pipe2)
        eng.set_points(xy); eng.build_costs(); eng.tour_nn(0, 0)
        eng.tour_copy(1, 0); eng.tour_two_opt(1)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            eng.tour_copy(1, 0); sw, _ = eng.tour_two_opt(1)
        dt = (time.perf_counter() - t0) / reps
        _, cost, _ = eng.tour_store(1, want_path=False)
        i = eng.info()
        print(f"n={n} elem={i['elem']} kernel={i['kernel']} pipe2={i['pipe2']} fused={i['fused']}: {sw} sweeps {dt*1e3:.2f} ms {1e6*dt/sw:.2f} us/sweep cost={cost:.0f}", flush=True)
        eng.close()

single(4096, T.ELEM_U16, 2)
single(4096, T.ELEM_U16, 3)
single(8192, T.ELEM_U16, 0)
batch("d18512", 8)
batch("n16384", 8)
batch("n4096", 64, T.ELEM_I32)
batch("n4096", 64, T.ELEM_F64)
