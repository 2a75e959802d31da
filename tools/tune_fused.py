#!/usr/bin/env python3
"""Whole-search time of the one-launch-per-sweep path over launch geometries (kernel, block, workgroups per tour).
usage: python tools/tune_fused.py [n] [u16|i32|f64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import travellingsalesmanoptimization_amd as T
from bench import reference_points
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ename = sys.argv[2] if len(sys.argv) > 2 else "u16"
elem = {"u16": T.ELEM_U16, "i32": T.ELEM_I32, "f64": T.ELEM_F64}[ename]
xy = reference_points(n, 123)
eng = T.Engine(0)
eng.set_option(T.OPT_ELEM, elem)
eng.set_points(xy); eng.build_costs(); eng.tour_nn(0, 0)
ev = T.evals_per_sweep(n)
grid = [(0, 0, 0)]
for k in (3, 2):
    for b in (0, 256, 512, 1024):
        for w in (0, 256, 384, 512, 640, 768, 1024):
            grid.append((k, b, w))
reps = 3 if n <= 8192 else 1
for k, b, w in grid:
    try:
        eng.set_option(T.OPT_KERNEL, k); eng.set_option(T.OPT_BLOCK, b); eng.set_option(T.OPT_WGS_PER_TOUR, w)
        eng.tour_copy(1, 0); eng.tour_two_opt(1)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.tour_copy(1, 0); sw, _ = eng.tour_two_opt(1)
        dt = (time.perf_counter() - t0) / reps
        i = eng.info()
        print(f"{ename} n={n} k{i['kernel']} block={i['block']:4d} wgs={i['wgs_per_tour']:4d} lds={i['lds_bytes']:6d} pipe2={i['pipe2']} fused={i['fused']}: "
              f"{1e3*dt:8.2f} ms {1e6*dt/sw:7.2f} us/sweep {sw*ev/dt/1e9:8.1f} Gevals/s", flush=True)
    except T.TspGpuError as e:
        print(f"{ename} k{k} block={b} wgs={w}: {str(e)[:90]}", flush=True)
eng.close()
