#!/usr/bin/env python3
"""Sweep-kernel tuning table on the GPU box: mean back-to-back duration of the sweep kernel
for (elem, block, wgs_per_tour) grids.  usage: python tools/tune_sweep.py [n] [seed] [u16,i32,f64]"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import travellingsalesmanoptimization_amd as T
sys.path.insert(0, os.path.join(ROOT, "bench_helpers"))
from bench import reference_points

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 123
xy = reference_points(n, seed)
ev = T.evals_per_sweep(n)
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for elem, ename, bpe in [(T.ELEM_U16, "u16", 4), (T.ELEM_I32, "i32", 8), (T.ELEM_F64, "f64", 16)]:
    if only and ename not in only: continue
    eng = T.Engine(0)
    eng.set_option(T.OPT_ELEM, elem)
    eng.set_points(xy); eng.build_costs()
    eng.tour_nn(0, 0)
    for kernel, block, wgs, depth in [(2, 0, 0, 0)] + [(2, b, w, d) for b in (512, 1024) for w in (256, 512) for d in (2, 4, 8)] + \
                                     [(1, b, w, 0) for b in (512, 1024) for w in (512, 1024)]:
        try:
            eng.set_option(T.OPT_KERNEL, kernel); eng.set_option(T.OPT_BLOCK, block); eng.set_option(T.OPT_WGS_PER_TOUR, wgs)
            eng.set_option(T.OPT_DEPTH, depth)
            ms = eng.time_sweep(0, 30)
            i = eng.info()
            print(f"{ename} k{kernel} block={i['block']:4d} wgs={i['wgs_per_tour']:4d} depth={i['depth']} lds={i['lds_bytes']:6d} "
                  f"{ms*1e3:7.1f} us  {ev/ms/1e6:8.1f} Gevals/s  {ev*bpe/ms/1e6:7.0f} GB/s algorithmic", flush=True)
        except T.TspGpuError as e:
            print(f"{ename} k{kernel} block={block} wgs={wgs}: {e}", flush=True)
    eng.close()
