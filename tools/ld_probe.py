#!/usr/bin/env python3
"""Does a power-of-two row stride cost bandwidth?  Back-to-back sweep time for n around a power of two.
usage: python tools/ld_probe.py n1 n2 ... [--elem u16|i32|f64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import travellingsalesmanoptimization_amd as T
from bench import reference_points
args = [a for a in sys.argv[1:] if not a.startswith("--")]
elem = {"u16": T.ELEM_U16, "i32": T.ELEM_I32, "f64": T.ELEM_F64, "auto": T.ELEM_AUTO}[
    next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--elem=")), "auto")]
kernel = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--kernel=")), "0"))
for n in [int(a) for a in args]:
    eng = T.Engine(0)
    eng.set_option(T.OPT_ELEM, elem)
    eng.set_option(T.OPT_KERNEL, kernel)
    eng.set_points(reference_points(n, 123)); eng.build_costs(); eng.tour_nn(0, 0)
    eng.set_option(T.OPT_FUSED, 0)
    ms = eng.time_sweep(0, 30)
    i = eng.info()
    ev = T.evals_per_sweep(n)
    bpe = {1: 16, 2: 8, 3: 4}[i["elem"]]
    print(f"n={n} ld={i['ld']} elem={i['elem']} kernel={i['kernel']} block={i['block']} wgs={i['wgs_per_tour']} "
          f"{ms*1e3:8.1f} us  {ev/ms/1e6:7.1f} Gevals/s  {ev*bpe/ms/1e6:6.0f} GB/s algorithmic", flush=True)
    eng.close()
