#!/usr/bin/env python3
"""Where does the sweep kernel's time go?  0 = real kernel, 1 = rows move but no pair is
evaluated, 2 = pairs are evaluated but no row traffic (diagnostic builds; wrong results)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import travellingsalesmanoptimization_amd as T
from bench import reference_points
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
xy = reference_points(n, 123)
for elem, ename in [(T.ELEM_U16, "u16"), (T.ELEM_I32, "i32"), (T.ELEM_F64, "f64")]:
    eng = T.Engine(0)
    eng.set_option(T.OPT_ELEM, elem)
    eng.set_points(xy); eng.build_costs(); eng.tour_nn(0, 0)
    for kernel, block, wgs, depth in [(0, 0, 0, 0), (3, 0, 0, 0), (3, 0, 1024, 0), (3, 256, 0, 0), (3, 1024, 0, 0), (2, 1024, 256, 2), (1, 0, 0, 0)]:
        row = []
        for ab in (0, 1, 2):
            try:
                eng.set_option(T.OPT_KERNEL, kernel)
            except T.TspGpuError:
                pass
            eng.set_option(T.OPT_BLOCK, block); eng.set_option(T.OPT_WGS_PER_TOUR, wgs); eng.set_option(T.OPT_DEPTH, depth)
            eng.set_option(99, ab)
            try:
                row.append(eng.time_sweep(0, 30) * 1e3)
            except T.TspGpuError as e:
                row.append(float("nan"))
        i = eng.info()
        print(f"{ename} k{i['kernel']} block={i['block']:4d} wgs={i['wgs_per_tour']:4d} depth={i['depth']}  full={row[0]:6.1f} us  no-eval={row[1]:6.1f} us  no-traffic={row[2]:6.1f} us", flush=True)
    eng.set_option(99, 0)
    eng.close()
