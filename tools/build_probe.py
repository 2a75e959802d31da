#!/usr/bin/env python3
"""K1 (tspgpu_build_costs) launch time by storage and kernel: the upper triangle computed once and stored twice
(k_build_costs_tri, default for integer cells) against every cell computed (k_build_costs_int)"""
import sys
sys.path.insert(0, ".")
from bench import reference_points, draw_points
draw_points([(1024, 1), (4096, 123), (16384, 123)])
import travellingsalesmanoptimization_amd as T
eng = T.Engine(0)
for n, seed in ((1024, 1), (4096, 123), (16384, 123)):
    xy = reference_points(n, seed)
    for elem, name, b in ((T.ELEM_U16, "u16", 2), (T.ELEM_I32, "i32", 4), (T.ELEM_F64, "f64", 8)):
        for build, tile in (((0, 0), (0, 64), (1, 0)) if elem == T.ELEM_U16 else ((0, 0), (1, 0)) if elem != T.ELEM_F64 else ((0, 0),)):
            eng.set_option(T.OPT_ELEM, elem); eng.set_option(T.OPT_BUILD_KERNEL, build); eng.set_option(92, tile)
            eng.set_points(xy); eng.build_costs()
            ms = eng.time_build(20)
            i = eng.info()
            nb = b * i["n"] * i["ld"]
            print(f"n={n} {name} build_kernel={build} tile={tile or 'auto'}: {ms*1e3:.2f} us, {nb/ms/1e6:.0f} GB/s stored = {nb/ms/1e6/8000:.3f} of 8 TB/s", flush=True)
eng.set_option(T.OPT_BUILD_KERNEL, 0); eng.set_option(92, 0)
eng.close()
