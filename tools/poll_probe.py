#!/usr/bin/env python3
"""exchange poll back-off probe (hook 95): us per sweep of the LDS-resident descent with s_sleep between polls"""
import sys, time
sys.path.insert(0, ".")
import travellingsalesmanoptimization_amd as T
from bench import reference_points, draw_points
draw_points([(4096, 123), (1024, 123)])
eng = T.Engine(0)
for n in (4096, 1024):
    eng.set_option(T.OPT_ELEM, T.ELEM_U16)
    eng.set_points(reference_points(n, 123)); eng.build_costs()
    eng.tour_nn(0, 0)
    for z in (0, 1, 2, 4, 8, 16):
        eng.set_option(95, z)
        ts = []
        for rep in range(6):
            eng.tour_copy(1, 0); eng.tour_store(1, want_path=False)
            t0 = time.perf_counter(); sw, rc = eng.tour_two_opt(1); ts.append(time.perf_counter() - t0)
        print(f"n={n} poll_sleep={z}: {min(ts[1:])/sw*1e6:.2f} us/sweep", flush=True)
eng.close()
