#!/usr/bin/env python3
"""Launch shape of the streamed one-launch-per-sweep kernel for single uint16 tours past the resident kernel's size
(fnl4461 = BASELINE config 2, n = 5000 / 6000 / 8192 uniform): microseconds per sweep for the automatic plan and for
forced block sizes.  usage: python tools/tune_n4461.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import travellingsalesmanoptimization_amd as T
from bench import read_tsplib, reference_points
def bench(xy, label):
    eng = T.Engine(0)
    eng.set_option(T.OPT_PERSIST, 0)
    eng.set_points(xy); eng.build_costs()
    eng.tour_nn(0, 0)
    def run(tag):
        ts = []
        for rep in range(3):
            eng.tour_copy(1, 0); eng.tour_store(1, want_path=False)
            t0 = time.perf_counter(); sw, rc = eng.tour_two_opt(1); ts.append(time.perf_counter() - t0)
        i = eng.info()
        print(f"{label} {tag}: {min(ts[1:])/sw*1e6:.2f} us/sweep kernel={i['kernel']} G={i['wgs_per_tour']} BT={i['block']} fused={i['fused']} pipe2={i['pipe2']}", flush=True)
    run("auto")
    for block in (256, 320, 384, 448, 512, 576, 640, 704, 768, 1024):
        for wgs in (0,):
            try:
                eng.set_option(T.OPT_KERNEL, 2); eng.set_option(T.OPT_WGS_PER_TOUR, wgs); eng.set_option(T.OPT_BLOCK, block); eng.set_option(T.OPT_FUSED, 2)
                run(f"kernel=2 wgs={wgs} block={block}")
            except Exception as e:
                print(f"{label} block={block}: {str(e)[:80]}")
    eng.close()
bench(read_tsplib(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data", "fnl4461.tsp"))[0], "fnl4461")
for n in (5000, 6000, 8192):
    bench(reference_points(n, 123), f"n={n}")
