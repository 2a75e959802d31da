#!/usr/bin/env python3
"""Launch shape of the streamed one-launch-per-sweep kernel for single uint16 tours between the half-window kernel's limit
(~5400) and 8192: microseconds per sweep for the automatic plan and for forced (block, workgroups) pairs."""
import sys, time, os
sys.path.insert(0, ".")
from bench import reference_points, draw_points
sizes = [5600, 6144, 7168, 8192]
draw_points([(n, 123) for n in sizes])
import travellingsalesmanoptimization_amd as T
for n in sizes:
    eng = T.Engine(0)
    eng.set_option(T.OPT_PERSIST, 0)
    eng.set_points(reference_points(n, 123)); eng.build_costs()
    eng.tour_nn(0, 0)
    def run(tag):
        ts = []
        for rep in range(3):
            eng.tour_copy(1, 0); eng.tour_store(1, want_path=False)
            t0 = time.perf_counter(); sw, rc = eng.tour_two_opt(1, max_sweeps=200); ts.append(time.perf_counter() - t0)
        i = eng.info()
        print(f"n={n} {tag}: {min(ts[1:])/sw*1e6:.2f} us/sweep kernel={i['kernel']} G={i['wgs_per_tour']} BT={i['block']} fused={i['fused']} pipe2={i['pipe2']} lds={i['lds_bytes']}", flush=True)
    run("auto")
    for block in (384, 448, 512, 640, 768, 896, 1024):
        for wgs in (0, 256, 512):
            try:
                eng.set_option(T.OPT_KERNEL, 2); eng.set_option(T.OPT_WGS_PER_TOUR, wgs); eng.set_option(T.OPT_BLOCK, block); eng.set_option(T.OPT_FUSED, 2)
                run(f"block={block} wgs={wgs}")
            except Exception as e:
                print(f"n={n} block={block} wgs={wgs}: {str(e)[:80]}")
    eng.close()
