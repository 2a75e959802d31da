#!/usr/bin/env python3
"""The reference's own f64 cells at n=4096 (16 bytes per evaluation, 134 MB per sweep): microseconds per launch of the
one-launch-per-sweep kernel for the automatic plan and forced (block, workgroups, depth, pipe2) combinations."""
import sys, time
sys.path.insert(0, ".")
from bench import reference_points, draw_points
draw_points([(4096, 123)])
import travellingsalesmanoptimization_amd as T
eng = T.Engine(0)
eng.set_option(T.OPT_ELEM, T.ELEM_F64)
eng.set_points(reference_points(4096, 123)); eng.build_costs()
eng.tour_nn(0, 0)
def run(tag):
    ts = []
    for rep in range(3):
        eng.tour_copy(1, 0); eng.tour_store(1, want_path=False)
        t0 = time.perf_counter(); sw, rc = eng.tour_two_opt(1, max_sweeps=300); ts.append(time.perf_counter() - t0)
    i = eng.info()
    us = min(ts[1:]) / sw * 1e6
    print(f"{tag}: {us:.2f} us/sweep = {134.1e6/us/1e6/8:.3f} of 8 TB/s  kernel={i['kernel']} G={i['wgs_per_tour']} BT={i['block']} depth={i['depth']} fused={i['fused']} pipe2={i['pipe2']} lds={i['lds_bytes']}", flush=True)
run("auto")
for pipe2 in (1, 0):
    for block in (512, 768, 1024):
        for wgs in (256, 512):
            for depth in (0, 2, 3):
                try:
                    eng.set_option(T.OPT_PIPE2, pipe2); eng.set_option(T.OPT_KERNEL, 2); eng.set_option(T.OPT_WGS_PER_TOUR, wgs); eng.set_option(T.OPT_BLOCK, block)
                    eng.set_option(T.OPT_DEPTH, depth); eng.set_option(T.OPT_FUSED, 2)
                    run(f"pipe2={pipe2} block={block} wgs={wgs} depth={depth}")
                except Exception as e:
                    print(f"pipe2={pipe2} block={block} wgs={wgs} depth={depth}: {str(e)[:70]}")
eng.close()
