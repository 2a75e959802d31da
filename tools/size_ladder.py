#!/usr/bin/env python3
"""One local search NN(0) -> 2-opt local optimum per size on uniform-random instances (reference generator, seed 123), with the
engine's defaults: which kernel runs, microseconds per sweep, evaluations per second -- the ladder across the limits of the
LDS-resident kernels (whole rows up to 4096, half windows up to ~5400, one launch per sweep beyond)."""
import sys, time
sys.path.insert(0, ".")
from bench import reference_points, draw_points
sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 3000, 4096, 4461, 5000, 5400, 5600, 6144, 8192, 16384]
draw_points([(n, 123) for n in sizes])
import travellingsalesmanoptimization_amd as T
for n in sizes:
    eng = T.Engine(0)
    eng.set_points(reference_points(n, 123)); eng.build_costs()
    eng.tour_nn(0, 0)
    ts = []
    for rep in range(3 if n <= 8192 else 2):
        eng.tour_copy(1, 0); eng.tour_store(1, want_path=False)
        t0 = time.perf_counter(); sw, rc = eng.tour_two_opt(1); ts.append(time.perf_counter() - t0)
    i = eng.info()
    kern = "k_str2opt (streamed, one launch)" if i.get("stream_persist") else \
           ("k_lds2opt_w (half windows)" if i["persist_window"] else "k_lds2opt (whole rows)") if i["persist"] else \
           ("k_sweep_fused" if i["fused"] else "k_sweep_*") + f" block {i['block']} x {i['wgs_per_tour']} wgs"
    best = min(ts[1:])
    print(f"n={n:6d} elem={['','f64','i32','u16'][i['elem']]} {kern:38s} sweeps={sw:5d} {best*1e3:9.3f} ms = {best/sw*1e6:7.2f} us/sweep  "
          f"{T.evals_per_sweep(n)*sw/best:.3e} evals/s  nominal HBM frac {T.evals_per_sweep(n)*4/(best/sw)/8e12:.3f}", flush=True)
    eng.close()
