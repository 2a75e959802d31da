import sys, time
sys.path.insert(0, ".")
import numpy as np
from bench import reference_points, draw_points
draw_points([(4096, 123)])
import travellingsalesmanoptimization_amd as T
for elem, name, bpe in ((T.ELEM_F64, "f64", 16), (T.ELEM_I32, "i32", 8), (T.ELEM_U16, "u16", 4)):
    eng = T.Engine(0)
    eng.set_option(T.OPT_ELEM, elem)
    eng.set_points(reference_points(4096, 123)); eng.build_costs()
    starts = np.arange(64, dtype=np.int32)
    eng.multistart_nn_2opt(starts[:8])
    t0 = time.perf_counter(); res = eng.multistart_nn_2opt(starts); dt = time.perf_counter() - t0
    eng.set_option(T.OPT_TIMING, 1); eng.timing_read(reset=True)
    res2 = eng.multistart_nn_2opt(starts)
    ms, launches = eng.timing_read(reset=True)
    eng.set_option(T.OPT_TIMING, 0)
    ev = T.evals_per_sweep(4096)
    i = eng.info()
    print(f"{name}: kernel={i['kernel']} wgs={i['wgs_per_tour']} block={i['block']} best={res['cost']} sweeps={res['sweeps']} wall {dt*1e3:.1f} ms "
          f"{res['sweeps']*ev/dt:.3e} evals/s  nominal {res['sweeps']*ev*bpe/dt/8e12:.3f}; sweep kernels alone {ms:.1f} ms over {launches} launches: "
          f"{res2['sweeps']*ev*bpe/(ms*1e-3)/1e12:.2f} TB/s = {res2['sweeps']*ev*bpe/(ms*1e-3)/8e12:.3f}", flush=True)
    eng.close()
