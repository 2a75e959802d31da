#!/usr/bin/env python3
"""bench.py -- headline benchmark: 2-opt delta evaluations per second and wall-clock to the
2-opt local optimum on the n=4096 uniform-random EUC_2D instance (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one full best-improvement 2-opt local search on the device: the nearest-
neighbour tour (already resident in HBM, next to the cost matrix) is copied into a work
slot and swept to its local optimum -- hundreds of (sweep, apply-move) launch pairs,
exactly the trajectory of the reference's ref_2opt (src/algorithms/refinment.c:3-37).
Matrix build and NN construction are outside the timed region, as in the reference
(src/main.c:177 starts the clock after tsp_compute_costs) and SURVEY 8(d).

  value        = valid pair evaluations per second, whole job: sum over ranks of
                 sweeps * n(n-3)/2, divided by the max-over-ranks wall time
  ms_per_step  = wall-clock of one NN(start) -> local optimum search (per rank)
  N > 1        = weak scaling: rank r searches from NN start r (the multi-start loop of
                 h_greedy_2opt, heuristics.c:82-111, sharded), then ONE RCCL MIN
                 all-reduce picks the best tour and its owner broadcasts it (4n bytes).

Extra objects on the JSON line: "roofline" (dominant kernel = the sweep kernel; achieved =
algorithmic bytes per launch / mean kernel duration from HIP events on the engine's
stream) and "cpu_baseline" (the reference's own CPU 2-opt on this box's host, 1 core,
bounded sample).  Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); ~6.3 TB/s achievable


def reference_points(n, seed):
    """src/tsp.c:468-476 + utils.h:23-26 on glibc: srand(seed); x,y = rand()/RAND_MAX*10000-5000.
    Re-stated here (4 lines of libc calls) so that the product bench does not import the oracle."""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.srand(ctypes.c_uint(seed))
    RAND_MAX = 2147483647
    xy = np.empty((n, 2), dtype=np.float64)
    for i in range(n):
        xy[i, 0] = (libc.rand() / RAND_MAX) * 10000 + (-5000)
        xy[i, 1] = (libc.rand() / RAND_MAX) * 10000 + (-5000)
    return xy


def cpu_baseline(n, seed, sample_sweeps):
    """The reference's CPU 2-opt on this host, one core (the reference is single-threaded).
    kind "reference": oracle/_ref/libtspref.so = the reference's own sources compiled in the
    authoring container; kind "port": oracle/cpu_ref.c (bit-identical restatement)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    evals = n * (n - 3) // 2
    try:
        ref = O.Reference()
        ref.random(n, seed)
        succ, _, _ = ref.nn(0)
        t0 = time.perf_counter()
        sweeps, cost, _ = ref.two_opt_counted(succ, sample_sweeps)
        dt = time.perf_counter() - t0
        kind = "reference"
    except (FileNotFoundError, OSError):
        xy = O.random_points(n, seed)
        c = O.cost_matrix(xy)
        succ, _ = O.nn_tour(c, 0)
        t0 = time.perf_counter()
        sweeps, cost = O.two_opt(c, succ, sample_sweeps)
        dt = time.perf_counter() - t0
        kind = "port"
    return {"value": sweeps * evals / dt, "unit": "evals/s", "cores": 1, "kind": kind,
            "sample": f"first {sweeps} of the sweeps of the same n={n} seed={seed} NN(0) local search, "
                      f"{dt:.1f} s, f64 matrix, gcc -O3",
            "ms_per_sweep": 1e3 * dt / sweeps, "host_cores_available": os.cpu_count()}


def load_traffic(workload_key):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(workload_key)
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--elem", choices=["auto", "u16", "i32", "f64"], default=os.environ.get("TSPGPU_BENCH_ELEM", "auto"),
                    help="matrix storage; auto = the engine's default (narrowest exact copy)")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--wgs", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cpu-sweeps", type=int, default=300, help="bounded CPU baseline sample (0 = skip)")
    ap.add_argument("--batch-starts", type=int, default=64, help="starts of the batched multi-start leg (0 = skip)")
    ap.add_argument("--no-other", action="store_true", help="skip the comparison run with the other matrix storage")
    ap.add_argument("--no-sizes", action="store_true", help="skip the n=1024 / n=16384 rows of the throughput table")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import travellingsalesmanoptimization_amd as T
    from travellingsalesmanoptimization_amd import multistart

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)

    n, seed = args.n, args.seed
    evals = T.evals_per_sweep(n)
    eng = T.Engine(local)
    ELEMS = {"auto": T.ELEM_AUTO, "u16": T.ELEM_U16, "i32": T.ELEM_I32, "f64": T.ELEM_F64}
    NAMES = {T.ELEM_U16: "u16", T.ELEM_I32: "i32", T.ELEM_F64: "f64"}
    BYTES = {T.ELEM_U16: 2, T.ELEM_I32: 4, T.ELEM_F64: 8}
    eng.set_option(T.OPT_ELEM, ELEMS[args.elem])
    eng.set_option(T.OPT_KERNEL, args.kernel)
    eng.set_option(T.OPT_WGS_PER_TOUR, args.wgs)
    eng.set_option(T.OPT_BLOCK, args.block)
    eng.set_option(T.OPT_BATCH, args.batch)

    # ---- untimed setup: instance, matrix build on the device, NN seed tour in slot 0
    xy = reference_points(n, seed)
    eng.set_points(xy)
    eng.build_costs()
    build_ms = eng.time_build(5)
    start = rank % n                       # rank r = the r-th iteration of h_greedy_2opt's loop
    t0 = time.perf_counter()
    eng.tour_nn(0, start)
    _, nn_cost, _ = eng.tour_store(0, want_path=False)
    nn_ms = 1e3 * (time.perf_counter() - t0)
    eng.tour_copy(1, 0)

    def step():
        eng.tour_copy(1, 0)                 # restore the NN tour (device to device)
        sweeps, _ = eng.tour_two_opt(1)     # sweep to the local optimum, resident
        return sweeps

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    my_sweeps = 0
    for _ in range(args.steps):
        my_sweeps += step()
        if world > 1:
            # the exchange step of the sharded multi-start: one MIN all-reduce + winner broadcast
            path, cost, _ = eng.tour_store(1)
            multistart.select_best(cost, start, path, device=dev)
    sync_all()
    dt = time.perf_counter() - t0
    path, final_cost, _ = eng.tour_store(1)

    tot_sweeps, tmax = my_sweeps, dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s = torch.tensor([my_sweeps], dtype=torch.int64, device=dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        tmax, tot_sweeps = float(t.item()), int(s.item())

    # ---- roofline of the dominant kernel: the same search once more with HIP events on the engine's
    # stream: around every batch of back-to-back launches on the one-launch-per-sweep path (batch
    # time / launches), around every sweep launch otherwise
    roof = None
    if rank == 0:
        eng.set_option(T.OPT_TIMING, 1)
        eng.timing_read(reset=True)
        eng.tour_copy(1, 0)
        eng.tour_two_opt(1)
        ms_total, launches = eng.timing_read(reset=True)
        eng.set_option(T.OPT_TIMING, 0)
        info = eng.info()
        bytes_per_eval = 2 * BYTES[info["elem"]]                     # 2 matrix elements per eval
        kernel_ms = ms_total / max(launches, 1)
        achieved = evals * bytes_per_eval / (kernel_ms * 1e-3) / 1e9
        back2back_ms = eng.time_sweep(1, 50)                         # 50 launches, no apply in between
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(f"n{n}_{NAMES[info['elem']]}" + ("_fused" if info.get("fused") else "")),
                "kernel": "k_sweep_fused (sweep + apply of the previous move, one launch per sweep)" if info.get("fused") else
                          {1: "k_sweep_simple", 2: "k_sweep_pipe", 3: "k_sweep_res", 4: "k_sweep_otf"}[info["kernel"]],
                "kernel_ms_mean": kernel_ms, "kernel_launches_timed": launches,
                "kernel_ms_back_to_back": back2back_ms,   # the sweep part alone (no move applied between launches)
                "algorithmic_bytes_per_launch": evals * bytes_per_eval,
                "bytes_per_eval": bytes_per_eval, "evals_per_launch": evals,
                "kernel_evals_per_s": evals / (kernel_ms * 1e-3)}

    base = None
    if rank == 0 and world == 1 and args.cpu_sweeps > 0:
        base = cpu_baseline(n, seed, args.cpu_sweeps)

    # ---- the other matrix storage, same workload, for comparison (N = 1 only)
    other = None
    if rank == 0 and world == 1 and not args.no_other:
        oelem = "f64" if eng.info()["elem"] != T.ELEM_F64 else "i32"   # the reference's own format, f64
        e2 = T.Engine(local)
        e2.set_option(T.OPT_ELEM, ELEMS[oelem])
        e2.set_option(T.OPT_BATCH, args.batch)
        e2.set_points(xy); e2.build_costs(); e2.tour_nn(0, start); e2.tour_copy(1, 0)
        e2.tour_copy(1, 0); e2.tour_two_opt(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sw2 = 0
        for _ in range(args.steps):
            e2.tour_copy(1, 0); sw2 += e2.tour_two_opt(1)[0]
        dt2 = time.perf_counter() - t1
        e2.set_option(T.OPT_TIMING, 1); e2.timing_read(reset=True)
        e2.tour_copy(1, 0); e2.tour_two_opt(1)
        ms2, l2 = e2.timing_read(reset=True)
        e2.set_option(T.OPT_TIMING, 0)
        bpe2 = 16 if oelem == "f64" else 8
        k2 = ms2 / max(l2, 1)
        _, c2, _ = e2.tour_store(1, want_path=False)
        other = {"matrix_elem": oelem, "value": sw2 * evals / dt2, "unit": "evals/s", "ms_per_step": 1e3 * dt2 / args.steps,
                 "final_cost": c2, "kernel_ms_mean": k2, "bytes_per_eval": bpe2,
                 "roofline_achieved_GBs": evals * bpe2 / (k2 * 1e-3) / 1e9,
                 "roofline_frac": evals * bpe2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "traffic": load_traffic(f"n{n}_{oelem}")}
        e2.close()

    # ---- batched multi-start on the same instance (h_greedy_2opt's loop, 64 starts in flight):
    # the throughput-bound regime, next to the latency-bound single search above
    batch = None
    if rank == 0 and world == 1 and args.batch_starts > 0:
        starts = np.arange(args.batch_starts, dtype=np.int32)
        eng.multistart_nn_2opt(starts[:4])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = eng.multistart_nn_2opt(starts)
        dtb = time.perf_counter() - t1
        batch = {"starts": int(args.batch_starts), "sweeps": int(res["sweeps"]), "seconds": dtb,
                 "value": res["sweeps"] * evals / dtb, "unit": "evals/s", "best_cost": res["cost"],
                 "includes": "NN construction + 2-opt of every start, host arrays in/out"}

    if rank == 0:
        info = eng.info()
        out = {
            "metric": "2-opt delta evals/sec/node (n=4096 EUC_2D, NN(0) tour to 2-opt local optimum)",
            "value": tot_sweeps * evals / tmax,
            "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {T.ELEM_U16: "uint16 costs, int32 deltas", T.ELEM_I32: "int32", T.ELEM_F64: "f64"}[info["elem"]],
            "data": "synthetic",
            "config": {"workload": f"uniform-random EUC_2D n={n} (reference generator -n {n} -seed {seed}), "
                                   f"cost matrix resident in HBM, NN(start=rank) tour -> best-improvement 2-opt "
                                   f"to the local optimum; one step = one full local search",
                       "n": n, "seed": seed, "evals_per_sweep": evals,
                       "sweeps_per_step_rank0": my_sweeps // max(args.steps, 1),
                       "matrix_elem": {T.ELEM_U16: "uint16 exact copy", T.ELEM_I32: "int32 exact copy", T.ELEM_F64: "f64"}[info["elem"]],
                       "sweep_kernel": info["kernel"], "wgs_per_tour": info["wgs_per_tour"],
                       "block": info["block"], "lds_bytes": info["lds_bytes"], "batch": args.batch,
                       "parallelism": f"multistart-shard{world}"},
            "wall_clock_to_local_optimum_ms": 1e3 * tmax / args.steps,
            "final_cost_rank0": final_cost, "nn_cost_rank0": nn_cost,
            "matrix_build_ms": build_ms, "nn_tour_ms": nn_ms,
            "roofline": roof, "cpu_baseline": base, "other_matrix_storage": other, "multistart_batch": batch,
        }
        if base:
            out["gpu_over_cpu"] = out["value"] / base["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
