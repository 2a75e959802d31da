#!/usr/bin/env python3
"""bench.py -- headline benchmark: 2-opt delta evaluations per second and wall-clock to the
2-opt local optimum on the n=4096 uniform-random EUC_2D instance (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one full best-improvement 2-opt local search on the device: the nearest-
neighbour tour (already resident in HBM, next to the cost matrix) is copied into a work
slot and swept to its local optimum -- hundreds of one-launch-per-sweep kernels, exactly the
trajectory of the reference's ref_2opt (src/algorithms/refinment.c:3-37).  Matrix build and NN
construction are outside the timed region, as in the reference (src/main.c:177 starts the
clock after tsp_compute_costs) and SURVEY 8(d).

  value        = valid pair evaluations per second, whole job: sum over ranks of
                 sweeps * n(n-3)/2, divided by the max-over-ranks wall time
  ms_per_step  = wall-clock of one NN(start) -> local optimum search (per rank)
  N > 1        = weak scaling: rank r searches from NN start r (the multi-start loop of
                 h_greedy_2opt, heuristics.c:82-111, sharded), then ONE RCCL MIN
                 all-reduce picks the best tour and its owner broadcasts it (4n bytes).

Objects on the same JSON line (rank 0; the auxiliary legs run after the timed region):
  roofline               dominant kernel (the sweep): algorithmic bytes per launch / mean launch duration from HIP events
                         on the engine's stream, against the 8 TB/s HBM peak
  roofline_build         K1 (k_build_costs): bytes stored / launch duration against the same peak
  cpu_baseline           the reference's own CPU 2-opt on this box's host, 1 core, bounded sample
  other_matrix_storage   the same workload with the reference's own f64 cells (16 B / evaluation)
  sizes                  the n = 1024 and n = 16384 rows of north_star's throughput table (each with its own parity gate
                         against a committed golden, its roofline and a bounded CPU sample)
  multistart_batch       64 starts in flight (NN + 2-opt, the throughput-bound regime)
  otf                    matrix-free sweep on pla85900 (config 5): bound "valu", ceiling from the kernel's ISA
                         (tools/otf_isa_count.py -> profiles/r03_otf_isa_ceiling.json)
  cpu_multistart_baseline  All-NN+2OPT on pr1002, all 1002 starts: the reference on ALL host cores of this box's share
                         (one process per core over disjoint start ranges) next to the engine's time
  host_c_path            the drop-in `tsp` binary (C host layer, TSP_GPU_DEVICES = the N devices of this run): the
                         multi-start sharded in C, exchange by RCCL (csrc/tspgpu_multi.cpp)
Nothing here reads /root/reference; goldens and instances come from tests/golden/.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); ~6.3 TB/s achievable
LDS_PEAK_GBS = 256 * 256 * 2.4    # MI355X_MICROARCH.md LDS: ds_read_b128 256 B/clk/CU x 256 CUs x 2.4 GHz = 157 TB/s
EXCHANGE_FLOOR_US = 2.1           # profiles/r03_xcd_exchange_probe.txt: one flat 256-workgroup slot exchange on an idle chip (2.0-2.2 us)

_POINTS = {}


def reference_points(n, seed):
    """src/tsp.c:468-476 + utils.h:23-26 on glibc: srand(seed); x,y = rand()/RAND_MAX*10000-5000.
    Re-stated here (4 lines of libc calls) so that the product bench does not import the oracle.
    The process-wide rand() stream is also drawn from by the HIP runtime while it initialises, so main() draws every
    point set it will need BEFORE the first GPU call (draw_points) and this function then only serves the cache."""
    if (n, seed) in _POINTS:
        return _POINTS[(n, seed)].copy()
    assert not _POINTS.get("sealed"), f"point set ({n}, {seed}) requested after the first GPU call: add it to draw_points()"
    import ctypes
    libc = ctypes.CDLL(None)
    libc.srand(ctypes.c_uint(seed))
    RAND_MAX = 2147483647
    xy = np.empty((n, 2), dtype=np.float64)
    for i in range(n):
        xy[i, 0] = (libc.rand() / RAND_MAX) * 10000 + (-5000)
        xy[i, 1] = (libc.rand() / RAND_MAX) * 10000 + (-5000)
    _POINTS[(n, seed)] = xy
    return xy.copy()


_VNS_RAND = None


def libc_rand_values(seed, count):
    """the first `count` values of glibc's rand() after srand(seed) (the stream a reference process would draw from)"""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.srand(ctypes.c_uint(seed))
    return np.array([libc.rand() for _ in range(count)], dtype=np.int32)


def draw_points(sets):
    """every uniform-random instance of the run, drawn before HIP is initialised; later requests must hit the cache"""
    for n, seed in sets:
        reference_points(n, seed)
    _POINTS["sealed"] = True


def read_tsplib(path):
    """NODE_COORD_SECTION reader (instances under tests/golden/data)"""
    n, xy, on, ewt = None, None, False, "EUC_2D"
    with open(path) as f:
        for line in f:
            t = line.replace(":", " ").split()
            if not t:
                continue
            if on:
                if t[0] == "EOF":
                    break
                xy[int(t[0]) - 1] = (float(t[1]), float(t[2]))
            elif t[0] == "DIMENSION":
                n = int(t[1]); xy = np.zeros((n, 2), dtype=np.float64)
            elif t[0] == "EDGE_WEIGHT_TYPE":
                ewt = t[1]
            elif t[0] == "NODE_COORD_SECTION":
                on = True
    return xy, ewt


def fnv1a(succ):
    """64-bit FNV-1a over the successor array, one step per node (the hash of the golden fixtures)"""
    h = 0xcbf29ce484222325
    for v in np.asarray(succ, dtype=np.int64).tolist():
        h = ((h ^ (v & 0xFFFFFFFF)) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def golden_two_opt(n, seed):
    """committed golden of the NN(0) -> local optimum search on `-n n -seed seed` (from the compiled reference)"""
    try:
        if n == 16384 and seed == 123:
            return json.load(open(os.path.join(GOLDEN, "golden_n16384_s123.json")))["two_opt"]
        return json.load(open(os.path.join(GOLDEN, "golden.json")))["random"][f"n{n}_s{seed}"]["two_opt"]
    except (OSError, KeyError):
        return None


def cpu_baseline(n, seed, sample_sweeps, prefer_port=False):
    """The reference's CPU 2-opt on this host, one core (the reference is single-threaded).
    kind "reference": oracle/_ref/libtspref.so = the reference's own sources compiled in the
    authoring container; kind "port": oracle/cpu_ref.c (bit-identical restatement)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    evals = n * (n - 3) // 2
    try:
        if prefer_port:
            raise FileNotFoundError
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


def _cpu_multistart_worker(args):
    """one process of the all-cores CPU baseline: NN + ref_2opt from starts [lo, hi) with the reference's own code"""
    path, lo, hi = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    try:
        ref = O.Reference(scratch=os.path.join("/tmp", f"tspref_scratch_{os.getpid()}"))
        ref.read_file(path)
        kind = "reference"
        best = (float("inf"), -1)
        t0 = time.perf_counter()
        for s in range(lo, hi):
            succ, _, _ = ref.nn(s)
            c = ref.ref_2opt(succ)
            if c < best[0]:
                best = (c, s)
        return best, time.perf_counter() - t0, kind
    except (FileNotFoundError, OSError):
        xy, _ = O.read_tsplib(path)
        c = O.cost_matrix(xy)
        t0 = time.perf_counter()
        p, cost, start, _ = O.multistart_nn_2opt(c, np.arange(lo, hi, dtype=np.int32))
        return (cost, start), time.perf_counter() - t0, "port"


def cpu_multistart_baseline(name, procs):
    """"for the multi-start comparison only, one process per core over disjoint start-node ranges with
    nproc stated" -- the whole All-NN+2OPT job of `name` on `procs` host cores."""
    import multiprocessing as mp
    path = os.path.join(DATA, name + ".tsp")
    n = read_tsplib(path)[0].shape[0]
    cuts = [n * i // procs for i in range(procs + 1)]
    jobs = [(path, cuts[i], cuts[i + 1]) for i in range(procs) if cuts[i + 1] > cuts[i]]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        res = pool.map(_cpu_multistart_worker, jobs)
    wall = time.perf_counter() - t0
    best = min(r[0] for r in res)
    return {"instance": name, "starts": n, "procs": len(jobs), "host_cores_available": os.cpu_count(),
            "seconds": wall, "slowest_process_compute_s": max(r[1] for r in res), "best_cost": best[0], "best_start": best[1],
            "kind": res[0][2], "what": "NN + ref_2opt from every start, reference code, one process per core over disjoint start ranges"}


def load_traffic(workload_key):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(workload_key)
        except Exception:
            return None
    return None


def guarded(fn, *a, **kw):
    """an auxiliary leg must not take the bench line down with it"""
    try:
        return fn(*a, **kw)
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--size", dest="n", type=int, default=4096, help="nodes of the uniform-random instance (--size: unambiguous behind torch.distributed.run)")
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--elem", choices=["auto", "u16", "i32", "f64"], default=os.environ.get("TSPGPU_BENCH_ELEM", "auto"),
                    help="matrix storage; auto = the engine's default (narrowest exact copy)")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--persist", type=int, default=1, choices=[0, 1, 2],
                    help="TSPGPU_OPT_PERSIST: 1 = the LDS-resident descent where it applies (default), 0 = one launch per sweep")
    ap.add_argument("--wgs", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cpu-sweeps", type=int, default=300, help="bounded CPU baseline sample (0 = skip)")
    ap.add_argument("--batch-starts", type=int, default=64, help="starts of the batched multi-start leg (0 = skip)")
    ap.add_argument("--job-starts", type=int, default=None, help="starts of the fixed-size sharded multi-start job timed at EVERY --gpus N (0 = skip)")
    ap.add_argument("--no-other", action="store_true", help="skip the comparison run with the other matrix storage")
    ap.add_argument("--no-sizes", action="store_true", help="skip the n=1024 / n=16384 rows of the throughput table")
    ap.add_argument("--no-otf", action="store_true", help="skip the matrix-free (pla85900) leg")
    ap.add_argument("--no-cpu-multistart", action="store_true", help="skip the all-cores CPU multi-start baseline")
    ap.add_argument("--no-host-c", action="store_true", help="skip the `tsp` binary (C host layer, multi-device) leg")
    ap.add_argument("--lean", action="store_true", help="headline + roofline only (profiling runs)")
    args = ap.parse_args()
    if args.lean:
        args.no_other = args.no_sizes = args.no_otf = args.no_cpu_multistart = args.no_host_c = True
        args.cpu_sweeps = 0
        args.batch_starts = 0
    if args.job_starts is None:
        args.job_starts = 0 if args.lean else 512

    draw_points([(args.n, args.seed), (1024, 1), (16384, 123), (1024, 123), (3584, 123), (4096, 123)])
    global _VNS_RAND
    _VNS_RAND = libc_rand_values(1, 64 * 1000 + 4096)          # the VNS leg's numbers: also before HIP is initialised
    import torch
    import torch.distributed as dist
    import travellingsalesmanoptimization_amd as T
    from travellingsalesmanoptimization_amd import multistart

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        # "nccl" = RCCL over xGMI; TSPGPU_BENCH_BACKEND=gloo rehearses the N > 1 branch on a box with fewer GPUs than ranks
        # (collectives on CPU tensors, ranks sharing a device): tests/test_host_c.py
        dist.init_process_group(os.environ.get("TSPGPU_BENCH_BACKEND", "nccl"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    backend = dist.get_backend() if world > 1 else None
    local = local % max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local) if backend != "gloo" else torch.device("cpu")
    aux = rank == 0 and world == 1          # the auxiliary single-GPU legs

    ELEMS = {"auto": T.ELEM_AUTO, "u16": T.ELEM_U16, "i32": T.ELEM_I32, "f64": T.ELEM_F64}
    NAMES = {T.ELEM_U16: "u16", T.ELEM_I32: "i32", T.ELEM_F64: "f64"}
    BYTES = {T.ELEM_U16: 2, T.ELEM_I32: 4, T.ELEM_F64: 8}
    CTYPE = {T.ELEM_U16: "unsigned short", T.ELEM_I32: "int", T.ELEM_F64: "double"}

    def kernel_name(info):
        if info.get("persist") and info.get("persist_window"):
            return ("k_lds2opt_w (uint16 matrix resident in LDS as half-window rows in tour order, ONE launch per descent; "
                    "kernel_ms_mean = launch duration / sweeps of the launch)")
        if info.get("persist"):
            return ("k_lds2opt (uint16 matrix resident in LDS in tour order, ONE launch per descent; "
                    "kernel_ms_mean = launch duration / sweeps of the launch)")
        if info.get("fused"):
            return "k_sweep_fused (sweep + apply of the previous move, one launch per sweep)"
        return {1: "k_sweep_simple", 2: "k_sweep_pipe", 3: "k_sweep_res", 4: "k_sweep_otf"}[info["kernel"]]

    def phase_clocks(eng):
        """the LDS-resident kernel's own phase clocks (option 98: wall_clock64 sums by thread 0 of every workgroup), one more
        descent of slot 0's tour: mean over workgroups, us per sweep"""
        eng.set_option(98, 1)
        try:
            eng.tour_copy(1, 0)
            eng.tour_two_opt(1)
            buf = np.zeros(1024 * 64, dtype=np.uint64)
            eng.L.tspgpu_debug_stamps(eng.ctx, buf.ctypes.data, buf.size)
        finally:
            eng.set_option(98, 0)
        st = buf.reshape(-1, 16)[:256, :9].astype(np.float64)
        st = st[st[:, 8] > 0]
        if not len(st):
            return None
        names = ["evaluation", "workgroup_reduction", "exchange", "reversal", "rows_fetched", "decode_and_swaps"]
        return {nm: float((st[:, i] / st[:, 8]).mean() / 100.0) for i, nm in enumerate(names)}

    def timed_search(eng, evals):
        """the same search once more with HIP events on the engine's stream: around every batch of back-to-back
        launches on the one-launch-per-sweep path (batch time / launches), around every sweep launch otherwise"""
        eng.set_option(T.OPT_TIMING, 1)
        eng.timing_read(reset=True)
        eng.tour_copy(1, 0)
        eng.tour_two_opt(1)
        ms_total, launches = eng.timing_read(reset=True)
        eng.set_option(T.OPT_TIMING, 0)
        info = eng.info()
        bpe = 2 * BYTES[info["elem"]]                     # 2 matrix elements per eval
        kernel_ms = ms_total / max(launches, 1)
        achieved = evals * bpe / (kernel_ms * 1e-3) / 1e9
        suffix = "_persist" if info.get("persist") else "_fused" if info.get("fused") else ""
        out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "traffic": None,    # PMC counters need their own rocprofv3 pass; the committed pass is quoted beside it
               "traffic_from_committed_profile": load_traffic(f"n{info['n']}_{NAMES[info['elem']]}" + suffix),
               "kernel": kernel_name(info), "kernel_ms_mean": kernel_ms, "kernel_launches_timed": launches,
               "algorithmic_bytes_per_launch": evals * bpe, "bytes_per_eval": bpe, "evals_per_launch": evals,
               "kernel_evals_per_s": evals / (kernel_ms * 1e-3)}
        if info.get("persist"):
            # One launch runs the whole descent; the unit of work is one SWEEP (= the work of one launch of the per-sweep
            # kernels: n(n-3)/2 evaluations, 2 matrix cells each).  `frac` keeps SURVEY 8(d)'s convention -- algorithmic
            # bytes / time / 8 TB/s -- although the cells come from LDS, not HBM (PMC: ~1.2 MB of HBM
            # traffic per sweep against 33.5 MB of algorithmic bytes): what limits a sweep is one grid-wide exchange, named
            # in `limited_by`; the ratio against the measured floor of one exchange is the secondary `exchange_floor_frac`.
            us = kernel_ms * 1e3
            ph = guarded(phase_clocks, eng)
            out.update({"limited_by": "latency: one grid-wide exchange per sweep (a fabric round trip all 256 workgroups wait for) + the "
                                      "workgroups whose move phase reloads rows; the cells are read from LDS, not HBM",
                        "us_per_sweep": us,
                        "exchange_floor_us": EXCHANGE_FLOOR_US, "exchange_floor_frac": EXCHANGE_FLOOR_US / us,
                        "exchange_floor_source": "profiles/r03_xcd_exchange_probe.txt (256 workgroups, 16-byte records at 64-byte stride, flat: 2.0-2.2 us; "
                                                 "the two-level form through the XCDs' L2s measures the same)",
                        "phase_us": ph,
                        "frac_nominal_hbm": achieved / HBM_PEAK_GBS, "achieved_nominal_hbm_GBs": achieved, "hbm_peak_GBs": HBM_PEAK_GBS,
                        "unit_of_work": "sweep (launch duration / sweeps run by the launch)",
                        "kernel_launches_timed": 1, "sweeps_timed": launches})
            if isinstance(ph, dict) and ph.get("evaluation"):
                # LDS bytes the evaluation reads per sweep = the algorithmic bytes (every pair: 2 cells of 2 bytes), over the
                # evaluation phase alone, against the chip's ds_read_b128 rate
                lds_gbs = evals * bpe / (ph["evaluation"] * 1e-6) / 1e9
                out["lds_frac"] = lds_gbs / LDS_PEAK_GBS
                out["lds_read_GBs_in_evaluation_phase"] = lds_gbs
                out["lds_peak_GBs"] = LDS_PEAK_GBS
            out["note"] = ("matrix cells are read from LDS; HBM traffic per sweep is the rows re-fetched after a move (see "
                           "traffic_from_committed_profile); the sweep is bound by the grid-wide exchange (DESIGN.md 4.7)")
        return out

    def build_roofline(eng):
        """K1: one launch stores sizeof(cell) * n * ld bytes (write-bound)"""
        info = eng.info()
        ms = eng.time_build(5)
        nbytes = BYTES[info["elem"]] * info["n"] * info["ld"]
        ach = nbytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "kernel_ms_mean": ms,
                "kernel": "k_build_costs<double>" if info["elem"] == T.ELEM_F64 else ((f"k_build_costs_tri128<{CTYPE[info['elem']]}, kind> (upper triangle computed once, every 128 x 128 tile stored twice)" if info["n"] >= 8192 else
                            f"k_build_costs_tri<{CTYPE[info['elem']]}, kind> (upper triangle computed once, every 64 x 64 tile stored twice)") if info["n"] >= 128 and info["elem"] == T.ELEM_U16 else f"k_build_costs_int<{CTYPE[info['elem']]}, kind>"),
                "algorithmic_bytes_per_launch": nbytes, "bytes_per_cell": BYTES[info["elem"]],
                "f64_cells_equivalent_GBs": 8 * info["n"] * info["ld"] / (ms * 1e-3) / 1e9}

    def parity(n, seed, sweeps, cost, path):
        g = golden_two_opt(n, seed)
        if g is None:
            return {"golden": None}
        got = {"sweeps": int(sweeps), "final_cost": float(cost), "final_fnv": fnv1a(path)}
        ok = all(got[k] == g[k] for k in got)
        return {"golden": f"tests/golden: n{n}_s{seed} two_opt (compiled reference)", "ok": ok, **got}

    def multistart_golden():
        """per-start results of h_greedy_2opt's loop on -n 4096 -seed 123 from the compiled reference"""
        try:
            g = json.load(open(os.path.join(GOLDEN, "golden_n4096_multistart.json")))
            return {e["start"]: e for e in g["starts"]} if (g["n"], g["seed"]) == (n, seed) else None
        except OSError:
            return None

    def multistart_gate(e, starts, res, check_slots=8):
        """a batched multi-start against the compiled reference's per-start goldens: winner (cost, start, tour hash), sweep
        total, and the first `check_slots` starts' own final cost and tour read back from their slots"""
        g = multistart_golden()
        if g is None or any(int(st) not in g for st in starts):
            return {"golden": None}
        want = [g[int(st)] for st in starts]
        best = min(want, key=lambda w: (w["cost"], w["start"]))
        got = {"best_cost": float(res["cost"]), "best_start": int(res["start"]), "best_fnv": fnv1a(res["path"]), "sweeps": int(res["sweeps"])}
        ok = got == {"best_cost": best["cost"], "best_start": best["start"], "best_fnv": best["fnv"], "sweeps": sum(w["sweeps"] for w in want)}
        slots_ok = 0
        for slot in range(min(check_slots, len(want))):
            p_, c_, _ = e.tour_store(slot)
            good = (float(c_), fnv1a(p_)) == (want[slot]["cost"], want[slot]["fnv"])
            slots_ok += good
            ok = ok and good
        return {"golden": f"tests/golden/golden_n4096_multistart.json ({len(want)} starts, compiled reference)", "ok": bool(ok),
                "per_start_checked": int(min(check_slots, len(want))), "per_start_ok": int(slots_ok), **got}

    def batch_measure(e, nstarts):
        """All-NN + 2-opt over starts 0..nstarts-1 of the resident instance in ONE batched call (h_greedy_2opt's loop,
        heuristics.c:74-116; what h_Greedy_2opt_mod_costs runs on a caller's f64 matrix): wall-clock, parity gate against the
        compiled reference's per-start goldens, and the dominant kernel's own time from HIP events"""
        starts = np.arange(nstarts, dtype=np.int32)
        e.multistart_nn_2opt(starts[:4])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = e.multistart_nn_2opt(starts)
        dtb = time.perf_counter() - t1
        info_b = e.info()
        gate_b = multistart_gate(e, starts, res)
        if gate_b.get("golden") and not gate_b["ok"]:
            raise SystemExit(f"parity gate of the batched multi-start failed: {gate_b}")
        bpe = 2 * BYTES[info_b["elem"]]
        # the same call once more with HIP events around every sweep launch of the batch (graphs off): the dominant kernel's
        # own time.  One launch sweeps every live tour of the batch: its algorithmic bytes = live tours x evals x bpe
        e.set_option(T.OPT_TIMING, 1); e.timing_read(reset=True)
        res_t = e.multistart_nn_2opt(starts)
        ms_total, launches = e.timing_read(reset=True)
        e.set_option(T.OPT_TIMING, 0)
        ach = res_t["sweeps"] * evals * bpe / (ms_total * 1e-3) / 1e9
        return {"starts": int(nstarts), "matrix_elem": NAMES[info_b["elem"]], "sweeps": int(res["sweeps"]), "seconds": dtb,
                "value": res["sweeps"] * evals / dtb, "unit": "evals/s", "best_cost": res["cost"], "best_start": int(res["start"]),
                "parity": gate_b,
                "includes": "NN construction + 2-opt of every start, host arrays in/out",
                "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                             "traffic": None, "traffic_from_committed_profile": load_traffic(f"n{n}_{NAMES[info_b['elem']]}_batch{nstarts}"),
                             "kernel": f"k_sweep_pipe<{CTYPE[info_b['elem']]}> (rows streamed, runs of {info_b['wgs_per_tour']} workgroups per tour; one launch sweeps every live tour of the batch)",
                             "kernel_ms_mean": ms_total / max(launches, 1), "kernel_launches_timed": int(launches),
                             "algorithmic_bytes_per_launch": res_t["sweeps"] * evals * bpe / max(launches, 1),
                             "mean_live_tours_per_launch": res_t["sweeps"] / max(launches, 1), "bytes_per_eval": bpe,
                             "note": "HIP events around every sweep launch of a second, identical call (k_apply and NN are outside); the "
                                     "tours of a batch share one matrix the last-level cache holds, so these are nominal bytes "
                                     "(SURVEY 8d), not HBM traffic"},
                # SURVEY 8(d)'s convention over the whole call (NN + sweeps + applies, host in/out)
                "nominal_hbm": {"bytes_per_eval": bpe, "achieved_GBs": res["sweeps"] * evals * bpe / dtb / 1e9,
                                "peak_GBs": HBM_PEAK_GBS, "frac": res["sweeps"] * evals * bpe / dtb / 1e9 / HBM_PEAK_GBS}}

    def search_row(n, seed, steps, warmup, elem, cpu_sweeps, port):
        """one row of the throughput table on its own engine (single GPU)"""
        evals = T.evals_per_sweep(n)
        e = T.Engine(local)
        try:
            e.set_option(T.OPT_ELEM, ELEMS[elem]); e.set_option(T.OPT_BATCH, args.batch)
            xy = reference_points(n, seed)
            e.set_points(xy); e.build_costs()
            broof = build_roofline(e)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.tour_nn(0, 0)
            _, nn_cost, _ = e.tour_store(0, want_path=False)
            nn_ms = 1e3 * (time.perf_counter() - t0)
            for _ in range(warmup):
                e.tour_copy(1, 0); e.tour_two_opt(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sw = 0
            for _ in range(steps):
                e.tour_copy(1, 0); sw += e.tour_two_opt(1)[0]
            dt = time.perf_counter() - t0
            path, cost, _ = e.tour_store(1)
            info = e.info()
            row = {"n": n, "seed": seed, "value": sw * evals / dt, "unit": "evals/s", "ms_per_step": 1e3 * dt / steps,
                   "steps": steps, "sweeps_per_step": sw // steps, "final_cost": cost, "nn_cost": nn_cost, "nn_tour_ms": nn_ms,
                   "matrix_elem": NAMES[info["elem"]], "sweep_kernel": info["kernel"], "wgs_per_tour": info["wgs_per_tour"],
                   "block": info["block"], "lds_bytes": info["lds_bytes"],
                   "parity": parity(n, seed, sw // steps, cost, path),
                   "roofline": timed_search(e, evals), "roofline_build": broof}
        finally:
            e.close()
        if cpu_sweeps > 0:
            row["cpu_baseline"] = guarded(cpu_baseline, n, seed, cpu_sweeps, port)
            if "value" in row["cpu_baseline"]:
                row["gpu_over_cpu"] = row["value"] / row["cpu_baseline"]["value"]
        return row

    n, seed = args.n, args.seed
    evals = T.evals_per_sweep(n)
    eng = T.Engine(local)
    eng.set_option(T.OPT_ELEM, ELEMS[args.elem])
    eng.set_option(T.OPT_KERNEL, args.kernel)
    eng.set_option(T.OPT_WGS_PER_TOUR, args.wgs)
    eng.set_option(T.OPT_BLOCK, args.block)
    eng.set_option(T.OPT_BATCH, args.batch)
    eng.set_option(T.OPT_PERSIST, args.persist)

    # ---- untimed setup: instance, matrix build on the device, NN seed tour in slot 0
    xy = reference_points(n, seed)
    eng.set_points(xy)
    eng.build_costs()
    broof = build_roofline(eng) if rank == 0 else None
    start = rank % n                       # rank r = the r-th iteration of h_greedy_2opt's loop
    torch.cuda.synchronize()
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

    info_main = eng.info()                 # (the path the timed steps took)
    gate = parity(n, seed, my_sweeps // max(args.steps, 1), final_cost, path) if rank == 0 and start == 0 else None
    if gate and gate.get("golden") and not gate["ok"]:
        raise SystemExit(f"parity gate failed: {gate}")        # a fast kernel whose result differs is not a result

    # ---- north_star's multi-GPU criterion: ONE fixed job -- h_greedy_2opt's loop (heuristics.c:74-116)
    # over the first `job_starts` start nodes of this instance -- sharded p mod N over the ranks (multistart.py), every rank's
    # shard one batched call of the HIP engine, ONE MIN all-reduce + ONE broadcast (RCCL) to select the winner.  Timed at
    # every N including 1, so the driver's N = 1, 2, 4, 8 runs give the strong-scaling curve of the same job.
    job = None
    if args.job_starts > 0:
        jstarts = np.arange(min(args.job_starts, n), dtype=np.int32)

        def solve_local(mine):
            r_ = eng.multistart_nn_2opt(mine)
            solve_local.mine, solve_local.res = mine, r_
            return r_

        def job_leg():
            eng.multistart_nn_2opt(jstarts[rank::world][:4])           # warm-up: plan, slots, graphs
            sync_all()
            t1 = time.perf_counter()
            res_j = multistart.multistart_nn_2opt(solve_local, jstarts, device=dev, n=n)
            sync_all()
            dtj = time.perf_counter() - t1
            # every rank checks its own shard against the compiled reference's per-start goldens
            gl = multistart_gate(eng, solve_local.mine, solve_local.res)
            ok_l = 1 if (gl.get("golden") is None or gl["ok"]) else 0
            if world > 1:
                tj = torch.tensor([dtj], dtype=torch.float64, device=dev)
                dist.all_reduce(tj, op=dist.ReduceOp.MAX)
                okt = torch.tensor([ok_l], dtype=torch.int64, device=dev)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                dtj, ok_l = float(tj.item()), int(okt.item())
            g = multistart_golden()
            whole = None
            if g is not None and all(int(st) in g for st in jstarts):
                want = [g[int(st)] for st in jstarts]
                best = min(want, key=lambda w: (w["cost"], w["start"]))
                whole = (float(res_j["cost"]), int(res_j["start"]), fnv1a(res_j["path"]), int(res_j["sweeps"])) == \
                        (best["cost"], best["start"], best["fnv"], sum(w["sweeps"] for w in want))
            out_j = {"what": f"All-NN + 2OPT (h_greedy_2opt, heuristics.c:74-116) over starts 0..{len(jstarts) - 1} of the n={n} instance: a FIXED job, "
                             "start p -> rank p mod N, one batched engine call per rank, one MIN all-reduce of the packed (cost, start, rank) key + "
                             "one broadcast of the winner's tour",
                     "scaling": "strong", "starts": int(len(jstarts)), "ranks": world, "backend": backend or "none (single rank: no exchange)",
                     "seconds": dtj, "sweeps": int(res_j["sweeps"]), "value": res_j["sweeps"] * evals / dtj, "unit": "evals/s",
                     "best_cost": float(res_j["cost"]), "best_start": int(res_j["start"]),
                     "tours_in_flight_per_rank": int(len(jstarts[rank::world])),
                     "parity": {"golden": gl.get("golden"), "every_rank_shard_ok": bool(ok_l) if gl.get("golden") else None,
                                "whole_job_ok": whole, "rank0_shard": gl}}
            if gl.get("golden") and (not ok_l or whole is False):
                raise SystemExit(f"parity gate of the sharded multi-start job failed: {out_j['parity']}")
            return out_j
        job = job_leg()      # not guarded: every rank takes part in its collectives, and a wrong result must fail the run
        eng.tour_nn(0, start)        # (the job used the slots: the legs below expect the NN tour in slot 0 again)

    roof = roof_fused = None
    if rank == 0:
        roof = timed_search(eng, evals)
        roof["kernel_ms_back_to_back"] = eng.time_sweep(1, 50)   # the sweep part alone (no move applied between launches)
        if eng.info().get("persist"):
            # the one-launch-per-sweep kernel on the same workload (what runs for f64 / int32 cells, n > 4096, batches)
            eng.set_option(T.OPT_PERSIST, 0)
            try:
                t1 = time.perf_counter()
                sw1 = step()
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                roof_fused = timed_search(eng, evals)
                roof_fused["ms_per_step"] = 1e3 * dt1
                roof_fused["value"] = sw1 * evals / dt1
            finally:
                eng.set_option(T.OPT_PERSIST, args.persist)

    base = None
    if aux and args.cpu_sweeps > 0:
        base = guarded(cpu_baseline, n, seed, args.cpu_sweeps)

    # ---- the other matrix storage, same workload: the reference's own f64 cells (N = 1 only)
    other = None
    if aux and not args.no_other:
        def other_leg():
            oelem = "f64" if eng.info()["elem"] != T.ELEM_F64 else "i32"
            e2 = T.Engine(local)
            try:
                e2.set_option(T.OPT_ELEM, ELEMS[oelem]); e2.set_option(T.OPT_BATCH, args.batch)
                e2.set_points(xy); e2.build_costs()
                b2 = build_roofline(e2)
                e2.tour_nn(0, start); e2.tour_copy(1, 0); e2.tour_two_opt(1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sw2 = 0
                for _ in range(args.steps):
                    e2.tour_copy(1, 0); sw2 += e2.tour_two_opt(1)[0]
                dt2 = time.perf_counter() - t1
                p2, c2, _ = e2.tour_store(1)
                r2 = timed_search(e2, evals)
                # the regime the reference's f64 caller runs in (h_Greedy_2opt_mod_costs: all starts, batched): VERDICT r3 item 6
                b2b = guarded(batch_measure, e2, args.batch_starts) if args.batch_starts > 0 else None
                return {"matrix_elem": oelem, "batch": b2b, "value": sw2 * evals / dt2, "unit": "evals/s", "ms_per_step": 1e3 * dt2 / args.steps,
                        "final_cost": c2, "parity": parity(n, seed, sw2 // args.steps, c2, p2), "roofline": r2, "roofline_build": b2,
                        # (flat copies of the two figures round 1 printed)
                        "kernel_ms_mean": r2["kernel_ms_mean"], "bytes_per_eval": r2["bytes_per_eval"],
                        "roofline_achieved_GBs": r2["achieved"], "roofline_frac": r2["frac"]}
            finally:
                e2.close()
        other = guarded(other_leg)

    # ---- batched multi-start on the same instance (h_greedy_2opt's loop, 64 starts in flight):
    # the throughput-bound regime, next to the latency-bound single search above
    batch = None
    if aux and args.batch_starts > 0:
        def batch_leg():
            return batch_measure(eng, args.batch_starts)
        batch = guarded(batch_leg)

    # ---- the other rows of north_star's throughput table (n = 1k / 16k), each on its own engine
    sizes = None
    if aux and not args.no_sizes and n == 4096:
        sizes = {"1024": guarded(search_row, 1024, 1, max(args.steps, 5), 1, "auto", 174, False),
                 "16384": guarded(search_row, 16384, 123, 2, 1, "auto", 6, True)}

    # ---- matrix-free sweep (config 5, pla85900): VALU-bound, no HBM roofline
    otf = None
    if aux and not args.no_otf:
        def otf_leg():
            pts, ewt = read_tsplib(os.path.join(DATA, "pla85900.tsp"))
            kind = {"EUC_2D": T.EUC_2D, "ATT": T.ATT, "CEIL_2D": T.CEIL_2D}[ewt]
            e3 = T.Engine(local)
            try:
                e3.set_points(pts, kind); e3.build_costs()
                info = e3.info()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                e3.tour_nn(0, 0)
                _, c_nn, _ = e3.tour_store(0, want_path=False)
                nn_s = time.perf_counter() - t1
                ms = e3.time_sweep(0, 5)
                e3.set_option(99, 3)                       # diagnostics: the same sweep with every pair evaluated in full
                ms_full = e3.time_sweep(0, 3)
                e3.set_option(99, 0)
                # NN(0) -> 2-opt LOCAL OPTIMUM on the matrix-free engine (BASELINE config 5's local search, refinment.c:3-37;
                # tests/test_gpu_parity.py::test_pla85900_config5_local_optimum_and_vns certifies this tour with an oracle sweep)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sw3, rc3 = e3.tour_two_opt(0)
                desc_s = time.perf_counter() - t1
                _, c_opt, _ = e3.tour_store(0, want_path=False)
            finally:
                e3.close()
            m = len(pts)
            ev = T.evals_per_sweep(m)
            ceil = json.load(open(os.path.join(ROOT, "profiles", "r03_otf_isa_ceiling.json")))["3"]   # CEIL_2D, integer coordinates
            return {"workload": f"pla85900 ({ewt}, n={m}), matrix-free: no n x n array (59 GB of doubles in the reference's format)",
                    "matrix_free": info["matrix_free"], "ms_per_sweep": ms, "evals_per_sweep": ev, "value": ev / (ms * 1e-3),
                    "unit": "evals/s", "nn_tour_s": nn_s, "nn_cost": c_nn,
                    "what_value_counts": "pairs DECIDED per second (n(n-3)/2 per sweep / time) with the exact early-out of round 4: a pair can improve "
                                         "only if one of its two squared distances undercuts a squared edge length (weights are monotone in the "
                                         "squared distance), so run-, thread- and pair-level tests on f32 squared distances leave out pairs whose "
                                         "delta is provably >= 0 before the two roots; same argmin, same trajectory (oracle-certified optimum)",
                    "descent": {"what": "NN(0) -> 2-opt local optimum, all in (one call, host polls included)", "sweeps": int(sw3), "rc": int(rc3),
                                "seconds": desc_s, "evals_per_s": sw3 * ev / desc_s, "ms_per_sweep_all_in": 1e3 * desc_s / max(sw3, 1),
                                "final_cost": c_opt},
                    "full_evaluation": {"what": "the same sweep with every pair evaluated in full (option 99 = 3: round 3's kernel)",
                                        "ms_per_sweep": ms_full, "value": ev / (ms_full * 1e-3), "unit": "evals/s"},
                    "roofline": {"bound": "valu", "achieved": ev / (ms_full * 1e-3), "peak": ceil["ceiling_evals_per_s"], "unit": "evals/s",
                                 "frac": ev / (ms_full * 1e-3) / ceil["ceiling_evals_per_s"], "traffic": None,
                                 "kernel": ceil["kernel"] + " (full evaluation of every pair: the instruction stream the ceiling counts)",
                                 "issue_cycles_per_pair": ceil["issue_cycles_per_pair"],
                                 "valu_per_pair": ceil["valu_per_pair"], "clock_hz": ceil["clock_hz"],
                                 "derivation": "static VALU count of the kernel's step loop (8 pair evaluations) priced per instruction class: "
                                               "tools/otf_isa_count.py -> profiles/r03_otf_isa_ceiling.json "},
                    "early_out_speedup": ms_full / ms}
        otf = guarded(otf_leg)

    # ---- All-NN+2OPT over all 1002 starts of pr1002: the engine vs the reference on every host core of this box's share
    cpu_ms = None
    if aux and not args.no_cpu_multistart:
        def cpu_ms_leg():
            pts, _ = read_tsplib(os.path.join(DATA, "pr1002.tsp"))
            e4 = T.Engine(local)
            try:
                e4.set_points(pts); e4.build_costs()
                e4.multistart_nn_2opt(np.arange(8, dtype=np.int32))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                res = e4.multistart_nn_2opt()
                gsec = time.perf_counter() - t1
            finally:
                e4.close()
            procs = min(16, len(os.sched_getaffinity(0)))      # the box's CPU share for one GPU
            out = cpu_multistart_baseline("pr1002", procs)
            out.update({"gpu_seconds": gsec, "gpu_best_cost": res["cost"], "gpu_sweeps": int(res["sweeps"]),
                        "gpu_evals_per_s": res["sweeps"] * T.evals_per_sweep(len(pts)) / gsec,
                        "same_result": res["cost"] == out["best_cost"], "gpu_over_all_cores": out["seconds"] / gsec})
            return out
        cpu_ms = guarded(cpu_ms_leg)

    info = eng.info() if rank == 0 else None
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # ---- BASELINE config 2: fnl4461, NN(0) then iterated 2-opt to the local optimum (golden: 603 sweeps)
    cfg2 = None
    if aux and not args.no_sizes:
        def cfg2_leg():
            pts, _ = read_tsplib(os.path.join(DATA, "fnl4461.tsp"))
            g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["instances"]["fnl4461"]["two_opt"]
            e5 = T.Engine(local)
            try:
                e5.set_points(pts); e5.build_costs()
                e5.tour_nn(0, 0)
                ts = []
                for _ in range(4):
                    e5.tour_copy(1, 0); e5.tour_store(1, want_path=False)
                    t1 = time.perf_counter()
                    sw, _ = e5.tour_two_opt(1)
                    ts.append(time.perf_counter() - t1)
                p5, c5, _ = e5.tour_store(1)
                info5 = e5.info()
                m = len(pts)
                dt5 = min(ts[1:])
                return {"instance": "fnl4461", "n": m, "sweeps": int(sw), "ms_to_local_optimum": 1e3 * dt5, "us_per_sweep": 1e6 * dt5 / sw,
                        "value": T.evals_per_sweep(m) * sw / dt5, "unit": "evals/s", "kernel": kernel_name(info5),
                        "matrix_elem": NAMES[info5["elem"]], "block": info5["block"], "wgs_per_tour": info5["wgs_per_tour"],
                        "roofline_frac": T.evals_per_sweep(m) * 2 * BYTES[info5["elem"]] / (dt5 / sw) / 1e9 / HBM_PEAK_GBS,
                        "parity": {"golden": "tests/golden/golden.json fnl4461 two_opt (compiled reference)",
                                   "ok": (int(sw), float(c5), fnv1a(p5)) == (g["sweeps"], g["final_cost"], g["final_fnv"]),
                                   "sweeps": int(sw), "final_cost": float(c5)}}
            finally:
                e5.close()
        cfg2 = guarded(cfg2_leg)

    # ---- mh_TabuSearch's walk (metaheuristic.c:86-245): k iterations from the 2-opt local optimum, LDS-resident and not
    tabu = None
    if aux and not args.no_sizes:
        def tabu_leg():
            k, out = 2000, {}
            for tn in (1024, 3584, 4096):
                e4 = T.Engine(local)
                try:
                    e4.set_option(T.OPT_ELEM, T.ELEM_U16)
                    e4.set_points(reference_points(tn, 123)); e4.build_costs()
                    seed0, c0 = e4.nn_tour(0)
                    c0, _, _ = e4.two_opt(seed0)
                    row = {"iterations": k, "from": "2-opt local optimum of NN(0), uniform-random seed 123"}
                    for mode, name in ((1, "lds_resident"), (0, "sweep_apply_kernels")):
                        e4.set_option(T.OPT_PERSIST, mode)
                        ts = []
                        for _ in range(2):
                            s0 = seed0.copy()
                            t1 = time.perf_counter()
                            _, bc, fc, _ = e4.tabu_search(s0, c0, k)
                            ts.append(time.perf_counter() - t1)
                        i4 = e4.info()
                        row[name] = {"us_per_iteration": 1e6 * min(ts) / k,
                                     "kernel": ("k_lds2opt_w<.,true> (half-window rows)" if i4["persist_window"] else "k_lds2opt<.,true>") if i4["persist"] else "k_sweep_*<TABU> + k_apply",
                                     "best_cost": bc, "final_cost": fc}
                    row["same_walk"] = row["lds_resident"]["final_cost"] == row["sweep_apply_kernels"]["final_cost"]
                    out[str(tn)] = row
                finally:
                    e4.close()
            return out
        tabu = guarded(tabu_leg)

    # ---- mh_VNS's loop (metaheuristic.c:279-318): local search + kicks, resident in the LDS kernels and with the kicks on the host
    vns = None
    if aux and not args.no_sizes:
        def vns_leg():
            out = {"what": "iterations of { ref_2opt, incumbent, rand() % 9 - 2 kicks } from the 2-opt local optimum of NN(0); the random "
                           "numbers are glibc rand() values after srand(1), drawn on the host before the first GPU call",
                   "note": "an iteration is ~10 sweeps (2.3 kicks on average, each repaired by several long reversals), not the ~5 of a "
                           "single-kick descent"}
            for name, pts in (("pr1002", read_tsplib(os.path.join(DATA, "pr1002.tsp"))[0]), ("n4096_s123", reference_points(4096, 123)),
                              ("fnl4461", read_tsplib(os.path.join(DATA, "fnl4461.tsp"))[0])):
                e6 = T.Engine(local)
                try:
                    e6.set_points(pts); e6.build_costs()
                    seed0, c0 = e6.nn_tour(0)
                    c0, _, _ = e6.two_opt(seed0)
                    row = {}
                    for mode, label, kk in ((1, "resident", 1000), (0, "host_kicks", 100)):
                        e6.set_option(T.OPT_PERSIST, mode)
                        ts = []
                        for _ in range(2):
                            p6, b6 = seed0.copy(), seed0.copy()
                            t1 = time.perf_counter()
                            r6 = e6.vns_search(p6, kk, _VNS_RAND, b6, c0)
                            ts.append(time.perf_counter() - t1)
                        i6 = e6.info()
                        row[label] = {"iterations": kk, "us_per_iteration": 1e6 * min(ts) / kk, "best_cost": r6["best_cost"],
                                      "rand_values_consumed": r6["consumed"], "rc": r6["rc"],
                                      "kernel": ("k_lds2opt_w" if i6["persist_window"] else "k_lds2opt") + " (whole loop in one launch)" if i6["persist"]
                                                else "one device local search per iteration, kicks on the host"}
                        if i6["persist"]:
                            row[label].update({"sweeps": i6["persist_sweeps"], "sweeps_per_iteration": i6["persist_sweeps"] / kk,
                                               "us_per_sweep_all_in": 1e6 * min(ts) / max(i6["persist_sweeps"], 1)})
                    out[name] = row
                finally:
                    e6.close()
            return out
        vns = guarded(vns_leg)

    # ---- the drop-in binary: C host layer, multi-start sharded in C over the N devices of this run, RCCL exchange.
    # A child process (its own HIP context); at N > 1 the other ranks have left and released their devices.
    host_c = None
    if not args.no_host_c:
        def host_c_leg():
            tsp = os.path.join(ROOT, "travellingsalesmanoptimization_amd", "host", "tsp")
            ndev = min(world, int(T._lib.load().tspgpu_device_count()))
            env = dict(os.environ, TSP_GPU_DEVICES=",".join(str(i) for i in range(ndev)), TSP_GPU_EXCHANGE="rccl", TSP_GPU_STATS="1")
            out = {"devices": ndev, "exchange": "rccl (ncclAllReduce(ncclMin, int64) + ncclBroadcast, one process, one thread per device)"}
            for key, argv, tmo in [("pr1002_all_starts", ["-f", os.path.join(DATA, "pr1002.tsp"), "-alg", "2OPT_GREEDY", "-q"], 180),
                                   ("d18512_t10", ["-f", os.path.join(DATA, "d18512.tsp"), "-alg", "2OPT_GREEDY", "-t", "10", "-q"], 240)]:
                t1 = time.perf_counter()
                r = subprocess.run([tsp, *argv], capture_output=True, text=True, timeout=tmo, env=env, cwd=ROOT)
                wall = time.perf_counter() - t1
                st = [json.loads(l.split("tspgpu-stats:", 1)[1]) for l in r.stderr.splitlines() if l.startswith("tspgpu-stats:")]
                nn_ = 1002 if key.startswith("pr1002") else 18512
                row = {"rc": r.returncode, "stdout": r.stdout.strip()[:80], "process_wall_s": wall}
                if st:
                    row.update(st[-1])
                    row["evals_per_s"] = st[-1]["sweeps"] * T.evals_per_sweep(nn_) / max(st[-1]["seconds"], 1e-9)
                else:
                    row["stderr_tail"] = r.stderr[-300:]
                out[key] = row
            out["pr1002_all_starts"]["golden_cost_266290"] = out["pr1002_all_starts"].get("stdout") == "Cost: 266290.00"   # whole stdout
            # BASELINE config 5 as it is named, end to end through the reference's CLI with no time limit: pla85900, VNS, on-the-fly
            # distances -- All-NN over the 85 900 starts, then 3 iterations of { 2-opt to the local optimum, kicks }
            # (tests/test_host_c.py::test_config5_vns_pla85900_end_to_end_through_the_host_layer certifies the incumbent)
            t1 = time.perf_counter()
            r = subprocess.run([tsp, "-f", os.path.join(DATA, "pla85900.tsp"), "-alg", "VNS", "-k", "3", "-q"], capture_output=True, text=True,
                               timeout=400, env=dict(env, TSP_ALLOW_EXT="1"), cwd=ROOT)
            out["config5_pla85900_vns_k3"] = {"rc": r.returncode, "stdout": r.stdout.strip()[:80], "process_wall_s": time.perf_counter() - t1,
                                              "what": "All-NN (85 900 NN tours) + 3 VNS iterations over on-the-fly distances, no -t"}
            return out
        host_c = guarded(host_c_leg)

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
                               + ("cost matrix built in HBM and held in LDS (rows in tour order) for the whole descent, "
                                  if info_main.get("persist") else "cost matrix resident in HBM, ")
                               + "NN(start=rank) tour -> best-improvement 2-opt to the local optimum; one step = one full local search",
                   "n": n, "seed": seed, "evals_per_sweep": evals,
                   "sweeps_per_step_rank0": my_sweeps // max(args.steps, 1),
                   "matrix_elem": {T.ELEM_U16: "uint16 exact copy", T.ELEM_I32: "int32 exact copy", T.ELEM_F64: "f64"}[info["elem"]],
                   "sweep_kernel": info["kernel"], "wgs_per_tour": info["wgs_per_tour"],
                   "block": info["block"], "lds_bytes": info["lds_bytes"], "batch": args.batch,
                   "descent": ({"kernel": "k_lds2opt_w" if info_main.get("persist_window") else "k_lds2opt", "launches_per_descent": 1,
                                "workgroups": info_main["persist_wgs"], "edges_per_workgroup": info_main["persist_edges"],
                                "lds_bytes": info_main["persist_lds"], "block": 768 if info_main.get("persist_window") else 512}
                               if info_main.get("persist") else {"kernel": "one launch per sweep"}),
                   "parallelism": f"multistart-shard{world}"},
        "wall_clock_to_local_optimum_ms": 1e3 * tmax / args.steps,
        "final_cost_rank0": final_cost, "nn_cost_rank0": nn_cost, "parity": gate,
        "matrix_build_ms": broof["kernel_ms_mean"], "nn_tour_ms": nn_ms,
        "roofline": roof, "roofline_one_launch_per_sweep": roof_fused, "roofline_build": broof, "cpu_baseline": base,
        "other_matrix_storage": other, "sizes": sizes, "multistart_batch": batch, "multistart_job": job, "otf": otf, "cpu_multistart_baseline": cpu_ms, "host_c_path": host_c,
        "tabu_walk": tabu, "vns_walk": vns, "config2_fnl4461": cfg2,
    }
    if base and "value" in base:
        out["gpu_over_cpu"] = out["value"] / base["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
