"""The C host layer (travellingsalesmanoptimization_amd/host): the reference's own entry
points and CLI over the gfx950 engine.  CPU part: command-line parsing (the cases of the
reference's test/main.c:13-69, which no longer compile there) and loud failure without a
GPU.  GPU part: the `tsp` binary's -q stdout contract against the golden costs, the way
scripts/compare_algs.py:67-72 drives the reference."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "travellingsalesmanoptimization_amd", "host")
TSP = os.path.join(HOST, "tsp")
DATA = os.path.join(ROOT, "tests", "golden", "data")


class Timespec(C.Structure):
    _fields_ = [("tv_sec", C.c_long), ("tv_nsec", C.c_long)]


class Solution(C.Structure):   # utils.h:42-47
    _fields_ = [("cost", C.c_double), ("path", C.POINTER(C.c_int)), ("ncomp", C.c_int), ("comp", C.POINTER(C.c_int))]


class Options(C.Structure):    # tsp.h:66-103
    _fields_ = [("timelimit", C.c_double), ("seed", C.c_int), ("graph_random", C.c_bool), ("graph_input", C.c_bool),
                ("inputfile", C.c_char_p), ("tofile", C.c_bool), ("k", C.c_int), ("policy", C.c_int),
                ("mileage_init", C.c_int), ("bl_patching", C.c_bool), ("init_mip", C.c_bool), ("skip_policy", C.c_int),
                ("callback_relaxation", C.c_bool), ("modified_costs", C.c_bool), ("hf_prob", C.c_double),
                ("lb_dynk", C.c_bool), ("lb_initk", C.c_int), ("lb_improv", C.c_double), ("lb_delta", C.c_int),
                ("lb_kstar", C.c_bool)]


class Instance(C.Structure):   # tsp.h:115-134
    _fields_ = [("alg", C.c_int), ("nnodes", C.c_int), ("c", Timespec), ("points", C.c_void_p), ("costs", C.c_void_p),
                ("best_solution", Solution), ("starting_node", C.c_int), ("threads_seeds", C.c_void_p),
                ("ncols", C.c_int), ("cplex_terminate", C.c_int)]


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(os.path.join(HOST, "libtsphost.so")):
        subprocess.run(["make", "-s", "-C", HOST], check=True)
    return C.CDLL(os.path.join(HOST, "libtsphost.so"))


def parse(host, *args):
    argv = (C.c_char_p * (len(args) + 1))(b"tsp", *[a.encode() for a in args])
    rc = host.tsp_parse_commandline(len(args) + 1, argv)
    return rc, Options.in_dll(host, "tsp_env"), Instance.in_dll(host, "tsp_inst")


def test_struct_layout_matches_reference_abi(host):
    """x86-64 layout of the reference's structs (what code compiled against its headers expects)"""
    # sizes printed by a program compiled against the reference headers: 32, 88, 96
    assert C.sizeof(Solution) == 32 and C.sizeof(Options) == 88 and C.sizeof(Instance) == 96


def test_cli_defaults_and_flags(host):
    rc, env, inst = parse(host, "-n", "40", "-q")
    assert rc == 0 and inst.nnodes == 40 and env.graph_random and not env.graph_input
    assert env.timelimit == -1.0 and env.seed == -1 and env.k == 2147483647 and env.policy == 3  # tsp.c:6-44
    assert inst.alg == 0 and inst.starting_node == 0

    rc, env, inst = parse(host, "-file", os.path.join(DATA, "berlin52.tsp"), "-q")      # test/main.c "-file"
    assert env.graph_input and env.inputfile.decode().endswith("berlin52.tsp")
    rc, env, inst = parse(host, "-n", "10", "-time", "12.5", "-q")                        # test/main.c "-time"
    assert env.timelimit == 12.5
    rc, env, inst = parse(host, "-n", "10", "-seed", "123", "-q")                         # test/main.c "-seed"
    assert env.seed == 123
    rc, env, inst = parse(host, "-n", "10", "-t", "-3", "-q")
    assert env.timelimit == -1.0                                                          # negative time ignored
    for name, alg in [("GREEDY", 0), ("GREEDY_ITER", 1), ("2OPT_GREEDY", 2), ("TABU_SEARCH", 3), ("VNS", 4),
                      ("CPLEX_NOSEC", 5), ("CPLEX_BENDERS", 6), ("EXTRA_MILEAGE", 7), ("CPLEX_BRANCH_CUT", 9),
                      ("HARD_FIXING", 10), ("LOCAL_BRANCHING", 11)]:
        rc, env, inst = parse(host, "-n", "10", "-alg", name, "-q")
        assert inst.alg == alg
    rc, env, inst = parse(host, "-n", "10", "-k", "200", "-skip", "2", "-q")
    assert env.k == 2                                                                     # sic, tsp.c:237
    rc, env, inst = parse(host, "-f", os.path.join(DATA, "berlin52.tsp"), "-n", "77", "-q")
    assert env.graph_input and not env.graph_random and inst.nnodes != 77                 # -n ignored after -f
    rc, env, inst = parse(host, "-n", "10", "--no_patching", "--no_relax", "--modify_costs", "--lb_dynk",
                          "--lb_kstar", "-hf_prob", "0.4", "-lb_initk", "20", "-lb_delta", "7", "-lb_improv", "0.5",
                          "-em", "RANDOM", "--to_file", "-q")
    assert (not env.bl_patching and not env.callback_relaxation and env.modified_costs and env.lb_dynk and
            env.lb_kstar and env.hf_prob == 0.4 and env.lb_initk == 20 and env.lb_delta == 7 and
            env.lb_improv == 0.5 and env.mileage_init == 1 and env.tofile)


def test_binary_help_and_missing_gpu():
    import torch
    r = subprocess.run([TSP, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "-alg" in r.stdout
    r = subprocess.run([TSP, "--all_algs"], capture_output=True, text=True)
    assert "2OPT_GREEDY" in r.stdout and "TABU_SEARCH" in r.stdout
    r = subprocess.run([TSP, "-f", "/nonexistent.tsp"], capture_output=True, text=True)
    assert r.returncode == 1
    if not torch.cuda.is_available():
        r = subprocess.run([TSP, "-f", os.path.join(DATA, "berlin52.tsp"), "-alg", "GREEDY"], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr   # loud, no silent host path


def run_q(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([TSP, *args, "-q"], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    return r.returncode, r.stdout.strip(), r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["berlin52", "eil51", "kroA100", "pr1002"])
def test_binary_greedy_and_iter(golden, name):
    for alg, key in [("GREEDY", "greedy"), ("GREEDY_ITER", "greedy_iter")]:
        rc, out, err = run_q("-f", os.path.join(DATA, name + ".tsp"), "-alg", alg)
        assert rc == 0, err
        assert out == "Cost: %.2f" % golden["algs"][f"{name}_{key}"]["cost"]   # compare_algs.py:72 scrapes this


PUBLISHED = json.load(open(os.path.join(ROOT, "tests", "golden", "published_heuristics_ric.json")))["instances"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PUBLISHED))
def test_binary_published_nn_columns(name):
    """what scripts/compare_algs.py:67-72 did to fill results/heuristics-ric.csv (columns NN, allNN): `tsp -f <instance>
    -alg GREEDY | GREEDY_ITER -q` and the number behind "Cost:" -- all 28 published values, from the drop-in binary"""
    for alg, col in [("GREEDY", "NN"), ("GREEDY_ITER", "allNN")]:
        rc, out, err = run_q("-f", os.path.join(DATA, name + ".tsp"), "-alg", alg)
        assert rc == 0, err
        assert out == "Cost: %.2f" % PUBLISHED[name][col]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["berlin52", "eil51", "kroA100", "pr1002"])
def test_binary_2opt_greedy(golden, name):
    rc, out, err = run_q("-f", os.path.join(DATA, name + ".tsp"), "-alg", "2OPT_GREEDY")
    assert rc == 0, err
    assert out == "Cost: %.2f" % golden["algs"][f"{name}_2opt_greedy"]["cost"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["berlin52", "eil51", "kroA100"])
def test_binary_tabu_and_vns(golden, name, tmp_path):
    os.makedirs(os.path.join(ROOT, "results"), exist_ok=True)
    rc, out, err = run_q("-f", os.path.join(DATA, name + ".tsp"), "-alg", "TABU_SEARCH", "-k", "200")
    assert rc == 0 and out == "Cost: %.2f" % golden["algs"][f"{name}_tabu_k200"]["cost"], err
    lines = open(os.path.join(ROOT, "results", "TabuResults.dat")).read().split()
    assert len(lines) == 200 and lines[0].startswith("0,")                       # metaheuristic.c:165
    rc, out, err = run_q("-f", os.path.join(DATA, name + ".tsp"), "-alg", "VNS", "-k", "200")
    assert rc == 0 and out == "Cost: %.2f" % golden["algs"][f"{name}_vns_k200"]["cost"], err


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["berlin52", "eil51", "kroA100"])
def test_binary_tabu_and_vns_matrix_free(golden, name):
    """the same runs with TSP_MATRIX_FREE=1 -- BASELINE config 5's engine (no n x n matrix, tsp_inst.costs NULL, every local
    search in the on-the-fly sweep) under mh_TabuSearch and mh_VNS -- against the compiled reference's goldens"""
    os.makedirs(os.path.join(ROOT, "results"), exist_ok=True)
    env = {"TSP_MATRIX_FREE": "1"}
    rc, out, err = run_q("-f", os.path.join(DATA, name + ".tsp"), "-alg", "TABU_SEARCH", "-k", "200", env=env)
    assert rc == 0 and out == "Cost: %.2f" % golden["algs"][f"{name}_tabu_k200"]["cost"], err
    rc, out, err = run_q("-f", os.path.join(DATA, name + ".tsp"), "-alg", "VNS", "-k", "200", env=env)
    assert rc == 0 and out == "Cost: %.2f" % golden["algs"][f"{name}_vns_k200"]["cost"], err


@pytest.mark.gpu
def test_binary_random_instance_and_deadline(golden):
    rc, out, err = run_q("-n", "1000", "-seed", "123", "-alg", "GREEDY_ITER")
    assert rc == 0 and out == "Cost: %.2f" % golden["algs"]["n1000_s123_greedy_iter"]["cost"], err
    # a time limit ends the run with a valid incumbent and exit status 0 (DEADLINE_EXCEEDED is a success)
    rc, out, err = run_q("-f", os.path.join(DATA, "fnl4461.tsp"), "-alg", "2OPT_GREEDY", "-t", "2")
    assert rc == 0 and out.startswith("Cost: ")
    assert 182566 <= float(out.split(":")[1]) < 229963      # between the optimum and NN(0)


@pytest.mark.gpu
def test_binary_rejects_att_like_the_reference():
    rc, out, err = run_q("-f", os.path.join(DATA, "att48.tsp"), "-alg", "GREEDY")
    assert rc == 1                                           # tsp.c:576-584
    rc, out, err = run_q("-f", os.path.join(DATA, "att48.tsp"), "-alg", "2OPT_GREEDY", env={"TSP_ALLOW_EXT": "1"})
    assert rc == 0 and float(out.split(":")[1]) >= 10628    # TSPLIB optimum of att48 as a bound


@pytest.mark.gpu
def test_binary_matrix_free_pla85900():
    """config 5 through the reference's CLI: 85 900 nodes, CEIL_2D, no n x n matrix anywhere
    (59 GB in the reference's format, which also overflows its int indices)"""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_large.json")))["pla85900"]
    rc, out, err = run_q("-f", os.path.join(DATA, "pla85900.tsp"), "-alg", "GREEDY", env={"TSP_ALLOW_EXT": "1"})
    assert rc == 0 and out == "Cost: %.2f" % g["nn_cost"], err
    # and on a small instance the matrix-free binary equals the matrix binary
    rc, out, err = run_q("-f", os.path.join(DATA, "kroA100.tsp"), "-alg", "2OPT_GREEDY", env={"TSP_MATRIX_FREE": "1"})
    assert rc == 0 and out == "Cost: 21360.00", err


@pytest.mark.gpu
def test_binary_vns_pla85900_honours_deadline():
    """config 5 as BASELINE names it (VNS over the matrix-free engine) under the reference's
    cooperative time limit: All-NN is cut at the deadline (heuristics.c:43-49 checks before every
    start), the result is a valid tour no worse than NN(0), and the process ends soon after -t"""
    import json, time
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_large.json")))["pla85900"]
    t0 = time.time()
    rc, out, err = run_q("-f", os.path.join(DATA, "pla85900.tsp"), "-alg", "VNS", "-k", "1", "-t", "12", env={"TSP_ALLOW_EXT": "1"})
    dt = time.time() - t0
    assert rc == 0 and out.startswith("Cost: "), err
    assert float(out.split(":")[1]) <= g["nn_cost"]
    assert dt < 60, dt


@pytest.mark.gpu
def test_config5_vns_pla85900_end_to_end_through_the_host_layer(O, tmp_path):
    """BASELINE config 5 as it is named -- pla85900.tsp, VNS (3-opt kick + GPU 2-opt), on-the-fly distances -- END TO END
    through the reference's own entry point (tsp_run_algorithm -> mh_VNS, metaheuristic.c:251-341) with NO time limit: All-NN
    over the 85 900 starts (h_Greedy_iterative, ~10 s), then k = 3 iterations of { ref_2opt to the local optimum, incumbent,
    kicks on the program's glibc stream }.  Round 3 could only watch this run being cut inside All-NN by its deadline; with
    the matrix-free early-out a descent takes ~3 s.  Checked: exit state T_OK, the incumbent is a valid tour whose cost is the
    oracle's recomputation, no worse than the All-NN tour, and -- an incumbent of mh_VNS is always the result of a ref_2opt
    -- a 2-opt LOCAL OPTIMUM by one full oracle sweep; and the `tsp` binary prints the same cost (deterministic: the
    private glibc stream of a fresh process)."""
    import numpy as np
    out = str(tmp_path / "best.npy")
    code = f"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {os.path.join(ROOT, "tests")!r})
os.environ["TSP_ALLOW_EXT"] = "1"
from test_host_c import Instance
host = C.CDLL({os.path.join(HOST, "libtsphost.so")!r})
argv = [b"tsp", b"-f", {os.path.join(DATA, "pla85900.tsp")!r}.encode(), b"-alg", b"VNS", b"-k", b"3", b"-q"]
arr = (C.c_char_p * len(argv))(*argv)
assert host.tsp_parse_commandline(len(argv), arr) == 0
host.tsp_read_input()
inst = Instance.in_dll(host, "tsp_inst")
host.utils_startclock(C.byref(inst, Instance.c.offset))
os.makedirs("results", exist_ok=True)
rc = host.tsp_run_algorithm()
n = inst.nnodes
path = np.ctypeslib.as_array(inst.best_solution.path, shape=(n,)).copy()
np.save({out!r}, path)
print("RESULT", rc, n, repr(inst.best_solution.cost), bool(inst.costs))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    rc, n, cost, has_matrix = int(line[1]), int(line[2]), float(line[3]), line[4] == "True"
    assert rc == 0 and n == 85900 and not has_matrix             # T_OK; no n x n matrix anywhere
    path = np.load(out)
    xy, ewt = O.read_tsplib(os.path.join(DATA, "pla85900.tsp"))
    assert ewt == "CEIL_2D" and O.valid_tour(path) and O.tour_cost_xy(xy, O.CEIL_2D, path) == cost
    assert cost < 162673661.0                                    # better than the best NN tour (All-NN: start 43632)
    d, mv = O.two_opt_best_move_xy(xy, O.CEIL_2D, path, threads=16)
    assert d >= -1e-7, (d, mv)                                   # a 2-opt local optimum by the reference's own scan
    rc2, out2, err2 = run_q("-f", os.path.join(DATA, "pla85900.tsp"), "-alg", "VNS", "-k", "3", env={"TSP_ALLOW_EXT": "1"})
    assert rc2 == 0 and out2 == "Cost: %.2f" % cost, (out2, err2[-500:])


@pytest.mark.gpu
def test_bench_line_contract():
    """bench.py prints ONE JSON line carrying the driver's contract fields, the roofline object and
    the parity gate (final cost of the headline workload)"""
    import json, subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sweeps", "3",
                          "--no-other", "--batch-starts", "8", "--job-starts", "16", "--no-sizes", "--no-otf", "--no-cpu-multistart", "--no-host-c"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    # roofline.frac = algorithmic bytes / time / 8 TB/s, also for the LDS-resident headline descent;
    # what actually limits a sweep there -- the grid-wide exchange -- is named beside it, with the exchange-floor ratio secondary
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_mean"] * 1e-3) / 1e9) < 1e-6 and 0.3 < r["frac"] < 1
    assert r["limited_by"].startswith("latency") and abs(r["exchange_floor_frac"] - r["exchange_floor_us"] / r["us_per_sweep"]) < 1e-9
    assert abs(r["us_per_sweep"] - 1e3 * r["kernel_ms_mean"]) < 1e-9 and r["frac_nominal_hbm"] == r["frac"]
    ph = r["phase_us"]
    assert set(ph) == {"evaluation", "workgroup_reduction", "exchange", "reversal", "rows_fetched", "decode_and_swaps"}
    assert 0.5 * r["us_per_sweep"] < sum(ph.values()) < 1.5 * r["us_per_sweep"] and ph["exchange"] > ph["evaluation"] > 0
    assert 0 < r["lds_frac"] < 1
    assert "held in LDS" in d["config"]["workload"] and "resident in HBM" not in d["config"]["workload"]
    assert d["final_cost_rank0"] == 488522.0 and d["config"]["sweeps_per_step_rank0"] == 609
    assert d["parity"]["ok"] is True and d["parity"]["final_fnv"]          # the in-run gate against the committed golden
    assert r["traffic"] is None and "traffic_from_committed_profile" in r   # PMC bytes are not measured by the run itself
    b = d["roofline_build"]
    assert b["bound"] == "hbm" and b["kernel"].startswith("k_build_costs") and abs(b["frac"] - b["achieved"] / 8000.0) < 1e-9
    for k in ("sizes", "otf", "cpu_multistart_baseline", "host_c_path", "other_matrix_storage", "multistart_batch", "multistart_job", "tabu_walk",
              "vns_walk", "config2_fnl4461"):
        assert k in d
    # the throughput-regime legs carry their own gates against the compiled reference's per-start goldens (VERDICT r3 missing 4, 5)
    mb = d["multistart_batch"]
    assert mb["parity"]["ok"] is True and mb["parity"]["per_start_ok"] == 8 and mb["starts"] == 8
    assert mb["roofline"]["bound"] == "hbm" and mb["roofline"]["kernel"].startswith("k_sweep_pipe") and mb["roofline"]["kernel_launches_timed"] > 500
    assert abs(mb["roofline"]["frac"] - mb["roofline"]["achieved"] / 8000.0) < 1e-9
    mj = d["multistart_job"]
    assert mj["ranks"] == 1 and mj["starts"] == 16 and mj["scaling"] == "strong" and mj["parity"]["whole_job_ok"] is True
    assert mj["parity"]["every_rank_shard_ok"] is True and abs(mj["value"] - mj["sweeps"] * 8382464 / mj["seconds"]) / mj["value"] < 1e-9
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert abs(d["value"] - 609 * 8382464 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["config"]["descent"]["kernel"] == "k_lds2opt" and d["config"]["descent"]["workgroups"] == 256
    assert r["kernel"].startswith("k_lds2opt") and r["sweeps_timed"] == 609 and r["kernel_launches_timed"] == 1
    # the streamed one-launch-per-sweep kernel on the same workload IS HBM-bound (PMC traffic ~ algorithmic bytes): bound "hbm"
    f = d["roofline_one_launch_per_sweep"]
    assert f["bound"] == "hbm" and f["peak"] == 8000.0 and abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-9 and f["unit"] == "GB/s"
    assert f["kernel"].startswith("k_sweep_fused") and f["kernel_launches_timed"] >= 576 and f["kernel_ms_mean"] > r["kernel_ms_mean"]


@pytest.mark.gpu
def test_bench_two_ranks_gloo_on_one_gpu():
    """the N > 1 branch of bench.py (torch.distributed launch, barrier, per-step exchange, MAX over ranks,
    SUM of sweeps, rank 0's host_c_path leg) rehearsed as two gloo ranks sharing device 0 -- collectives on CPU tensors,
    TSPGPU_BENCH_BACKEND=gloo -- so that the driver's first multi-GPU run cannot die on an untested branch.  No scaling
    number is taken from this (two processes on one GPU)."""
    import json, subprocess, sys, socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, TSPGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--size", "1024", "--seed", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "multistart-shard2"
    assert d["parity"]["ok"] is True                         # rank 0 searched from NN(0): the golden of n1024_s1
    ev = 1024 * 1021 // 2
    tot = d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] / ev        # sweeps summed over both ranks
    assert abs(tot - round(tot)) < 1e-3 and round(tot) >= 2 * 2 * 100   # two ranks x two steps x (> 100 sweeps each)
    assert d["config"]["sweeps_per_step_rank0"] == 174
    hc = d["host_c_path"]
    assert hc["devices"] == 1 and hc["pr1002_all_starts"]["golden_cost_266290"] is True


@pytest.mark.gpu
def test_bench_job_two_ranks_gloo_gate():
    """the fixed-size sharded multi-start job of bench.py (`multistart_job`: start p -> rank p mod N, one
    batched engine call per rank, ONE MIN all-reduce + ONE broadcast) launched exactly as the driver launches N = 2, as two
    gloo ranks sharing device 0: both shards and the whole job equal the compiled reference's per-start goldens, `ranks` is
    what torch.distributed saw.  No scaling number is taken from this (two processes on one GPU)."""
    import json, subprocess, sys, socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, TSPGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--lean", "--job-starts", "64"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    mj = d["multistart_job"]
    assert d["n_gpus"] == 2 and mj["ranks"] == 2 and mj["backend"] == "gloo" and mj["starts"] == 64 and mj["tours_in_flight_per_rank"] == 32
    assert mj["parity"]["whole_job_ok"] is True and mj["parity"]["every_rank_shard_ok"] is True
    assert (mj["best_cost"], mj["best_start"]) == (483772.0, 4)
    assert d["parity"]["ok"] is True                          # rank 0's headline descent from NN(0): 609 sweeps -> 488522


def _mod_costs_matrix(c, seed):
    """c[i][j] * (1 - x*_ij), symmetric, diagonal 0: the matrix cplex_model.c:1176-1258 builds per callback
    (same construction as oracle/make_golden_slow.py mod_costs_threads)"""
    import numpy as np
    n = c.shape[0]
    r = np.random.default_rng(seed)
    x = np.triu(r.random((n, n)), 1)
    x = x + x.T
    mc = c * (1.0 - x)
    np.fill_diagonal(mc, 0.0)
    return np.ascontiguousarray(mc)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kroA100", "n200_s3"])
def test_mod_costs_concurrent_threads(host, O, instances, name):
    """h_Greedy_2opt_mod_costs is the one entry point of the path that CPLEX enters from several worker
    threads at once, each with its own matrix and solution (cplex_model.c:1176-1258, <= 32 threads,
    cplex_model.h:12).  Four threads through libtsphost.so (ctypes drops the GIL), two rounds with the SAME
    buffers refilled in between (a pointer says nothing about the contents): every result equals the
    reference's for that matrix bit for bit, and a thread's device context dies with the thread."""
    import json
    import threading
    import numpy as np
    g = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "golden_mod_costs_threads.json")))["cases"]
         if c["instance"] == name]
    assert len(g) == 4
    c = instances(name)[1]
    n = c.shape[0]
    host.tsp_init()
    host.err_setverbosity(0)
    inst = Instance.in_dll(host, "tsp_inst")
    env = Options.in_dll(host, "tsp_env")
    inst.nnodes = n
    env.timelimit = -1.0
    host.utils_startclock(C.byref(inst, Instance.c.offset))
    host.h_Greedy_2opt_mod_costs.argtypes = [C.POINTER(Solution), C.c_void_p]
    host.tsp_gpu_thread_contexts.restype = C.c_int
    bufs = [np.empty((n, n), dtype=np.float64) for _ in g]          # one caller buffer per thread, reused
    out = {}

    def work(slot, case):
        bufs[slot][:] = _mod_costs_matrix(c, case["seed"])
        path = (C.c_int * n)()
        sol = Solution(0.0, C.cast(path, C.POINTER(C.c_int)), 0, None)
        rc = host.h_Greedy_2opt_mod_costs(C.byref(sol), bufs[slot].ctypes.data)
        out[(slot, case["seed"])] = (rc, sol.cost, np.array(path[:], dtype=np.int32), host.tsp_gpu_thread_contexts())

    for rnd in range(2):
        order = g if rnd == 0 else g[1:] + g[:1]                     # round 2: every buffer gets another matrix
        ts = [threading.Thread(target=work, args=(i, case)) for i, case in enumerate(order)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i, case in enumerate(order):
            rc, cost, path, live = out[(i, case["seed"])]
            assert rc == 0 and 1 <= live <= 4
            assert float(cost).hex() == case["cost_hex"], (rnd, case, cost)
            assert "%016x" % O.fnv1a(path) == case["fnv"]
        assert host.tsp_gpu_thread_contexts() == 0                   # destroyed at thread exit
    host.tsp_gpu_release()


@pytest.mark.parametrize("n", [52, 1002, 85900, 85902])
def test_vns_kick_matches_the_oracle_at_any_size(host, O, n):
    """vns_kick (metaheuristic.c:344-409) is host code on the glibc rand() stream: the host layer's kick against the
    oracle's restatement (itself pinned to the reference's kicks on berlin52 / kroA100 / pr1002) on cycles of config 5's
    size -- 85 900 nodes, and 85 902, where the reference's out-of-range read tour[n] meets the other malloc padding
    (4n + 8 a multiple of 16)"""
    import numpy as np
    host.tsp_init()
    host.err_setverbosity(0)
    inst = Instance.in_dll(host, "tsp_inst")
    inst.nnodes = n
    rng = np.random.default_rng(n)
    perm = rng.permutation(n).astype(np.int32)
    succ = np.empty(n, dtype=np.int32)
    succ[perm] = np.roll(perm, -1)                       # one random n-cycle
    want = succ.copy()
    O.libc_srand(77)
    for _ in range(25):
        O.vns_kick(want)
    got = succ.copy()
    host.tsp_srand.argtypes = [C.c_uint]
    host.tsp_srand(77)
    host.vns_kick.argtypes = [C.POINTER(Solution)]
    sol = Solution(0.0, got.ctypes.data_as(C.POINTER(C.c_int)), 0, None)
    for _ in range(25):
        assert host.vns_kick(C.byref(sol)) == 0
    assert np.array_equal(got, want) and O.valid_tour(got)


def kick_oob_cycle(n):
    """the seeded random cycle of tests/golden/golden_kick_oob.json (its "cycle" entry; oracle/ref_kick_probe.c)"""
    import numpy as np
    order, x, m = list(range(n)), (0x9E3779B97F4A7C15 ^ n) & (2 ** 64 - 1), 2 ** 64 - 1
    for i in range(n - 1, 0, -1):
        x = (x * 6364136223846793005 + 1442695040888963407) & m
        j = (x >> 33) % (i + 1)
        order[i], order[j] = order[j], order[i]
    o = np.array(order, dtype=np.int32)
    succ = np.empty(n, dtype=np.int32)
    succ[o] = np.roll(o, -1)
    return succ


KICK_OOB = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_kick_oob.json")))


@pytest.mark.parametrize("case", [c for c in KICK_OOB["cases"] if c["mode"] == "warm"], ids=lambda c: str(c["n"]))
def test_vns_kick_against_the_compiled_reference_at_large_n(host, O, case):
    """the model of vns_kick's out-of-range probes (metaheuristic.c:372) -- tour[-1] reads 0, tour[n]
    reads 0 unless n % 4 == 2, where it reads the size field of the next heap chunk, which no draw matches -- settled
    against the COMPILED REFERENCE run in a process of its own at n = 52 ... 85 902 (oracle/ref_kick_probe.c,
    oracle/make_golden_kick_oob.py): 25 kicks after srand(77) on the fixture's cycle leave the same tour in the
    reference, in the oracle and in the host layer; and the fixture records what the reference found at the two
    addresses (never an mmapped block: glibc's dynamic threshold has moved past 4n by the first kick)"""
    import numpy as np
    n = case["n"]
    succ = kick_oob_cycle(n)
    assert f"{O.fnv1a(succ):016x}" == case["fnv_before"]
    want = succ.copy()
    O.libc_srand(KICK_OOB["seed"])
    for _ in range(KICK_OOB["kicks"]):
        O.vns_kick(want)
    assert f"{O.fnv1a(want):016x}" == case["fnv"]                    # oracle == compiled reference
    host.tsp_init()
    host.err_setverbosity(0)
    Instance.in_dll(host, "tsp_inst").nnodes = n
    got = succ.copy()
    host.tsp_srand.argtypes = [C.c_uint]
    host.tsp_srand(KICK_OOB["seed"])
    host.vns_kick.argtypes = [C.POINTER(Solution)]
    sol = Solution(0.0, got.ctypes.data_as(C.POINTER(C.c_int)), 0, None)
    for _ in range(KICK_OOB["kicks"]):
        assert host.vns_kick(C.byref(sol)) == 0
    assert f"{O.fnv1a(got):016x}" == case["fnv"]                     # host layer == compiled reference
    # what the reference's process found at the two addresses: the model's premises
    assert case["before_values"] == [0] and not case["any_mmapped"]
    if n % 4 == 2:
        assert case["after_min"] > 0                                 # a chunk size field (odd: PREV_INUSE), never calloc padding
    else:
        assert (case["after_min"], case["after_max"]) == (0, 0)


# ------------------------------------------------------------------ the reference's signatures, one by one (VERDICT r2 "weak" 1)
class TabuSearch(C.Structure):   # tsp.h:105-113
    _fields_ = [("tenure", C.c_int), ("max_tenure", C.c_int), ("min_tenure", C.c_int), ("increment", C.c_bool),
                ("tabu_list", C.POINTER(C.c_int))]


def test_host_make_move_and_reverse_path_against_the_compiled_reference(host):
    """tabu_make_move cases 1-7 (metaheuristic.c:425-507: 1-3 single reversals, 4-6 the reference's two-step sequences with
    their variable shuffles, 7 the VNS kick) and ref_reverse_path (refinment.c:95-114) are host code in the drop-in layer:
    path AND prev after the call equal what the compiled reference leaves (tests/golden/golden_make_move.json,
    oracle/make_golden_moves.py); with prev == NULL the layer builds its own and leaves the same path"""
    import json
    import numpy as np
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_make_move.json")))
    ip = C.POINTER(C.c_int)
    host.tabu_make_move.argtypes = [ip, C.POINTER(Solution)] + [C.c_int] * 7
    host.ref_reverse_path.argtypes = [C.c_int] * 4 + [ip, ip]
    host.ref_reverse_path.restype = None
    host.tsp_init()
    inst = Instance.in_dll(host, "tsp_inst")
    seen = set()
    for case in g["cases"]:
        n = case["n"]
        inst.nnodes = n
        succ = np.array(case["succ"], dtype=np.int32)
        path = succ.copy()
        prev = np.empty(n, dtype=np.int32); prev[path] = np.arange(n, dtype=np.int32)
        a = case["args"]
        if case["case"] == 0:
            host.ref_reverse_path(a[0], a[1], a[2], a[3], prev.ctypes.data_as(ip), path.ctypes.data_as(ip))
        else:
            sol = Solution(0.0, path.ctypes.data_as(ip), 0, None)
            assert host.tabu_make_move(prev.ctypes.data_as(ip), C.byref(sol), case["case"], *a) == case["rc"]
            p2 = succ.copy()
            sol2 = Solution(0.0, p2.ctypes.data_as(ip), 0, None)
            assert host.tabu_make_move(None, C.byref(sol2), case["case"], *a) == case["rc"]
            assert p2.tolist() == case["path"], ("prev == NULL", case["case"], n)
        assert path.tolist() == case["path"], (case["case"], n)
        if case["case"] != 7:                       # (case 7 rewrites three successors and leaves prev alone)
            assert prev.tolist() == case["prev"], (case["case"], n)
        seen.add(case["case"])
    assert seen == set(range(8))


def _host_instance(host, name, env=None):
    """tsp_parse_commandline + tsp_read_input (-> tsp_compute_costs on the device) through the library"""
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        host.tsp_gpu_release()
        rc, tenv, inst = parse(host, "-f", os.path.join(DATA, name + ".tsp"), "-q")
        assert rc == 0
        host.tsp_read_input()
        host.utils_startclock(C.byref(inst, Instance.c.offset))
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
    return tenv, inst


@pytest.mark.gpu
def test_host_compute_costs_and_get_cost(host, O):
    """tsp_compute_costs (tsp.c:608-636) fills tsp_inst.costs from the device build, tsp_get_cost (tsp.c:638-640) indexes it;
    in matrix-free mode (no n x n array anywhere) tsp_get_cost recomputes the weight on the host: both against the oracle's
    matrix, EUC_2D and -- TSP_ALLOW_EXT -- ATT"""
    import numpy as np
    host.tsp_get_cost.restype = C.c_double
    host.tsp_get_cost.argtypes = [C.c_int, C.c_int]
    for name, env in (("kroA100", {}), ("pr1002", {}), ("kroA100", {"TSP_MATRIX_FREE": "1"}), ("att48", {"TSP_ALLOW_EXT": "1"}),
                      ("att48", {"TSP_ALLOW_EXT": "1", "TSP_MATRIX_FREE": "1"})):
        tenv, inst = _host_instance(host, name, env)
        xy, ewt = O.read_tsplib(os.path.join(DATA, name + ".tsp"))
        c = O.cost_matrix(xy, {"EUC_2D": O.EUC_2D, "ATT": O.ATT}[ewt])
        n = inst.nnodes
        assert n == len(xy)
        if "TSP_MATRIX_FREE" in env:
            assert not inst.costs
        else:
            got = np.ctypeslib.as_array(C.cast(inst.costs, C.POINTER(C.c_double)), shape=(n, n))
            assert np.array_equal(got, c)
        rng = np.random.default_rng(3)
        for i, j in [(0, 0), (0, 1), (n - 1, 0), (n - 1, n - 1)] + rng.integers(0, n, size=(200, 2)).tolist():
            assert host.tsp_get_cost(int(i), int(j)) == c[i, j], (name, env, i, j)
    host.tsp_free_instance()


@pytest.mark.gpu
def test_host_reference_signatures_on_the_device(host, O, golden):
    """h_greedyutil(start, tsp_solution*, costs) (heuristics.c:216-288), ref_2opt_once(tsp_solution*, costs) looped as a
    caller would (refinment.c:39-93: every delta, the golden cost trace and final tour), ref_2opt(.., update_incumbent),
    tabu_best_move(path, cost, tabu_search*, iter) (metaheuristic.c:188-245: the reference goldens of 12 moves) -- through
    libtsphost.so with the reference's own signatures and struct layouts"""
    import numpy as np
    ip = C.POINTER(C.c_int)
    host.h_greedyutil.argtypes = [C.c_int, C.POINTER(Solution), C.c_void_p]
    host.ref_2opt_once.argtypes = [C.POINTER(Solution), C.c_void_p]
    host.ref_2opt_once.restype = C.c_double
    host.ref_2opt.argtypes = [C.POINTER(Solution), C.c_void_p, C.c_bool]
    host.tabu_best_move.argtypes = [ip, C.POINTER(C.c_double), C.POINTER(TabuSearch), C.c_int]
    host.tsp_init_solution.argtypes = [C.c_int, C.POINTER(Solution)]
    for name in ("kroA100", "pr1002"):
        tenv, inst = _host_instance(host, name)
        n = inst.nnodes
        g = golden["instances"][name]["two_opt"]
        sol = Solution()
        assert host.tsp_init_solution(n, C.byref(sol)) == 0
        assert host.h_greedyutil(0, C.byref(sol), inst.costs) == 0
        path = np.ctypeslib.as_array(sol.path, shape=(n,))
        assert sol.cost == g["nn_cost"] and "%016x" % O.fnv1a(path) == g["nn_fnv"]
        assert host.h_greedyutil(n, C.byref(sol), inst.costs) == 14        # UNAVAILABLE: heuristics.c:224-227
        assert host.h_greedyutil(0, C.byref(sol), inst.costs) == 0
        nn = path.copy()
        run, sweeps = sol.cost, 0
        while True:
            d = host.ref_2opt_once(C.byref(sol), inst.costs)
            sweeps += 1
            if d >= -1e-7:
                break
            run += d
            assert sol.cost == run
            if sweeps <= len(g["trace"]):
                assert run == g["trace"][sweeps - 1]
        assert (sweeps, sol.cost, "%016x" % O.fnv1a(path)) == (g["sweeps"], g["final_cost"], g["final_fnv"])
        # ref_2opt with the incumbent update: the same local optimum from the NN tour, tsp_inst.best_solution takes it
        path[:] = nn
        inst.best_solution.path = C.cast(C.create_string_buffer(4 * n), ip)
        inst.best_solution.cost = 1e300
        assert host.ref_2opt(C.byref(sol), inst.costs, True) == 0
        assert sol.cost == g["final_cost"] and inst.best_solution.cost == g["final_cost"]
        assert "%016x" % O.fnv1a(np.ctypeslib.as_array(inst.best_solution.path, shape=(n,))) == g["final_fnv"]
        inst.best_solution.path = None
    for case in golden["tabu_move"]:
        tenv, inst = _host_instance(host, case["instance"])
        n = inst.nnodes
        xy, _ = O.read_tsplib(os.path.join(DATA, case["instance"] + ".tsp"))
        succ, cost = O.nn_tour(O.cost_matrix(xy), 0)
        tl = np.full(n, -1, dtype=np.int32)
        ts = TabuSearch(case["tenure"], 0, 0, True, tl.ctypes.data_as(ip))
        cc = C.c_double(cost)
        for it, want in enumerate(case["steps"]):
            assert host.tabu_best_move(succ.ctypes.data_as(ip), C.byref(cc), C.byref(ts), it) == 0
            assert (cc.value, "%016x" % O.fnv1a(succ), "%016x" % O.fnv1a(tl)) == (want["cost"], want["fnv"], want["tabu_fnv"])
    host.tsp_free_instance()


@pytest.mark.gpu
@pytest.mark.parametrize("policy,tenure_of", [(0, lambda ts: 30), (1, lambda ts: -(-(ts[1] + ts[2]) // 2))])
def test_host_tabu_search_other_policies(host, O, policy, tenure_of):
    """mh_TabuSearch under the tenure policies no CLI flag selects (metaheuristic.c:126-143: POL_FIXED -> 30, POL_SIZE ->
    ceil((max + min) / 2) -- integer division first, as in the reference): the loop steps through tabu_best_move on the
    device; incumbent after k = 40 iterations against the oracle's moves with the same tenure"""
    import numpy as np
    tenv, inst = _host_instance(host, "kroA100")
    n = inst.nnodes
    xy, _ = O.read_tsplib(os.path.join(DATA, "kroA100.tsp"))
    c = O.cost_matrix(xy)
    tenv.policy = policy
    tenv.k = 40
    inst.best_solution.cost = 1e300
    host.tsp_run_algorithm.restype = C.c_int
    inst.alg = 3
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        assert host.tsp_run_algorithm() == 0
    finally:
        os.chdir(cwd)
    # the oracle: h_greedy_2opt's winner, then 40 tabu moves with the policy's tenure, strict-< incumbent
    best, bc, _, _ = O.multistart_nn_2opt(c)
    succ, cost = best.copy(), bc
    tl = np.full(n, -1, dtype=np.int32)
    ts = (int(0.125 * n + 1), int(0.25 * n), int(0.125 * n))           # tabu_init: tenure, max, min (metaheuristic.c:65-84)
    ten = tenure_of(ts)
    for it in range(40):
        cost, _ = O.tabu_move(c, succ, cost, tl, ten, it)
        if cost < bc:
            bc, best = cost, succ.copy()
    assert inst.best_solution.cost == bc
    assert np.array_equal(np.ctypeslib.as_array(inst.best_solution.path, shape=(n,)), best)
    host.tsp_free_instance()


@pytest.mark.gpu
def test_host_lazy_costs(host, O, golden):
    """tsp_lazy_costs (what the `tsp` executable runs with): tsp_compute_costs builds the matrix on the device only,
    tsp_inst.costs stays NULL -- the heuristic entry points take it as "the instance's matrix" --, tsp_get_cost computes single
    weights on demand, and only an explicit tsp_host_costs() downloads the matrix; the same results as with the eager copy"""
    import numpy as np
    lazy = C.c_bool.in_dll(host, "tsp_lazy_costs")
    host.tsp_get_cost.restype = C.c_double
    host.tsp_get_cost.argtypes = [C.c_int, C.c_int]
    host.h_greedyutil.argtypes = [C.c_int, C.POINTER(Solution), C.c_void_p]
    host.ref_2opt.argtypes = [C.POINTER(Solution), C.c_void_p, C.c_bool]
    host.tsp_init_solution.argtypes = [C.c_int, C.POINTER(Solution)]
    lazy.value = True
    try:
        tenv, inst = _host_instance(host, "pr1002")
        n = inst.nnodes
        assert not inst.costs                                   # nothing on the host yet
        g = golden["instances"]["pr1002"]["two_opt"]
        sol = Solution()
        assert host.tsp_init_solution(n, C.byref(sol)) == 0
        assert host.h_greedyutil(0, C.byref(sol), inst.costs) == 0 and sol.cost == g["nn_cost"]
        assert host.ref_2opt(C.byref(sol), inst.costs, False) == 0 and sol.cost == g["final_cost"]
        assert not inst.costs                                   # the whole heuristic path ran without a host matrix
        xy, _ = O.read_tsplib(os.path.join(DATA, "pr1002.tsp"))
        c = O.cost_matrix(xy)
        # single lookups are computed on demand (the reference's arithmetic, bit-identical to the device matrix): no download
        for i, j in ((3, 977), (977, 3), (0, 1), (500, 500), (1001, 0)):
            assert host.tsp_get_cost(i, j) == c[i, j]
        assert not inst.costs
        host.tsp_host_costs.restype = C.c_void_p
        assert host.tsp_host_costs() and inst.costs                     # the explicit call materialises it
        got = np.ctypeslib.as_array(C.cast(inst.costs, C.POINTER(C.c_double)), shape=(n, n))
        assert np.array_equal(got, c) and host.tsp_get_cost(3, 977) == c[3, 977]
    finally:
        lazy.value = False
        host.tsp_free_instance()


@pytest.mark.gpu
def test_binary_walks_at_resident_sizes_against_the_compiled_reference():
    """whole metaheuristic runs of the compiled reference at the sizes the device-resident loops take
    (tests/golden/golden_walks.json, oracle/make_golden_walks.py: `-alg VNS -k 200` and `-alg TABU_SEARCH -k 200` on pr1002 --
    whole-row LDS kernel --, `-alg VNS -k 12` on fnl4461 -- BASELINE config 3's instance under config 5's algorithm, half-window
    kernel): the `tsp` binary prints the reference's cost"""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_walks.json")))["runs"]
    assert "pr1002_vns_k200" in g
    for key, r in g.items():
        alg = {3: "TABU_SEARCH", 4: "VNS"}[r["alg"]]
        rc, out, err = run_q("-f", os.path.join(DATA, r["instance"] + ".tsp"), "-alg", alg, "-k", str(r["k"]))
        assert rc == 0, (key, err[-500:])
        assert out == "Cost: %.2f" % r["cost"], (key, out)


@pytest.mark.gpu
def test_mh_vns_stops_on_a_code_8_that_is_not_a_refill():
    """tspgpu_vns_search answers RESOURCE_EXHAUSTED (8) both for "the random numbers ran out, call again"
    and for real failures (here: the LDS-resident loop forced, TSPGPU_OPT_PERSIST = 2, on an instance it does not take --
    n = 52 < 64).  mh_VNS must end with the reference's fatal exit (log_fatal + tsp_handlefatal, exit status 1), not
    call again for ever.  Run in a child process: the fatal path exits."""
    code = f"""
import ctypes as C, os, sys
host = C.CDLL({os.path.join(HOST, "libtsphost.so")!r})
argv = [b"tsp", b"-f", {os.path.join(DATA, "berlin52.tsp")!r}.encode(), b"-alg", b"VNS", b"-k", b"50"]
arr = (C.c_char_p * len(argv))(*argv)
assert host.tsp_parse_commandline(len(argv), arr) == 0
host.tsp_read_input()
host.tsp_gpu.restype = C.c_void_p
g = host.tsp_gpu()
gpu = C.CDLL({os.path.join(ROOT, "travellingsalesmanoptimization_amd", "csrc", "libtspgpu.so")!r})
gpu.tspgpu_set_option.argtypes = [C.c_void_p, C.c_int, C.c_long]
sys.path.insert(0, {ROOT!r})
from travellingsalesmanoptimization_amd import _lib
assert gpu.tspgpu_set_option(g, _lib.OPT_PERSIST, 2) == 0
os.makedirs("results", exist_ok=True)
rc = host.tsp_run_algorithm()
print("returned", rc)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 1 and "returned" not in r.stdout, (r.returncode, r.stdout, r.stderr)
    assert "Error in local search" in r.stderr and "does not apply" in r.stderr, r.stderr
