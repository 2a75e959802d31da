"""The N>1 path on CPU: world_size-2 `gloo` run of the sharded multi-start driver
(travellingsalesmanoptimization_amd/multistart.py).  The per-rank solver is injected; here
it is the oracle (test infrastructure), on the GPU box it is Engine.multistart_nn_2opt."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, float_costs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from travellingsalesmanoptimization_amd import multistart
    xy, _ = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", name + ".tsp"))
    c = O.cost_matrix(xy)
    if float_costs:
        c = np.ascontiguousarray(c * 0.731)
    n = len(xy)

    def solve_local(starts):
        path, cost, start, sweeps = O.multistart_nn_2opt(c, starts)
        return {"path": path, "cost": cost, "start": start, "sweeps": sweeps}

    starts = np.arange(n, dtype=np.int32) if not float_costs else np.arange(0, n, 3, dtype=np.int32)
    res = multistart.multistart_nn_2opt(solve_local, starts)
    q.put((rank, res["cost"], res["start"], res["sweeps"], O.fnv1a(res["path"])))
    dist.destroy_process_group()


def _run(world, name, float_costs=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, float_costs, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    return sorted(out)


@pytest.mark.parametrize("name", ["berlin52", "kroA100"])
def test_two_ranks_equal_sequential_golden(golden, name):
    """sharding start i -> rank i mod 2 and one MIN all-reduce gives the sequential
    h_greedy_2opt result (golden, from the reference) on every rank"""
    g = golden["algs"][name + "_2opt_greedy"]
    out = _run(2, name)
    assert len(out) == 2
    for rank, cost, start, sweeps, fnv in out:
        assert cost == g["cost"] and f"{fnv:016x}" == g["fnv"]
    assert out[0][1:] == out[1][1:]


def test_three_ranks_float_costs_allgather_path(O):
    """non-integer costs cannot be packed into the int64 key: all-gather fallback"""
    out = _run(3, "berlin52", float_costs=True)
    xy, _ = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", "berlin52.tsp"))
    c = np.ascontiguousarray(O.cost_matrix(xy) * 0.731)
    want = O.multistart_nn_2opt(c, np.arange(0, 52, 3, dtype=np.int32))
    for rank, cost, start, sweeps, fnv in out:
        assert (cost, start, sweeps, fnv) == (want[1], want[2], want[3], O.fnv1a(want[0]))


def test_pack_move_order():
    """the per-sweep key of the sharded sweep: the reference's (delta, a, b) order, int64-safe"""
    sys.path.insert(0, ROOT)
    from travellingsalesmanoptimization_amd import multistart as M
    ks = [M.pack_move(-5, 3, 9), M.pack_move(-5, 3, 10), M.pack_move(-5, 4, 5), M.pack_move(-4, 0, 1), M.pack_move(0, 0, 0)]
    assert ks == sorted(ks) and len(set(ks)) == len(ks) and max(ks) < 2 ** 63
    assert M.pack_move(-(2 ** 28) + 1, 131070, 131071) > 0
    assert M.pack_move(-0.5, 1, 2) is None and M.pack_move(-(2 ** 28), 1, 2) is None and M.pack_move(-1, 1, 2 ** 17) is None


def test_pack_key_and_sharding():
    sys.path.insert(0, ROOT)
    from travellingsalesmanoptimization_amd import multistart as M
    assert M.pack_key(266290.0, 1001, 3) == (266290 << 32) | (1001 << 8) | 3
    assert M.pack_key(1121.03, 5) is None
    # the rank in the low byte never decides: a start lives on exactly one rank
    assert M.pack_key(7657.0, 12, 7) < M.pack_key(7657.0, 13, 0) < M.pack_key(7658.0, 0, 0)
    # lowest cost wins; ties go to the lowest start id (the strict < of tsp.c:671 under
    # ascending iteration order)
    keys = [M.pack_key(7657.0, 51), M.pack_key(7657.0, 12), M.pack_key(7700.0, 0)]
    assert min(keys) == M.pack_key(7657.0, 12)
    s = np.arange(10)
    assert M.shard_starts(s, 1, 4).tolist() == [1, 5, 9]
    assert sum(len(M.shard_starts(s, r, 4)) for r in range(4)) == 10
    # single process: no collective is issued
    c, st, p = M.select_best(5.0, 3, np.arange(4, dtype=np.int32))
    assert (c, st) == (5.0, 3)


def test_c_driver_selection_orders_agree():
    """the RCCL path of the C multi-device driver (csrc/tspgpu_multi.cpp) has only ever run on one rank.  Its
    selection is pure arithmetic on the keys the devices reduce with ncclMin -- one packed int64, or, for costs that do not
    pack (fractional, >= 2^31, list position >= 2^24), the cost's IEEE bit pattern and then position | rank --:
    tspgpu_multi_select computes those keys on the host (no device) and must pick the same winner as the host exchange's
    (cost, position, rank) order, ties and empty devices included"""
    sys.path.insert(0, ROOT)
    import travellingsalesmanoptimization_amd as T
    L = T._lib.load()
    rng = np.random.default_rng(5)

    def both(cost, pos):
        cost = np.ascontiguousarray(cost, dtype=np.float64); pos = np.ascontiguousarray(pos, dtype=np.int64)
        a = L.tspgpu_multi_select(cost, pos, len(cost), 0)
        b = L.tspgpu_multi_select(cost, pos, len(cost), 1)
        assert a == b, (cost, pos, a, b)
        return a

    assert both([7657.0, 7657.0, 7700.0], [51, 12, 0]) == 1                    # tie on the cost: the earliest list position
    assert both([5.0, 4.0, 4.0, 9.0], [3, -1, 8, 1]) == 2                      # a device that found nothing never wins
    assert both([1.0, 2.0], [-1, -1]) == -1
    assert both([1121.03, 1121.02, 1121.03], [0, 9, 1]) == 1                   # fractional costs: the bit-pattern branch
    assert both([2.0 ** 31, 2.0 ** 31 + 1, 2.0 ** 31], [7, 0, 3]) == 2          # costs >= 2^31 do not pack; tie -> position 3 < 7
    assert both([10.0, 10.0], [2 ** 24 + 5, 2 ** 24 + 1]) == 1                 # positions >= 2^24 do not pack
    assert both([0.0, 0.0, 0.0], [2, 1, 1]) in (1, 2)                          # (equal cost and position cannot happen: a start lives on one device)
    for _ in range(300):
        G = int(rng.integers(1, 9))
        kind = rng.integers(0, 4)
        base = [1000.0, 1000.5, 2.0 ** 31 + 10, 1e15][kind]
        cost = base + rng.integers(0, 4, size=G) * (0.25 if kind == 1 else 1.0)
        pos = rng.permutation(64)[:G].astype(np.int64) + (2 ** 24 if rng.integers(0, 5) == 0 else 0)
        pos[rng.random(G) < 0.2] = -1
        both(cost, pos)
    assert L.tspgpu_multi_select(np.array([-1.0]), np.array([0], dtype=np.int64), 1, 1) in (0, -2)   # (a negative cost: refused or packed away)


# ---------------------------------------------------------------------------
# intra-sweep sharding: two ranks share ONE GPU (gloo for the 8-byte key), each evaluates half of
# the runs of every sweep; the trajectory must be the single-GPU one
# ---------------------------------------------------------------------------
def _shard_worker(rank, world, port, name, matrix_free, cap, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import travellingsalesmanoptimization_amd as T
    from travellingsalesmanoptimization_amd import multistart
    kind = O.EUC_2D
    if name.startswith("n"):
        xy = O.random_points(int(name[1:]), 123)
    else:
        xy, ewt = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", name + ".tsp"))
        if name == "pla85900":
            kind = O.CEIL_2D
    eng = T.Engine(0)
    eng.set_option(T.OPT_MATRIX_FREE, 1 if matrix_free else 2)
    eng.set_points(xy, kind); eng.build_costs()
    eng.tour_nn(0, 0)
    sweeps = multistart.sharded_two_opt(eng, 0, max_sweeps=cap)
    succ, cost, _ = eng.tour_store(0)
    q.put((rank, sweeps, cost, O.fnv1a(succ), eng.info()["kernel"]))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name,matrix_free,cap", [("pr1002", False, -1), ("n1000", True, -1), ("d18512", False, 5), ("d18512", True, 5),
                                                  ("pla85900", True, 3)])
def test_sharded_sweep_two_ranks_one_gpu(name, matrix_free, cap):
    import json
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    if name == "pla85900":      # config 5: the case the sharded sweep exists for
        m = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_large.json")))["pla85900"]["moves"][cap - 1]
        g = {"sweeps": cap, "final_cost": m["cost"], "final_fnv": m["fnv"]}
    else:
        g = (golden["instances"].get(name) or golden["random"]["n1000_s123"])["two_opt"]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, name, matrix_free, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, sweeps, cost, fnv, kernel in out:
        assert (sweeps, cost, "%016x" % fnv) == (cap if cap > 0 else g["sweeps"], g["final_cost"], g["final_fnv"]), out
        assert (kernel == 4) == matrix_free


# ---------------------------------------------------------------------------
# the sharded multi-start with the PRODUCT as the per-rank solver: two ranks share the one GPU of the
# test box (gloo for the key exchange), each runs Engine.multistart_nn_2opt on its share of the starts
# (start i -> rank i mod 2); the result must be the sequential h_greedy_2opt one (golden, reference)
# ---------------------------------------------------------------------------
def _engine_worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import travellingsalesmanoptimization_amd as T
    from travellingsalesmanoptimization_amd import multistart
    xy, _ = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", name + ".tsp"))
    n = len(xy)
    eng = T.Engine(0)
    eng.set_points(xy); eng.build_costs()
    calls = []

    def solve_local(starts):
        calls.append(len(starts))
        return eng.multistart_nn_2opt(starts)          # the HIP engine, through the C ABI

    res = multistart.multistart_nn_2opt(solve_local, np.arange(n, dtype=np.int32), n=n)
    q.put((rank, res["cost"], res["start"], res["sweeps"], O.fnv1a(res["path"]), bool(O.valid_tour(res["path"])), calls))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kroA100", "pr1002"])
def test_sharded_multistart_engine_two_ranks_one_gpu(name):
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["algs"][name + "_2opt_greedy"]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n = {"kroA100": 100, "pr1002": 1002}[name]
    for rank, cost, start, sweeps, fnv, valid, calls in out:
        assert (cost, "%016x" % fnv) == (g["cost"], g["fnv"]) and valid, out
        assert calls == [len(range(rank, n, world))]       # each rank solved exactly its own share, once
    assert out[0][1:5] == out[1][1:5]


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)       # "nccl" IS RCCL on ROCm
    import oracle as O
    import travellingsalesmanoptimization_amd as T
    from travellingsalesmanoptimization_amd import multistart
    xy, _ = O.read_tsplib(os.path.join(ROOT, "tests", "golden", "data", "kroA100.tsp"))
    eng = T.Engine(0)
    eng.set_points(xy); eng.build_costs()
    res = eng.multistart_nn_2opt()
    dev = torch.device("cuda", 0)
    # integer costs: all_reduce(MIN, int64) + broadcast; non-integer: all_reduce + all_gather + broadcast
    c1, s1, p1 = multistart.select_best(res["cost"], res["start"], res["path"], device=dev, force=True)
    c2, s2, p2 = multistart.select_best(res["cost"] + 0.25, res["start"], res["path"], device=dev, force=True)
    torch.cuda.synchronize()
    q.put((res["cost"], res["start"], O.fnv1a(res["path"]), c1, s1, O.fnv1a(p1), c2, s2, O.fnv1a(p2), dist.get_backend()))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_executes_the_exchange_on_one_rank():
    """the collective of the sharded multi-start, executed by RCCL itself: a 1-rank "nccl" process group on the
    MI355X, select_best forced through all_reduce(MIN, int64) + broadcast (and the all-gather variant of the
    float-cost path) next to a live engine context on the same GPU"""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["algs"]["kroA100_2opt_greedy"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    cost, start, fnv, c1, s1, f1, c2, s2, f2, backend = out
    assert backend == "nccl"
    assert (cost, "%016x" % fnv) == (g["cost"], g["fnv"])
    assert (c1, s1, f1) == (cost, start, fnv)
    assert (c2, s2, f2) == (cost + 0.25, start, fnv)
