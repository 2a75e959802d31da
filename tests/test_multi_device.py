"""Multi-device multi-start behind the C boundary (include/tspgpu.h tspgpu_multi_*, csrc/tspgpu_multi.cpp; the `tsp`
binary through TSP_GPU_DEVICES): h_greedy_2opt / h_Greedy_iterative sharded over several engine contexts, one
RCCL MIN all-reduce + one broadcast per call (src/algorithms/heuristics.c:74-116, src/tsp.c:669-676).

The test box has ONE MI355X, so the sharding is exercised two ways: (1) a real one-rank RCCL communicator
(G = 1, exchange forced to RCCL: ncclCommInitAll, ncclAllReduce(ncclMin, int64) and ncclBroadcast execute), and
(2) G = 2..3 contexts aliasing device 0 (threads, interleaved start lists, key order, owner selection), which must
exchange on the host because a communicator needs distinct devices."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import data_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TSP = os.path.join(ROOT, "travellingsalesmanoptimization_amd", "host", "tsp")


def test_multi_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from travellingsalesmanoptimization_amd import _lib
    L = _lib.load()
    m = C.c_void_p()
    dev = np.array([0, 1], dtype=np.int32)
    assert L.tspgpu_multi_create(dev, 2, C.byref(m)) == _lib.UNAVAILABLE and not m
    assert L.tspgpu_multi_create(dev, 0, C.byref(m)) == _lib.INVALID_ARGUMENT


def fx(O, v):
    return f"{O.fnv1a(v):016x}"


@pytest.mark.gpu
def test_one_rank_rccl_communicator_executes_the_exchange(O, golden):
    import travellingsalesmanoptimization_amd as T
    m = T.MultiEngine([0])
    try:
        m.set_option(T.MOPT_EXCHANGE, T.EXCHANGE_RCCL)
        for name in ("kroA100", "pr1002"):
            xy, _ = O.read_tsplib(data_path(name))
            m.set_points(xy); m.build_costs()
            m.prepare()
            g = golden["algs"][name + "_2opt_greedy"]
            res = m.multistart_nn_2opt()
            assert res["rc"] == 0 and (res["cost"], fx(O, res["path"])) == (g["cost"], g["fnv"])
            info = m.info()
            assert info["exchange_last"] == 2 and info["rccl_init_s"] > 0 and info["distinct"] == 1
            gi = golden["algs"][name + "_greedy_iter"]
            path, cost, start, done, rc = m.nn_all()
            assert rc == 0 and done == len(xy) and (cost, start, fx(O, path)) == (gi["cost"], gi["starting_node"], gi["fnv"])
        assert m.info()["exchanges"] == 4
    finally:
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_aliased_contexts_shard_like_the_sequential_loop(O, golden, instances, devices):
    import travellingsalesmanoptimization_amd as T
    m = T.MultiEngine(devices)
    try:
        assert m.info()["distinct"] == 0 and m.info()["exchange_next"] == 1
        with pytest.raises(T.TspGpuError) as ei:
            m.set_option(T.MOPT_EXCHANGE, T.EXCHANGE_RCCL)      # a communicator needs distinct devices
        assert ei.value.code == 9
        for name in ("kroA100", "pr1002"):
            xy, _ = O.read_tsplib(data_path(name))
            m.set_points(xy); m.build_costs()
            g = golden["algs"][name + "_2opt_greedy"]
            res = m.multistart_nn_2opt()
            assert res["rc"] == 0 and (res["cost"], fx(O, res["path"])) == (g["cost"], g["fnv"])
            assert m.info()["exchange_last"] == 1
            gi = golden["algs"][name + "_greedy_iter"]
            path, cost, start, done, rc = m.nn_all()
            assert rc == 0 and done == len(xy) and (cost, start, fx(O, path)) == (gi["cost"], gi["starting_node"], gi["fnv"])
            for i in range(len(devices)):
                assert m.device_info(i)["n"] == len(xy)
        # an explicit start list: the winner is the lowest cost, ties to the EARLIEST LIST POSITION, whichever
        # context ran it; compared with the oracle's sequential loop over the same list
        xy, c = instances("n200_s3")
        m.set_points(xy); m.build_costs()
        starts = np.array([7, 199, 0, 33, 34, 150, 3, 7, 90, 12, 13], dtype=np.int32)
        want = O.multistart_nn_2opt(c, starts)
        res = m.multistart_nn_2opt(starts)
        assert (res["cost"], res["start"], res["sweeps"]) == (want[1], want[2], want[3]) and np.array_equal(res["path"], want[0])
        one = m.multistart_nn_2opt(starts[:1])                  # fewer starts than contexts: the others sit idle
        w1 = O.multistart_nn_2opt(c, starts[:1])
        assert (one["cost"], one["start"]) == (w1[1], w1[2]) and np.array_equal(one["path"], w1[0])
        late = m.multistart_nn_2opt(time_left_s=0.0)
        assert late["rc"] == 4 and O.valid_tour(late["path"]) and O.tour_cost(c, late["path"]) == late["cost"]
    finally:
        m.close()


@pytest.mark.gpu
def test_multi_rejects_a_device_that_is_not_there():
    import travellingsalesmanoptimization_amd as T
    with pytest.raises(T.TspGpuError) as ei:
        T.MultiEngine([0, 63])
    assert ei.value.code == 3
    e = dict(os.environ, TSP_GPU_DEVICES="0, 63")
    r = subprocess.run([TSP, "-f", data_path("berlin52"), "-alg", "2OPT_GREEDY"], capture_output=True, text=True, timeout=120, env=e, cwd=ROOT)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


def run_q(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([TSP, *args, "-q"], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    return r.returncode, r.stdout.strip(), r.stderr


def stats(err):
    return [json.loads(l.split("tspgpu-stats:", 1)[1]) for l in err.splitlines() if l.startswith("tspgpu-stats:")]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kroA100", "pr1002"])
def test_binary_multi_device_paths(golden, name):
    """the drop-in binary: `tsp -alg 2OPT_GREEDY` / GREEDY_ITER print the golden `Cost:` line whichever way the
    multi-start is sharded (scripts/compare_algs.py:72 scrapes it)"""
    f = data_path(name)
    g2, gi = golden["algs"][name + "_2opt_greedy"]["cost"], golden["algs"][name + "_greedy_iter"]["cost"]
    for env, kind, ndev in [({"TSP_GPU_DEVICES": "0", "TSP_GPU_EXCHANGE": "rccl"}, "rccl", 1),
                            ({"TSP_GPU_DEVICES": "0,0"}, "host", 2), ({"TSP_GPU_DEVICES": "0"}, "none", 1)]:
        env = dict(env, TSP_GPU_STATS="1")
        rc, out, err = run_q("-f", f, "-alg", "2OPT_GREEDY", env=env)
        assert rc == 0 and out == "Cost: %.2f" % g2, (out, err)     # stdout is the contract: nothing else on it (RCCL's banner goes to stderr)
        st = stats(err)
        assert len(st) == 1 and st[0]["call"] == "h_greedy_2opt" and st[0]["exchange"] == kind and st[0]["devices"] == ndev
        assert st[0]["best_cost"] == g2 and st[0]["sweeps"] > 0 and (st[0]["rccl_init_s"] > 0) == (kind == "rccl")
        rc, out, err = run_q("-f", f, "-alg", "GREEDY_ITER", env=env)
        assert rc == 0 and out == "Cost: %.2f" % gi, err
    e = dict(os.environ, TSP_GPU_DEVICES="0,0", TSP_GPU_EXCHANGE="rccl")      # (without -q: QUIET logs nothing, errors.c:47-62)
    r = subprocess.run([TSP, "-f", f, "-alg", "2OPT_GREEDY"], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 1 and "distinct devices" in r.stderr           # loud, not a silent host exchange
    rc, out, err = run_q("-f", f, "-alg", "TABU_SEARCH", "-k", "20", env={"TSP_GPU_DEVICES": "0,0"})
    assert rc == 0 and out.startswith("Cost: ")            # the other algorithms run on the first device's context
