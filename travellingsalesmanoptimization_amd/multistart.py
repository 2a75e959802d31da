"""Multi-start sharded over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What is sharded: the outer loop of h_greedy_2opt (src/algorithms/heuristics.c:82-111) --
one nearest-neighbour seed per start node, each followed by ref_2opt; iterations are
independent except for the incumbent minimum (src/tsp.c:669-676, strict <).

Exchange step (the only collective on the path): ONE all-reduce(MIN) of a packed int64
key (cost << 32 | start) -- integer costs below 2^31, so the minimum key is the lowest
cost, ties to the lowest start id, which is what the sequential strict-< loop keeps --
followed by one broadcast of the winning successor array (4n bytes) from its owner.
Non-integer costs (the mod-costs path) use an all-gather of (cost, start) pairs instead.
Each rank rebuilds the cost matrix from the 16n-byte coordinate array; matrices are
never shipped.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_starts(starts, rank, world):
    """start i -> rank i mod world (interleaved, SURVEY 8e: keeps "all starts < S done"
    roughly true at any deadline cut-off)."""
    return np.ascontiguousarray(np.asarray(starts, dtype=np.int32)[rank::world])


def pack_key(cost, start):
    c = int(cost)
    if c != cost or not (0 <= c < 2 ** 31):
        return None
    return (c << 32) | int(start)


def select_best(cost, start, path, device="cpu", group=None):
    """All ranks call this with their local winner; every rank returns the global
    (cost, start, path).  `start` < 0 means "nothing found on this rank"."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n = len(path)
    if world == 1:
        return cost, start, path
    rank = dist.get_rank(group)
    none = start < 0
    key = None if none else pack_key(cost, start)
    # every rank must take the same branch: agree on "all keys packable"
    flag = torch.tensor([0 if (key is not None or none) else 1], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()) == 0:
        k = torch.tensor([key if key is not None else (2 ** 63 - 1)], dtype=torch.int64, device=device)
        dist.all_reduce(k, op=dist.ReduceOp.MIN, group=group)        # the one MIN all-reduce
        kmin = int(k.item())
        if kmin == 2 ** 63 - 1:
            return float("inf"), -1, path
        best_cost, best_start = float(kmin >> 32), kmin & 0xFFFFFFFF
        mine = (not none) and key == kmin
    else:
        pair = torch.tensor([float("inf") if none else cost, float(start)], dtype=torch.float64, device=device)
        allp = [torch.empty_like(pair) for _ in range(world)]
        dist.all_gather(allp, pair, group=group)
        rows = [(float(p[0]), int(p[1])) for p in allp if int(p[1]) >= 0]
        if not rows:
            return float("inf"), -1, path
        best_cost, best_start = min(rows)
        mine = (not none) and (cost, start) == (best_cost, best_start)
    owner = torch.tensor([rank if mine else world], dtype=torch.int64, device=device)
    dist.all_reduce(owner, op=dist.ReduceOp.MIN, group=group)
    buf = torch.from_numpy(np.ascontiguousarray(path, dtype=np.int32).copy()).to(device)
    dist.broadcast(buf, src=int(owner.item()), group=group)           # winner's tour, 4n bytes
    return best_cost, best_start, buf.cpu().numpy()


def multistart_nn_2opt(solve_local, starts, device="cpu", group=None):
    """solve_local(starts_for_this_rank) -> dict(cost, start, path, sweeps) is the per-GPU
    engine call (Engine.multistart_nn_2opt on the GPU box).  Returns the global result on
    every rank; equals the sequential h_greedy_2opt over `starts` when no deadline is set."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_starts(starts, rank, world)
    if len(mine):
        res = solve_local(mine)
        cost, start, path, sweeps = res["cost"], res["start"], res["path"], res.get("sweeps", 0)
    else:
        n = None
        cost, start, path, sweeps = float("inf"), -1, None, 0
    if path is None:
        # a rank without work still takes part in the collectives; it needs n for the buffer
        nt = torch.tensor([0], dtype=torch.int64, device=device)
        dist.all_reduce(nt, op=dist.ReduceOp.MAX, group=group)
        path = np.zeros(int(nt.item()), dtype=np.int32)
    elif world > 1:
        nt = torch.tensor([len(path)], dtype=torch.int64, device=device)
        dist.all_reduce(nt, op=dist.ReduceOp.MAX, group=group)
    cost, start, path = select_best(cost, start, path, device, group)
    total = sweeps
    if world > 1:
        t = torch.tensor([sweeps], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        total = int(t.item())
    return {"cost": cost, "start": start, "path": path, "sweeps": total}
