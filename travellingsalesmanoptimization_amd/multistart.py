"""Multi-start sharded over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What is sharded: the outer loop of h_greedy_2opt (src/algorithms/heuristics.c:82-111) --
one nearest-neighbour seed per start node, each followed by ref_2opt; iterations are
independent except for the incumbent minimum (src/tsp.c:669-676, strict <).

Exchange step (the only collective on the path): ONE all-reduce(MIN) of a packed int64
key (cost:31 | start:24 | rank:8) -- integer costs below 2^31, so the minimum key is the
lowest cost, ties to the lowest start id, which is what the sequential strict-< loop keeps,
and its low byte names the owner -- followed by one broadcast of the winning successor array
(4n bytes) from that rank.  A non-integer cost anywhere (the mod-costs path) turns the same
all-reduce into the signal for an all-gather of (cost, start, rank) triples instead.
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


NONE_KEY = 2 ** 63 - 1      # "nothing found on this rank"
UNPACKABLE = -1             # a non-integer (or out-of-range) cost somewhere: every rank falls back to the all-gather


def pack_key(cost, start, rank=0):
    """cost:31 | start:24 | rank:8 -- the minimum key is the lowest cost, ties to the lowest
    start id (what the sequential strict-< loop keeps); the rank rides along so that the owner
    of the winning tour is known without another collective."""
    c = int(cost)
    if c != cost or not (0 <= c < 2 ** 31) or not (0 <= int(start) < 2 ** 24) or not (0 <= rank < 256):
        return None
    return (c << 32) | (int(start) << 8) | int(rank)


def select_best(cost, start, path, device="cpu", group=None, force=False):
    """All ranks call this with their local winner; every rank returns the global
    (cost, start, path).  `start` < 0 means "nothing found on this rank".
    Collectives: ONE all-reduce(MIN) of the packed key + one broadcast of the winner's tour.
    A single rank needs no exchange and issues none, unless `force` asks for the collectives
    anyway (the 1-rank RCCL smoke test: the exact calls of the N-rank path on one GPU)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n = len(path)
    if world == 1 and not (force and dist.is_initialized()):
        return cost, start, path
    rank = dist.get_rank(group)
    none = start < 0
    key = NONE_KEY if none else pack_key(cost, start, rank)
    k = torch.tensor([UNPACKABLE if key is None else key], dtype=torch.int64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.MIN, group=group)            # the one MIN all-reduce
    kmin = int(k.item())
    if kmin == NONE_KEY:
        return float("inf"), -1, path
    if kmin != UNPACKABLE:
        best_cost, best_start, owner = float(kmin >> 32), (kmin >> 8) & 0xFFFFFF, kmin & 0xFF
    else:
        pair = torch.tensor([float("inf") if none else cost, float(start), float(rank)], dtype=torch.float64, device=device)
        allp = [torch.empty_like(pair) for _ in range(world)]
        dist.all_gather(allp, pair, group=group)
        rows = [(float(p[0]), int(p[1]), int(p[2])) for p in allp if int(p[1]) >= 0]
        if not rows:
            return float("inf"), -1, path
        best_cost, best_start, owner = min(rows)
    buf = torch.from_numpy(np.ascontiguousarray(path, dtype=np.int32).copy()).to(device)
    dist.broadcast(buf, src=owner, group=group)                       # winner's tour, 4n bytes
    return best_cost, best_start, buf.cpu().numpy()


def multistart_nn_2opt(solve_local, starts, device="cpu", group=None, n=None, total_sweeps=True):
    """solve_local(starts_for_this_rank) -> dict(cost, start, path, sweeps) is the per-GPU
    engine call (Engine.multistart_nn_2opt on the GPU box).  Returns the global result on
    every rank; equals the sequential h_greedy_2opt over `starts` when no deadline is set.
    `n` (instance size) lets a rank without work size its buffer without a collective;
    `total_sweeps` adds one SUM all-reduce of the statistics."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_starts(starts, rank, world)
    if len(mine):
        res = solve_local(mine)
        cost, start, path, sweeps = res["cost"], res["start"], res["path"], res.get("sweeps", 0)
    else:
        cost, start, path, sweeps = float("inf"), -1, None, 0
    if world > 1 and n is None:
        # some rank may be without work (fewer starts than ranks): agree on the buffer size
        nt = torch.tensor([0 if path is None else len(path)], dtype=torch.int64, device=device)
        dist.all_reduce(nt, op=dist.ReduceOp.MAX, group=group)
        n = int(nt.item())
    if path is None:
        path = np.zeros(int(n or 0), dtype=np.int32)
    cost, start, path = select_best(cost, start, path, device, group)
    total = sweeps
    if world > 1 and total_sweeps:
        t = torch.tensor([sweeps], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        total = int(t.item())
    return {"cost": cost, "start": start, "path": path, "sweeps": total}


# ---------------------------------------------------------------------------
# Intra-sweep sharding (optional, config 5): ONE local search whose every sweep is split
# over the ranks -- worthwhile when a sweep is much longer than a collective (pla85900: 6.5 ms on
# one GPU).  Every rank keeps an identical replica of the tour; per sweep each evaluates its share
# of the runs (Engine.tour_sweep_part), ONE all-reduce(MIN) of the packed (delta, a, b) key picks
# the reference's move, every rank applies it (Engine.tour_apply_move).
# ---------------------------------------------------------------------------
def pack_move(delta, a, b):
    """(delta + 2^28):29 | a:17 | b:17 with a < b -- the reference's (delta, a, b) order as one int64"""
    d = int(delta)
    if d != delta or not (-(2 ** 28) < d <= 0) or not (0 <= a <= b < 2 ** 17):
        return None
    return ((d + 2 ** 28) << 34) | (int(a) << 17) | int(b)


def sharded_two_opt(eng, slot, device="cpu", group=None, max_sweeps=-1):
    """ref_2opt (refinment.c:3-37) on `slot`, every sweep sharded over the ranks of `group`.
    Returns the number of sweeps (the final, non-improving one included, as the reference counts)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sweeps = 0
    while max_sweeps < 0 or sweeps < max_sweeps:
        d, a, b = eng.tour_sweep_part(slot, rank, world)
        if world > 1:
            key = pack_move(d, a, b)
            k = torch.tensor([UNPACKABLE if key is None else key], dtype=torch.int64, device=device)
            dist.all_reduce(k, op=dist.ReduceOp.MIN, group=group)        # the one collective of a sweep
            kmin = int(k.item())
            if kmin != UNPACKABLE:
                d, a, b = float((kmin >> 34) - 2 ** 28), (kmin >> 17) & 0x1FFFF, kmin & 0x1FFFF
            else:                                                       # non-integer deltas somewhere
                tri = torch.tensor([d, float(a), float(b)], dtype=torch.float64, device=device)
                allt = [torch.empty_like(tri) for _ in range(world)]
                dist.all_gather(allt, tri, group=group)
                d, a, b = min((float(t[0]), int(t[1]), int(t[2])) for t in allt)
        eng.tour_apply_move(slot, a, b, d)
        sweeps += 1
        if d >= 0:
            break
    return sweeps
