"""Utterance sharding for multi-GPU inference (one process per GPU, no data-path collective).

Each utterance's ODE solve touches only its own (mu, mask, c, z) (attention is within-utterance,
models/diffusion_transformer.py:107), so the path shards over independent units.  The policy
mirrors the reference's training-side ``DistributedBucketSampler`` (datas/sampler.py:67-114):
sort by length, cut into batches of similar length, deal batches to ranks.  The reference cuts
buckets of a FIXED count; a padded batch costs count x cost(longest), so fixed counts give the
rank holding the longest bucket 1.6x the work of the one holding the shortest (256 utterances
U{600..1000} in 8 batches of 32: max/mean 1.24).  Here the sorted list is cut into batches of
EQUAL COST and variable count -- ``batch_size`` is the MEAN count -- then dealt greedily
(longest-processing-time first) so the ranks finish together.
"""
from typing import List, Sequence, Tuple

# The engine indexes rows with 32 bits: 2 (CFG) x B x T x filter(1024) < 2^31 (engine.cpp: "B*T too large").
MAX_PADDED_FRAMES = (1 << 20) - 1


def utterance_cost(length: int) -> float:
    """Relative cost of one utterance per estimator evaluation: MACs per frame are
    12,320,768 + 3,072*T, i.e. linear + attention-quadratic in T."""
    return float(length) * (12320768.0 + 3072.0 * float(length))


def batch_cost(lengths: Sequence[int], batch: Sequence[int]) -> float:
    """A padded batch costs (number of items) x cost(max length in the batch)."""
    return len(batch) * utterance_cost(max(int(lengths[i]) for i in batch))


def _sorted_order(lengths: Sequence[int]) -> List[int]:
    return sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))


def _max_count(first_len: int, max_count: int) -> int:
    """Largest item count a batch whose longest utterance has `first_len` frames may hold."""
    return max(1, min(max_count, MAX_PADDED_FRAMES // max(int(first_len), 1)))


def _greedy_cuts(sl: List[int], cap: float, max_count: int) -> List[int]:
    """Cut points of the sorted-descending length list `sl` when every batch takes as many
    utterances as fit under `cap` (cost = count x cost(first)).  Returns the start indices."""
    starts, i, n = [], 0, len(sl)
    while i < n:
        starts.append(i)
        c1 = utterance_cost(sl[i])
        take = int(cap / c1) if c1 > 0 else n
        i += max(1, min(take, _max_count(sl[i], max_count), n - i))
    return starts


def make_batches(lengths: Sequence[int], batch_size: int, num_batches: int = 0, max_count: int = 0) -> List[List[int]]:
    """Length-sorted batches of utterance indices (longest first; each batch pads to its own max).

    num_batches == 0: fixed-count batches of `batch_size` (the reference sampler's rule).
    num_batches  > 0: at most `num_batches` contiguous batches of EQUAL COST: the smallest cap such
    that greedy filling needs <= num_batches batches (bisection), then boundary moves that raise the
    cheapest batch without exceeding the cap.  `max_count` (default 2 x batch_size) bounds a batch's
    item count (workspace is sized by count x longest)."""
    order = _sorted_order(lengths)
    n = len(order)
    if num_batches <= 0 or n == 0:
        return [order[i:i + batch_size] for i in range(0, n, batch_size)]
    max_count = max_count or 2 * batch_size
    sl = [int(lengths[i]) for i in order]
    num_batches = min(num_batches, n)
    # bisection on the cap: lo is infeasible (or 0), hi feasible
    lo, hi = 0.0, sum(utterance_cost(sl[0]) for _ in range(n))
    need = -(-n // max_count)
    if need > num_batches:              # count bound forces more batches than asked: equal-cost cut of `need` batches
        num_batches = need
    for _ in range(64):
        mid = 0.5 * (lo + hi)
        if len(_greedy_cuts(sl, mid, max_count)) <= num_batches:
            hi = mid
        else:
            lo = mid
        if hi - lo <= 1e-9 * hi:
            break
    starts = _greedy_cuts(sl, hi, max_count)
    # the greedy cut leaves the LAST batch light: walk boundaries backwards, handing items to the later (lighter) batch
    # while that lowers the pair's maximum -- never above the cap, so the makespan bound is kept
    ends = starts[1:] + [n]

    def cost(a, b):
        return (b - a) * utterance_cost(sl[a]) if b > a else 0.0

    for _ in range(4 * len(starts)):
        moved = False
        for k in range(len(starts) - 1, 0, -1):
            a0, a1, b1 = starts[k - 1], starts[k], ends[k]
            while a1 - a0 > 1:
                cur = max(cost(a0, a1), cost(a1, b1))
                new = max(cost(a0, a1 - 1), cost(a1 - 1, b1))
                if new < cur and (b1 - (a1 - 1)) <= _max_count(sl[a1 - 1], max_count) and cost(a1 - 1, b1) <= hi * (1 + 1e-9):
                    a1 -= 1
                    moved = True
                else:
                    break
            starts[k] = a1
            ends[k - 1] = a1
        if not moved:
            break
    return [order[a:b] for a, b in zip(starts, ends) if b > a]


def assign_batches(lengths: Sequence[int], batch_size: int, world_size: int, equal_cost: bool = True) -> List[List[List[int]]]:
    """Deterministic assignment of length-sorted batches to ranks.
    Returns per_rank[r] = list of batches (lists of utterance indices). Every utterance appears
    exactly once; identical on every rank (no communication needed).

    equal_cost (default): ceil(n / batch_size) batches rounded UP to a multiple of world_size, cut at equal cost
    (variable count, mean `batch_size`), so every rank gets the same number of near-equal batches.
    equal_cost=False: the reference sampler's fixed-count buckets (kept for A/B and for callers that need exact counts)."""
    if world_size < 1 or batch_size < 1:
        raise ValueError("world_size and batch_size must be >= 1")
    n = len(lengths)
    if equal_cost and n:
        nb = -(-n // batch_size)
        nb = -(-nb // world_size) * world_size
        batches = make_batches(lengths, batch_size, num_batches=nb)
    else:
        batches = make_batches(lengths, batch_size)
    batches.sort(key=lambda b: (-batch_cost(lengths, b), b[0]))
    load = [0.0] * world_size
    per_rank: List[List[List[int]]] = [[] for _ in range(world_size)]
    for b in batches:
        r = min(range(world_size), key=lambda k: (load[k], k))
        per_rank[r].append(b)
        load[r] += batch_cost(lengths, b)
    return per_rank


def shard_for_rank(lengths: Sequence[int], batch_size: int, world_size: int, rank: int, equal_cost: bool = True) -> List[List[int]]:
    return assign_batches(lengths, batch_size, world_size, equal_cost)[rank]


def imbalance(lengths: Sequence[int], per_rank: List[List[List[int]]]) -> Tuple[float, float]:
    """(max rank cost / mean rank cost, padded frames / valid frames) of an assignment."""
    costs = [sum(batch_cost(lengths, b) for b in bs) for bs in per_rank]
    mean = sum(costs) / max(len(costs), 1)
    padded = sum(len(b) * max(int(lengths[i]) for i in b) for bs in per_rank for b in bs)
    valid = sum(int(lengths[i]) for bs in per_rank for b in bs for i in b)
    return (max(costs) / mean if mean > 0 else 1.0, padded / max(valid, 1))


def scaling_ceiling(lengths: Sequence[int], per_rank: List[List[List[int]]]) -> float:
    """Cost-model speed-up of this assignment over ONE rank solving the same batches back to back:
    sum of all batch costs / the heaviest rank's cost (= world_size / imbalance)."""
    costs = [sum(batch_cost(lengths, b) for b in bs) for bs in per_rank]
    return sum(costs) / max(costs) if costs and max(costs) > 0 else 1.0
