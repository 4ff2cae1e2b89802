"""Utterance sharding for multi-GPU inference (one process per GPU, no data-path collective).

Each utterance's ODE solve touches only its own (mu, mask, c, z) (attention is within-utterance,
models/diffusion_transformer.py:107), so the path shards over independent units.  The policy
mirrors the reference's training-side ``DistributedBucketSampler`` (datas/sampler.py:67-114):
sort by length, cut into batches of similar length, deal batches to ranks -- here greedily by
estimated cost so the ranks finish together (scaling limit = padding + imbalance).
"""
from typing import List, Sequence, Tuple


def utterance_cost(length: int) -> float:
    """Relative cost of one utterance per estimator evaluation: MACs per frame are
    12,320,768 + 3,072*T, i.e. linear + attention-quadratic in T."""
    return float(length) * (12320768.0 + 3072.0 * float(length))


def make_batches(lengths: Sequence[int], batch_size: int) -> List[List[int]]:
    """Length-sorted batches of utterance indices (longest first; each batch pads to its own max)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[i:i + batch_size] for i in range(0, len(order), batch_size)]


def batch_cost(lengths: Sequence[int], batch: Sequence[int]) -> float:
    """A padded batch costs (number of items) x cost(max length in the batch)."""
    return len(batch) * utterance_cost(max(int(lengths[i]) for i in batch))


def assign_batches(lengths: Sequence[int], batch_size: int, world_size: int) -> List[List[List[int]]]:
    """Deterministic LPT assignment of the length-sorted batches to ranks.
    Returns per_rank[r] = list of batches (lists of utterance indices). Every utterance appears
    exactly once; identical on every rank (no communication needed)."""
    if world_size < 1 or batch_size < 1:
        raise ValueError("world_size and batch_size must be >= 1")
    batches = make_batches(lengths, batch_size)
    batches.sort(key=lambda b: (-batch_cost(lengths, b), b[0]))
    load = [0.0] * world_size
    per_rank: List[List[List[int]]] = [[] for _ in range(world_size)]
    for b in batches:
        r = min(range(world_size), key=lambda k: (load[k], k))
        per_rank[r].append(b)
        load[r] += batch_cost(lengths, b)
    return per_rank


def shard_for_rank(lengths: Sequence[int], batch_size: int, world_size: int, rank: int) -> List[List[int]]:
    return assign_batches(lengths, batch_size, world_size)[rank]


def imbalance(lengths: Sequence[int], per_rank: List[List[List[int]]]) -> Tuple[float, float]:
    """(max rank cost / mean rank cost, padded frames / valid frames) of an assignment."""
    costs = [sum(batch_cost(lengths, b) for b in bs) for bs in per_rank]
    mean = sum(costs) / max(len(costs), 1)
    padded = sum(len(b) * max(int(lengths[i]) for i in b) for bs in per_rank for b in bs)
    valid = sum(int(lengths[i]) for bs in per_rank for b in bs for i in b)
    return (max(costs) / mean if mean > 0 else 1.0, padded / max(valid, 1))
