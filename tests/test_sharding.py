"""Utterance sharding (multi-GPU inference path): host logic + a world_size-2 gloo run on CPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stabletts_amd import sharding


def _check(lengths, per_rank):
    seen = sorted(i for bs in per_rank for b in bs for i in b)
    assert seen == list(range(len(lengths)))
    for bs in per_rank:
        for b in bs:                                   # every batch is a contiguous run of the length-sorted list
            ls = [lengths[i] for i in b]
            assert ls == sorted(ls, reverse=True)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
def test_config4_every_utterance_once_and_balanced(seed):
    """BASELINE config 4: 256 utterances U{600..1000}, mean batch 32.  Equal-cost, variable-count buckets: the heaviest rank is
    within 3 % of the mean (fixed-count buckets of 32: 1.24 at 8 ranks -- asserted too, so the reason for the policy stays visible)."""
    lengths = np.random.default_rng(seed).integers(600, 1001, size=256).tolist()
    for ws in (1, 2, 4, 8):
        per_rank = sharding.assign_batches(lengths, 32, ws)
        _check(lengths, per_rank)
        imb, pad = sharding.imbalance(lengths, per_rank)
        assert imb <= 1.03 and pad <= 1.05, (ws, imb, pad)
        assert all(len(bs) == max(1, 8 // ws) for bs in per_rank)           # same number of buckets on every rank
        assert sharding.scaling_ceiling(lengths, per_rank) >= ws / 1.03
        assert all(len(b) * max(lengths[i] for i in b) <= sharding.MAX_PADDED_FRAMES for bs in per_rank for b in bs)
    fixed = sharding.assign_batches(lengths, 32, 8, equal_cost=False)
    _check(lengths, fixed)
    assert all(len(b) == 32 for bs in fixed for b in bs)
    assert sharding.imbalance(lengths, fixed)[0] > 1.2


def test_skewed_lengths():
    """80 % short / 20 % long utterances.  With 1024 utterances the buckets are fine-grained enough for 3 % at every world size;
    with 256 the 51 long utterances fill four buckets of 12-14 at 8 ranks (one utterance = 7.7 % of a bucket) and the long / short
    halves cannot trade work without mixing lengths in one bucket: 1.09 measured, gate 1.10 (more, smaller buckets would balance it
    and cost more than that in per-solve fixed time: ~4 ms of a 20-ms solve)."""
    rng = np.random.default_rng(9)
    for n, gate8 in ((1024, 1.03), (256, 1.10)):
        lengths = np.concatenate([rng.integers(150, 300, size=n * 4 // 5), rng.integers(800, 1001, size=n - n * 4 // 5)]).tolist()
        for ws in (2, 4, 8):
            per_rank = sharding.assign_batches(lengths, 32, ws)
            _check(lengths, per_rank)
            imb, pad = sharding.imbalance(lengths, per_rank)
            assert imb <= (gate8 if ws == 8 else 1.03), (n, ws, imb)
            assert pad <= 1.07, (n, ws, pad)
            assert all(len(b) <= 64 for bs in per_rank for b in bs)          # count bound: 2 x the mean


def test_count_bound_and_row_index_limit():
    """A bucket never holds more than max_count utterances nor more padded frames than the engine's 32-bit row index allows."""
    lengths = [20] * 500 + [1000] * 12
    per_rank = sharding.assign_batches(lengths, 32, 4)
    _check(lengths, per_rank)
    assert all(len(b) <= 64 for bs in per_rank for b in bs)
    batches = sharding.make_batches([40000] * 100, 32, num_batches=2, max_count=64)
    assert all(len(b) * 40000 <= sharding.MAX_PADDED_FRAMES for b in batches) and sum(len(b) for b in batches) == 100


def test_edge_cases():
    assert sharding.assign_batches([], 4, 2) == [[], []]
    per = sharding.assign_batches([5], 4, 3)
    assert sum(len(bs) for bs in per) == 1
    with pytest.raises(ValueError):
        sharding.assign_batches([1, 2], 0, 1)


def _worker(rank, world, port, lengths, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.shard_for_rank(lengths, 8, world, rank)
    ids = torch.full((len(lengths),), -1, dtype=torch.long)
    flat = [i for b in mine for i in b]
    ids[: len(flat)] = torch.tensor(flat, dtype=torch.long)
    gathered = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids)            # test-only collective: the data path itself has none
    frames = torch.tensor([float(sum(lengths[i] for i in flat))])
    dist.all_reduce(frames)
    if rank == 0:
        allids = sorted(int(v) for g in gathered for v in g if v >= 0)
        q.put((allids, float(frames)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    lengths = np.random.default_rng(1).integers(100, 400, size=40).tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    allids, frames = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert allids == list(range(40))
    assert frames == float(sum(lengths))
