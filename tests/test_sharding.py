"""Utterance sharding (multi-GPU inference path): host logic + a world_size-2 gloo run on CPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stabletts_amd import sharding


def test_every_utterance_once_and_balanced():
    rng = np.random.default_rng(0)
    lengths = rng.integers(600, 1001, size=256).tolist()
    for ws in (1, 2, 4, 8):
        per_rank = sharding.assign_batches(lengths, 32, ws)
        seen = sorted(i for bs in per_rank for b in bs for i in b)
        assert seen == list(range(256))
        imb, pad = sharding.imbalance(lengths, per_rank)
        assert imb < 1.25 and pad < 1.05, (ws, imb, pad)
        assert all(len(b) <= 32 for bs in per_rank for b in bs)


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
