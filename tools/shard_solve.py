#!/usr/bin/env python
"""Utterance-sharded inference in miniature (test helper, also a usage example of stabletts_amd.sharding):
every rank solves the batches sharding.assign_batches deals to it; rank 0 gathers the mels by utterance id.

  python tools/shard_solve.py --out one.pt                                   (one process)
  BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 ... tools/shard_solve.py --out two.pt
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--batch", type=int, default=3)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev if os.environ.get("BENCH_SHARE_GPU") == "1" else local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import oracle
    from oracle.inputs import make_inputs
    from stabletts_amd import sharding
    from stabletts_amd.flow_matching import CFMDecoder

    sd = oracle.make_state_dict(1234)
    fs, fc = oracle.make_cfg_params(4321)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="bf16")
    dec.estimator.load_state_dict(sd)
    dec = dec.to(dev)
    kw = dict(fake_speaker=fs.to(dev), fake_content=fc.to(dev), cfg_strength=3.0)

    lengths = np.random.default_rng(5).integers(60, 260, size=args.n).tolist()
    per_rank = sharding.assign_batches(lengths, args.batch, world)
    mine = {}
    with torch.inference_mode():
        for batch in per_rank[rank]:
            T = max(lengths[i] for i in batch)
            # every utterance has its own seeded content, independent of the batch it lands in
            items = [make_inputs(1, T, seed=1000 + i, lengths=[lengths[i]]) for i in batch]
            cat = {k: torch.cat([it[k] for it in items]).to(dev) for k in ("mu", "mask", "c", "z")}
            out = dec(cat["mu"], cat["mask"], 4, 1.0, cat["c"], "euler", kw, z=cat["z"]).cpu()
            for j, i in enumerate(batch):
                mine[i] = out[j, :, :lengths[i]].clone()
    if world > 1:
        gathered = [None] * world
        dist.gather_object(mine, gathered if rank == 0 else None, dst=0)
        if rank == 0:
            mine = {k: v for g in gathered for k, v in g.items()}
    if rank == 0:
        torch.save(dict(world=world, mel=mine, imbalance=sharding.imbalance(lengths, per_rank)), args.out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
