#!/usr/bin/env python
"""DDP training of the native CFM decoder in miniature (BASELINE config 5 shape: train.py:49-51,78-81): every rank
wraps the decoder in DistributedDataParallel, calls compute_loss on its share of a fixed batch, and AdamW steps.

  python tools/train_ddp.py --out one.pt                                            (one process, whole batch)
  python -m torch.distributed.run --nproc-per-node N ... tools/train_ddp.py --out n.pt --backend nccl   (one GPU per rank: RCCL)
  BENCH_SHARE_GPU=1 ... --backend gloo                                              (ranks share the visible GPUs: tests)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--items", type=int, default=4)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--trace", action="store_true", help="record the host-side order of native backward parts and DDP bucket launches "
                    "(saved as 'trace' in --out): the reducer must start on the first buckets before the last part is enqueued")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and wrap in DDP even for one rank "
                    "(a 1-GPU box can exercise the nccl = RCCL backend that way)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev if os.environ.get("BENCH_SHARE_GPU") == "1" else local)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import oracle
    from oracle.inputs import make_inputs
    from stabletts_amd.flow_matching import CFMDecoder

    sd = oracle.make_state_dict(1234)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=args.dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.to(dev).eval()            # eval: dropout off, so that the two runs are comparable step by step
    model = dec
    if use_dist:
        model = torch.nn.parallel.DistributedDataParallel(dec, device_ids=[dev.index])
    trace = None
    if args.trace:
        from stabletts_amd import autograd as st_autograd
        trace = st_autograd.TRACE = []
        if use_dist:
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

            def hook(state, bucket):        # the default all-reduce, with the launch recorded
                trace.append(("bucket", bucket.index(), int(bucket.buffer().numel())))
                return default_hooks.allreduce_hook(state, bucket)
            model.register_comm_hook(None, hook)
    opt = torch.optim.AdamW(dec.parameters(), lr=2e-4)
    B, T = args.items, args.frames
    inp = make_inputs(B, T, seed=61)                       # equal lengths: every rank's loss has the same normaliser
    x1 = make_inputs(B, T, seed=62)["z"]
    gen = torch.Generator().manual_seed(5)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    losses = []
    for step in range(args.steps):
        t_rand = torch.rand(B, 1, 1, generator=gen); z = torch.randn(B, 128, T, generator=gen)
        opt.zero_grad()
        if trace is not None:
            trace.append(("step", step))
        if use_dist:       # DDP hooks fire on the module's forward: route compute_loss through it
            loss, _ = _DDPLoss(model)(x1[sl].to(dev), inp["mask"][sl].to(dev), inp["mu"][sl].to(dev), inp["c"][sl].to(dev),
                                      t_rand[sl].to(dev), z[sl].to(dev))
        else:
            loss, _ = dec.compute_loss(x1.to(dev), inp["mask"].to(dev), inp["mu"].to(dev), inp["c"].to(dev),
                                       t_rand=t_rand.to(dev), z=z.to(dev))
        loss.backward()
        opt.step()
        lv = loss.detach().clone()
        if use_dist:
            dist.all_reduce(lv); lv /= world
        losses.append(float(lv))
    if rank == 0:
        keep = ["final_proj.weight", "blocks.3.block.mlp.conv_2.weight", "blocks.0.block.attn.conv_v.weight", "cond_proj.0.bias",
                "blocks.5.block.adaLN_modulation.2.weight", "time_mlp.layer.0.weight"]
        params = {k: v.detach().cpu() for k, v in dec.estimator.named_parameters() if k in keep}
        torch.save(dict(world=world, losses=losses, params=params, trace=trace), args.out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


class _DDPLoss:
    """compute_loss through DistributedDataParallel.forward (train.py calls the DDP-wrapped StableTTS, whose forward
    calls decoder.compute_loss, models/model.py:173): DDP only prepares its gradient hooks inside ITS forward."""
    def __init__(self, ddp):
        self.ddp = ddp
        inner = ddp.module
        if not hasattr(inner, "_orig_forward"):
            inner._orig_forward = inner.forward
            inner.forward = lambda *a, **k: (inner.compute_loss(*a[:4], t_rand=a[4], z=a[5]) if k.get("_loss", True) and len(a) == 6
                                             else inner._orig_forward(*a, **k))

    def __call__(self, *a):
        return self.ddp(*a)


if __name__ == "__main__":
    main()
