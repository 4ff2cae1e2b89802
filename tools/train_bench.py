#!/usr/bin/env python
"""Training-step timing of the native decoder (BASELINE config 5 shape, one GPU): compute_loss forward + backward on
B utterances x T frames, per-item t, dropout on.  Prints one JSON line (not the headline metric)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import oracle
    from oracle.inputs import make_inputs
    from stabletts_amd.flow_matching import CFMDecoder
    sd = oracle.make_state_dict(1234)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=args.dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda().train()
    opt = torch.optim.AdamW(dec.parameters(), lr=1e-4)
    B, T = args.batch, args.frames
    inp = {k: v.cuda() for k, v in make_inputs(B, T, seed=0, ragged=True).items() if k != "lengths"}
    x1 = make_inputs(B, T, seed=1)["z"].cuda()
    times = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}

    def step(timed):
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _ = dec.compute_loss(x1, inp["mask"], inp["mu"], inp["c"])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if timed:
            times["fwd"] += t1 - t0; times["bwd"] += t2 - t1; times["opt"] += t3 - t2
        return float(loss.detach())

    for _ in range(2):
        step(False)
    losses = [step(True) for _ in range(args.steps)]
    n = args.steps
    fwd_flops = 2.0 * (12320768 + 3072 * T + 4325376) * B * T           # one evaluation incl. the prenet
    total = (times["fwd"] + times["bwd"]) / n
    print(json.dumps({"workload": f"compute_loss fwd+bwd, B={B} x T={T} ragged, dropout 0.1, {args.dtype} operands",
                      "ms_forward": times["fwd"] / n * 1e3, "ms_backward": times["bwd"] / n * 1e3, "ms_optimizer": times["opt"] / n * 1e3,
                      "frames_per_sec_fwd_bwd": B * T / total, "tflops_fwd_bwd_3x_forward": 3 * fwd_flops / total / 1e12,
                      "losses": losses, "device_GB": torch.cuda.max_memory_allocated() / 1e9,
                      "engine_GB": dec.estimator.engine().device_bytes() / 1e9}))


if __name__ == "__main__":
    main()
