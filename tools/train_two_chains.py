"""Experiment (round 6): would TWO independent half-batch training chains on two streams beat one full-batch chain?  Two decoders with the
same weights, B/2 utterances each, forward + backward enqueued on their own torch streams (no optimizer, no gradient merge: an upper bound
of what a two-part training step inside the engine could return), against one decoder at B.  Prints ms per forward + backward."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import oracle
    from oracle.inputs import make_inputs
    from stabletts_amd.flow_matching import CFMDecoder
    dev = torch.device("cuda", 0)
    sd = oracle.make_state_dict(1234)
    B, T, steps = 64, 1000, 8

    def mk():
        d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
        d.estimator.load_state_dict(sd)
        return d.to(dev).train(True)

    raw = make_inputs(B, T, seed=0, ragged=True)
    order = torch.argsort(raw["lengths"], descending=True)          # deal the utterances alternately: equal work per half
    halves = [order[0::2], order[1::2]]
    x1 = make_inputs(B, T, seed=1)["z"]
    full = {k: v.to(dev) for k, v in raw.items() if k != "lengths"}
    part = [{k: v[h].to(dev) for k, v in raw.items() if k != "lengths"} for h in halves]
    x1p = [x1[h].to(dev) for h in halves]
    x1 = x1.to(dev)
    res = {}
    with torch.enable_grad():
        one = mk()

        def step_one():
            one.zero_grad(set_to_none=True)
            loss, _ = one.compute_loss(x1, full["mask"], full["mu"], full["c"])
            loss.backward()
        for _ in range(3):
            step_one()
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(steps):
            step_one()
        torch.cuda.synchronize(dev)
        res["ms_one_chain_B64"] = (time.perf_counter() - t0) / steps * 1e3
        del one
        two = [mk(), mk()]
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

        def step_two():
            losses = []
            for i in range(2):
                two[i].zero_grad(set_to_none=True)
                with torch.cuda.stream(streams[i]):
                    loss, _ = two[i].compute_loss(x1p[i], part[i]["mask"], part[i]["mu"], part[i]["c"])
                    losses.append(loss)
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    losses[i].backward()
        for _ in range(3):
            step_two()
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(steps):
            step_two()
        torch.cuda.synchronize(dev)
        res["ms_two_chains_B32_each"] = (time.perf_counter() - t0) / steps * 1e3
        # one half alone (what a chain costs without the other)
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(steps):
            two[0].zero_grad(set_to_none=True)
            loss, _ = two[0].compute_loss(x1p[0], part[0]["mask"], part[0]["mu"], part[0]["c"])
            loss.backward()
        torch.cuda.synchronize(dev)
        res["ms_one_chain_B32"] = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps(res))


if __name__ == "__main__":
    main()
