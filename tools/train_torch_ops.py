#!/usr/bin/env python
"""Which ATen / runtime launches does ONE training step add around the native kernels?  torch.profiler over three steps of bench.py's
train_step shape (B=64 x T=1000 ragged, dropout, AdamW): device kernels / memsets / memcpys grouped by name, with the CPU op that issued
the small ones.  Developer tool (run on the GPU box)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder

dev = torch.device("cuda", 0)
dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
dec.estimator.load_state_dict(oracle.make_state_dict(1234))
dec = dec.to(dev).train(True)
opt = torch.optim.AdamW(dec.parameters(), lr=1e-4)
raw = make_inputs(64, 1000, seed=0, ragged=True)
inp = {k: v.to(dev) for k, v in raw.items() if k != "lengths"}
x1 = make_inputs(64, 1000, seed=1)["z"].to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = dec.compute_loss(x1, inp["mask"], inp["mu"], inp["c"])
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 or "fill" in e.key.lower() or "copy" in e.key.lower() or "zero" in e.key.lower()]
rows.sort(key=lambda e: -e.count)
print(f"{'count/step':>10} {'dev us/step':>12}  op")
for e in rows[:60]:
    print(f"{e.count / 3:10.1f} {e.device_time_total / 3:12.1f}  {e.key[:110]}")
