#!/usr/bin/env python
"""Are two builds of the library bitwise identical on one training step?  usage: python tools/grad_bitwise_ab.py libA.so libB.so
(developer tool: one process per library, the same seeds, B = 8 x T = 1000 ragged, dropout on; compares loss and all 116 gradients)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 3 and sys.argv[1] == "--run":
    sys.path.insert(0, ROOT)
    import oracle
    from oracle.inputs import make_inputs
    from stabletts_amd.flow_matching import CFMDecoder
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    dec.estimator.load_state_dict(oracle.make_state_dict(1234))
    dec = dec.cuda().train(True)
    inp = make_inputs(8, 1000, seed=5, ragged=True)
    x1 = make_inputs(8, 1000, seed=6)["z"]
    g0 = torch.Generator().manual_seed(3)
    t_rand = torch.rand(8, 1, 1, generator=g0); z = torch.randn(8, 128, 1000, generator=g0)
    torch.manual_seed(77)
    mu = inp["mu"].cuda().requires_grad_(True)
    loss, _ = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), mu, inp["c"].cuda(), t_rand=t_rand.cuda(), z=z.cuda())
    loss.backward()
    torch.cuda.synchronize()
    torch.save({"loss": float(loss.detach()), "gmu": mu.grad.cpu(), "g": {n: p.grad.cpu() for n, p in dec.estimator.named_parameters()}}, sys.argv[2])
    sys.exit(0)
outs = []
for i, lib in enumerate(sys.argv[1:3]):
    out = f"/tmp/grad_ab_{i}.pt"
    env = dict(os.environ)
    if lib != "default":
        env["STABLETTS_HIP_LIB"] = lib
    subprocess.run([sys.executable, os.path.abspath(__file__), "--run", out], env=env, check=True)
    outs.append(torch.load(out))
a, b = outs
diff = [n for n in a["g"] if not torch.equal(a["g"][n], b["g"][n])]
print("loss equal:", a["loss"] == b["loss"], " d mu equal:", bool(torch.equal(a["gmu"], b["gmu"])), " parameter gradients that differ:", len(diff), diff[:6])
