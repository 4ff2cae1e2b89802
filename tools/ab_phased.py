#!/usr/bin/env python
"""A/B of the phased K loop (developer tool): same solve with ST_PHASED=0 / 1 engines -- outputs must be bitwise equal
(same accumulation order) -- and the solve time of each, interleaved."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 1000)
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
sd = oracle.make_state_dict(1234)
fs, fc = oracle.make_cfg_params(4321)
decs = []
for ph in ("0", "1"):
    os.environ["ST_PHASED"] = ph
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt).cuda()
    d.estimator.load_state_dict(sd)
    d.estimator.engine()
    decs.append(d)
g = {k: v.cuda() for k, v in make_inputs(B, T, seed=0).items() if k != "lengths"}
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
run = lambda d: d(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"])
outs = [run(d) for d in decs]
torch.cuda.synchronize()
print("bitwise equal:", torch.equal(outs[0], outs[1]), "max diff", float((outs[0] - outs[1]).abs().max()))
for rep in range(3):
    for i, d in enumerate(decs):
        for _ in range(2): run(d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): run(d)
        torch.cuda.synchronize()
        print(f"rep {rep} phased={i}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/solve", flush=True)
for i, d in enumerate(decs):
    eng = d.estimator.engine(); eng.profile_enable(True); run(d); torch.cuda.synchronize()
    pr = eng.profile_read(); eng.profile_enable(False)
    print(f"phased={i}", {k: round(v["total_ms"], 2) for k, v in pr.items() if v["launches"]})
