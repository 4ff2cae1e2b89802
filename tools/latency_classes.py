#!/usr/bin/env python
"""Per-kernel-class time of one small solve (B x T, n=10 euler, CFG off by default): where a latency-bound solve spends it."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
cfg = len(sys.argv) > 3 and sys.argv[3] == "cfg"
sd = oracle.make_state_dict(1234)
fs, fc = oracle.make_cfg_params(4321)
dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256).cuda()
dec.estimator.load_state_dict(sd)
g = {k: v.cuda() for k, v in make_inputs(B, T, seed=0).items() if k != "lengths"}
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0) if cfg else None
run = lambda: dec(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"])
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
eng = dec.estimator.engine()
eng.profile_enable(True)
run(); torch.cuda.synchronize()
pr = eng.profile_read(); eng.profile_enable(False)
print(json.dumps({"B": B, "T": T, "cfg": cfg, "ms_per_solve": ms,
                  "classes_ms": {k: round(v["total_ms"], 3) for k, v in pr.items() if v["launches"]},
                  "launches": {k: v["launches"] for k, v in pr.items() if v["launches"]}}))
