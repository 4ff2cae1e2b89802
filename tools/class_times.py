#!/usr/bin/env python
"""Per-class kernel times of one headline solve (developer tool; honours STABLETTS_HIP_LIB / ST_PHASED)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 1000)
d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256).cuda()
d.estimator.load_state_dict(oracle.make_state_dict(1234))
fs, fc = oracle.make_cfg_params(4321)
RAGGED = os.environ.get("CLASS_TIMES_RAGGED") == "1"       # len ~ U{0.6 T .. T}, like bench.py --ragged
g = {k: v.cuda() for k, v in make_inputs(B, T, seed=0, ragged=RAGGED).items() if k != "lengths"}
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
run = lambda: d(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"])
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
eng = d.estimator.engine(); eng.profile_enable(True); run(); torch.cuda.synchronize()
pr = eng.profile_read()
print("ragged" if RAGGED else "all-ones", os.environ.get("STABLETTS_HIP_LIB", "default").split("/")[-1], "phased=" + os.environ.get("ST_PHASED", "0"), f"{ms:.2f} ms",
      {k: round(v["total_ms"], 2) for k, v in pr.items() if v["launches"] and k in ("prep", "prenet", "in_proj", "qkv_rope", "attention", "out_proj", "ffn_conv1", "ffn_conv2", "lsc_conv", "final_proj", "ode_update")})
