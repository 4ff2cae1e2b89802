"""Solve time over batch sizes (developer tool): median wall time of CFMDecoder.forward (10 Euler steps, CFG 3.0) per (B, T)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder
dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
dec.estimator.load_state_dict(oracle.make_state_dict(1234))
dec = dec.cuda()
fs, fc = oracle.make_cfg_params(4321)
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
out = []
for B, T in [(2, 500), (4, 500), (4, 1000), (8, 500), (8, 1000), (12, 1000), (16, 1000), (24, 1000)]:
    g = {k: v.cuda() for k, v in make_inputs(B, T, seed=0).items() if k != "lengths"}
    ts = []
    for i in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[3:])
    out.append(f"{B}x{T}: {ts[len(ts) // 2] * 1e3:.2f}")
print(os.environ.get("ST_BIG_MIN_BLOCKS", "192(default)"), " | ".join(out))
