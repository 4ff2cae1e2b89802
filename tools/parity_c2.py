#!/usr/bin/env python
"""Parity of BASELINE config 2 as benchmarked (B=32 x T=1000, all-ones mask, 10 Euler steps, CFG 3.0) against the fp32 oracle on two
rows, for a list of ST_FUSED_FFN settings (1 = direct fused FFN, 3 = Winograd F(2,3)); f16 operands.  GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder

torch.set_num_threads(16)
sd = oracle.make_state_dict(1234)
fs, fc = oracle.make_cfg_params(4321)
inp = make_inputs(32, 1000, seed=0)
rows = [3, 29]
sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
with torch.inference_mode():
    ref = oracle.cfm_forward(sd, sub["mu"], sub["mask"], 10, sub["z"], sub["c"], "euler", dict(fake_speaker=fs, fake_content=fc, cfg_strength=3.0))
g = {k: v.cuda() for k, v in inp.items() if k != "lengths"}
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
for mode in sys.argv[1:] or ["1", "3"]:
    os.environ["ST_FUSED_FFN"] = mode
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256).cuda()
    d.estimator.load_state_dict(sd)
    out = d(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"]).cpu()[rows]
    mel = float((out - ref).abs().max() / ref.abs().max())
    disp = float((out - ref).abs().max() / (ref - sub["z"]).abs().max())
    rms = float((out - ref).pow(2).mean().sqrt() / (ref - sub["z"]).pow(2).mean().sqrt())
    print(f"ST_FUSED_FFN={mode}: mel rel {mel:.3e}  displacement rel {disp:.3e}  (rms {rms:.3e})")
