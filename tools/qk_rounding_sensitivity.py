#!/usr/bin/env python
"""Which 16-bit operand carries the one-evaluation error with peaky attention (CPU, fp32 oracle only).

The oracle is run with ONE operand family of the attention rounded to f16 and everything else in fp32 -- q and k (post-RoPE, as the
native kernels round them), v, or all three -- for the weight variants of tools/parity_trained.py, at B=2 x T=1000 (ragged).  The
deviation from the un-rounded oracle is what ANY implementation with f16 attention operands inherits, before its own kernels add
anything:  python tools/qk_rounding_sensitivity.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs

torch.set_num_threads(min(16, os.cpu_count() or 1))
r16 = lambda x: x.half().float()
inp = make_inputs(2, 1000, seed=81, lengths=[1000, 655])
t = torch.tensor(0.5)
for ada, qk in [(0.02, 1.0), (0.15, 1.0), (0.15, 3.0), (0.02, 6.0)]:
    sd = oracle.make_state_dict(1234, ada_std=ada)
    for i in range(6):
        for nm in ("q", "k"):
            sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] * qk
    with torch.inference_mode():
        taps = {}
        ref = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
        smax = (taps["b0.q"] @ taps["b0.k"].transpose(-1, -2)).amax(-1) / 8.0
        row = [f"ada_std {ada:4.2f} q/k x{qk:3.1f} (block-0 row maxima of the scores: median {float(smax.median()):5.1f}, max {float(smax.max()):5.1f}):"]
        for name, f in [("q,k -> f16", lambda q, k, v: dict(q=r16(q), k=r16(k), v=v)), ("v -> f16", lambda q, k, v: dict(q=q, k=k, v=r16(v))),
                        ("q,k,v -> f16", lambda q, k, v: dict(q=r16(q), k=r16(k), v=r16(v)))]:
            out = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], qkv_subst=[f] * 6)
            row.append(f"{name}: {float((out - ref).abs().max() / ref.abs().max()):.2e}")
    print("  ".join(row), flush=True)
