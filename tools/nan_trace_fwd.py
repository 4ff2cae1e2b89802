#!/usr/bin/env python
"""Developer diagnostic: first internal tensor of ONE inference evaluation (st_estimator_forward under debug capture) that is
non-finite or far from the fp32 oracle's, for weight variants.  python tools/nan_trace_fwd.py [dtype] [B] [T]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    lens = [T, int(T * 0.873), int(T * 0.655), int(T * 0.512)][:B] + [T] * max(0, B - 4)
    inp = make_inputs(B, T, seed=81, lengths=lens)
    g0 = torch.Generator().manual_seed(19)
    torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)
    tq = torch.tensor(0.5)
    for tag, ada, qk in (("gates", 0.15, 1.0), ("qk6", 0.02, 6.0), ("both", 0.15, 6.0)):
        sd = oracle.make_state_dict(1234, ada_std=ada)
        for i in range(6):
            for nm in ("q", "k"):
                sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] * qk
        taps = {}
        with torch.no_grad():
            ref = oracle.decoder_forward(sd, tq, z, inp["mask"], inp["mu"], inp["c"], taps=taps)
        dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
        dec.estimator.load_state_dict(sd)
        dec = dec.cuda().eval()
        eng = dec.estimator.engine()
        eng.debug_capture(True)
        with torch.no_grad():
            out = dec.estimator(tq.cuda(), z.cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
        torch.cuda.synchronize()
        print(f"== {tag}: ada_std {ada}, q/k x{qk}: output nonfinite {int((~torch.isfinite(out)).sum())}, "
              f"rel err {float((out - ref).abs().max() / ref.abs().max()):.3e}")
        names = ["h0"]
        for i in range(6):
            names += [f"b{i}.{n}" for n in ("x1", "h1", "q", "k", "vt", "attn", "x2", "h2", "u", "x3")]
        names += ["v"]
        for n in names:
            try:
                a = eng.debug_fetch(n)
            except Exception as ex:
                print(f"  {n:8s} (not captured: {ex})"); continue
            bad = int((~np.isfinite(a)).sum())
            fin = a[np.isfinite(a)]
            print(f"  {n:8s} nonfinite {bad:9d} / {a.size:9d}   max |finite| {np.abs(fin).max() if fin.size else float('nan'):.3e}" +
                  (f"   oracle max {float(taps[n].abs().max()):.3e}" if n in taps else ""))
            if bad:
                break
        eng.debug_capture(False)
        del dec


if __name__ == "__main__":
    main()
