#!/usr/bin/env python
"""The shipped default on trial with trained-like weights at benchmark size (round-4 review, item 1).

A released checkpoint has adaLN gates of O(1) (the reference zero-initialises them only at init, models/estimator.py:98-101), so the
attention / FFN branch errors reach the residual stream un-attenuated.  For every weight variant (ada_std, q/k factor) and every
ST_FUSED_FFN mode given on the command line: B=32 x T=1000 RAGGED, n=10 Euler, CFG 3.0 on the DEFAULT engine (no capture, no other
overrides: the big-grid fused FFN + weight-stationary q/k/v / out-projection + two solve parts), two rows (longest, shortest)
against the fp32 oracle run on those utterances alone: ONE evaluation and the whole SOLVE.  f16 operands.  GPU box.

    python tools/parity_trained.py [modes, default "1 3"]      e.g.  python tools/parity_trained.py 1 3
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder

torch.set_num_threads(16)
VARIANTS = [(0.02, 1.0), (0.15, 1.0), (0.15, 3.0)]


def weights(ada, qk):
    sd = oracle.make_state_dict(1234, ada_std=ada)
    for i in range(6):
        for nm in ("q", "k"):
            sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] * qk
    return sd


def main():
    modes = sys.argv[1:] or ["1", "3"]
    fs, fc = oracle.make_cfg_params(4321)
    inp = make_inputs(32, 1000, seed=0, ragged=True)
    lens = inp["mask"][:, 0].sum(-1)
    rows = [int(lens.argmax()), int(lens.argmin())]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    g = {k: v.cuda() for k, v in inp.items() if k != "lengths"}
    kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
    t = torch.tensor(0.5)
    print(f"B=32 x T=1000 ragged (rows {rows}: {int(lens[rows[0]])} / {int(lens[rows[1]])} frames), n=10 Euler, CFG 3.0, f16 operands, default engine")
    for ada, qk in VARIANTS:
        sd = weights(ada, qk)
        with torch.inference_mode():
            ref1 = oracle.decoder_forward(sd, t, sub["z"], sub["mask"], sub["mu"], sub["c"])
            ref = oracle.cfm_forward(sd, sub["mu"], sub["mask"], 10, sub["z"], sub["c"], "euler", dict(fake_speaker=fs, fake_content=fc, cfg_strength=3.0))
        for mode in modes:
            os.environ["ST_FUSED_FFN"] = mode
            d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256).cuda()
            d.estimator.load_state_dict(sd)
            with torch.no_grad():
                one = d.estimator(t.cuda(), g["z"], g["mask"], g["mu"], g["c"]).cpu()[rows]
                out = d(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"]).cpu()[rows]
            del os.environ["ST_FUSED_FFN"]
            e1 = float((one - ref1).abs().max() / ref1.abs().max())
            mel = float((out - ref).abs().max() / ref.abs().max())
            disp = float((out - ref).abs().max() / (ref - sub["z"]).abs().max())
            rms = float((out - ref).pow(2).mean().sqrt() / (ref - sub["z"]).pow(2).mean().sqrt())
            print(f"ada_std {ada:4.2f} q/k x{qk:3.1f}  ST_FUSED_FFN={mode}: one evaluation {e1:.3e} | solve: displacement {disp:.3e} (rms {rms:.3e}) mel {mel:.3e}"
                  f"  finite {bool(torch.isfinite(out).all())}", flush=True)
            del d


if __name__ == "__main__":
    main()
