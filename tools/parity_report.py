#!/usr/bin/env python
"""Measured parity numbers of the native path vs the fp32 oracle, both operand types (run on the GPU box).
Prints one JSON object; the figures quoted in DESIGN.md / README.md come from here."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle.inputs import make_inputs  # noqa: E402
from stabletts_amd.flow_matching import CFMDecoder  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def main():
    torch.set_num_threads(16)
    sd = oracle.make_state_dict(1234)
    fs, fc = oracle.make_cfg_params(4321)
    res = {}
    with torch.inference_mode():
        nfe_in = make_inputs(3, 257, seed=14, lengths=[257, 130, 64])
        t = torch.tensor(0.3)
        nfe_ref = oracle.decoder_forward(sd, t, nfe_in["z"], nfe_in["mask"], nfe_in["mu"], nfe_in["c"])
        c1 = make_inputs(1, 500, seed=0)
        c1_ref = oracle.cfm_forward(sd, c1["mu"], c1["mask"], 10, c1["z"], c1["c"], "euler", None)
        c2 = make_inputs(2, 1000, seed=7, lengths=[1000, 731])
        kw = dict(fake_speaker=fs, fake_content=fc, cfg_strength=3.0)
        c2_ref = oracle.cfm_forward(sd, c2["mu"], c2["mask"], 10, c2["z"], c2["c"], "euler", kw)
        for dt in ("bf16", "f16"):
            d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
            d.estimator.load_state_dict(sd)
            d = d.cuda()
            g = lambda x: x.cuda()  # noqa: E731
            out = d.estimator(t, g(nfe_in["z"]), g(nfe_in["mask"]), g(nfe_in["mu"]), g(nfe_in["c"])).cpu()
            r = {"one_nfe_rel": rel(out, nfe_ref)}
            o1 = d(g(c1["mu"]), g(c1["mask"]), 10, 1.0, g(c1["c"]), "euler", None, z=g(c1["z"])).cpu()
            r["config1_mel_rel"] = rel(o1, c1_ref)
            r["config1_displacement_rel"] = float((o1 - c1_ref).abs().max() / (c1_ref - c1["z"]).abs().max())
            kwg = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
            o2 = d(g(c2["mu"]), g(c2["mask"]), 10, 1.0, g(c2["c"]), "euler", kwg, z=g(c2["z"])).cpu()
            r["config2_rows_mel_rel"] = rel(o2, c2_ref)
            r["config2_rows_displacement_rel"] = float((o2 - c2_ref).abs().max() / (c2_ref - c2["z"]).abs().max())
            res[dt] = r
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
