#!/usr/bin/env python
"""Is THIS checkpoint inside what the native f16 / bf16 operands can represent, and how close is the native path to fp32 on it?

For someone switching a trained StableTTS model over: loads the decoder weights of a reference checkpoint
(`decoder.estimator.*` keys of models/model.py's state_dict, or a bare estimator state_dict), then on synthetic inputs of the model's
own shapes
  * runs the fp32 oracle (CPU) with taps and reports the magnitudes that matter for 16-bit operands: max |u| of every FFN
    intermediate (f16 overflows at 65,504; the opt-in Winograd FFN at 32,752), max |q.k| score (peaky attention: section 2 of
    DESIGN.md), max |x| of the residual stream;
  * runs ONE evaluation and a 10-step Euler solve with CFG natively (GPU) with each operand type and prints the parity figures
    (one evaluation: max|v - v_ref| / max|v_ref|; solve: displacement metric) and whether the output is finite.
Nothing here is part of the product path; the oracle is test infrastructure.

    python tools/validate_checkpoint.py --ckpt checkpoint_0.pt [--frames 500] [--items 2] [--no-gpu]
    python tools/validate_checkpoint.py --seeded 0.15            # no checkpoint at hand: the seeded weights with adaLN std 0.15
"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle.inputs import make_inputs  # noqa: E402


def load_decoder_weights(path):
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    pref = "decoder.estimator."
    if any(k.startswith(pref) for k in sd):
        est = {k[len(pref):]: v.float() for k, v in sd.items() if k.startswith(pref)}
        fs = sd.get("fake_speaker"); fc = sd.get("fake_content")
    else:
        est, fs, fc = {k: v.float() for k, v in sd.items()}, None, None
    return est, fs, fc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt")
    ap.add_argument("--seeded", type=float, default=None, help="use the seeded test weights with this adaLN std instead of a checkpoint")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--items", type=int, default=2)
    ap.add_argument("--no-gpu", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    if args.ckpt:
        sd, fs, fc = load_decoder_weights(args.ckpt)
    else:
        sd, fs, fc = oracle.make_state_dict(1234, ada_std=args.seeded if args.seeded is not None else 0.02), None, None
    n_layers = sum(1 for k in sd if k.endswith("time_fusion.film.weight"))
    C = sd["in_proj.weight"].shape[0]; M = sd["final_proj.weight"].shape[0]; F_ = sd["blocks.0.block.mlp.conv_1.weight"].shape[0]
    G = sd["blocks.0.block.adaLN_modulation.2.weight"].shape[1] if "blocks.0.block.adaLN_modulation.0.weight" not in sd else sd["blocks.0.block.adaLN_modulation.0.weight"].shape[1]
    print(f"decoder: {n_layers} blocks, hidden {C}, filter {F_}, n_feats {M}, gin {G}, {sum(v.numel() for v in sd.values()):,} parameters")
    if fs is None or fc is None:
        fs, fc = oracle.make_cfg_params(4321)
        print("(no fake_speaker / fake_content in the file: seeded CFG null parameters)")
    fs, fc = fs.float().reshape(1, -1), fc.float().reshape(1, -1, 1)
    if (M, G) != (128, 256):
        print("note: the synthetic inputs below are made for n_feats 128 / gin 256"); return
    lengths = [args.frames] + [max(1, int(args.frames * 0.73))] * (args.items - 1)
    inp = make_inputs(args.items, args.frames, seed=7, lengths=lengths)
    t = torch.tensor(0.5)
    taps = {}
    with torch.inference_mode():
        ref1 = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
        kw = dict(fake_speaker=fs, fake_content=fc, cfg_strength=3.0)
        ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 10, inp["z"], inp["c"], "euler", kw)
    umax = max(float(taps[f"b{i}.u"].abs().max()) for i in range(n_layers))
    xmax = max(float(taps[f"b{i}.x3"].abs().max()) for i in range(n_layers))
    smax = max(float((taps[f"b{i}.q"] @ taps[f"b{i}.k"].transpose(-1, -2)).abs().max()) / math.sqrt(C // 4) for i in range(n_layers))
    gate = max(float(sd[f"blocks.{i}.block.adaLN_modulation.2.weight"].abs().max()) for i in range(n_layers))
    print(f"fp32 oracle, one evaluation at t = 0.5 ({args.items} x {args.frames} frames): max |FFN intermediate u| {umax:.3g} "
          f"(f16 limit 65,504; Winograd opt-in 32,752), max |residual stream| {xmax:.3g}, max |score| {smax:.3g} (natural units), "
          f"max |adaLN weight| {gate:.3g}")
    if umax > 32752:
        print("  -> the opt-in Winograd FFN (ST_FUSED_FFN=3) would overflow on this checkpoint" + ("; so would f16 operands: use operand_dtype='bf16'" if umax > 65504 else ""))
    if smax > 60:
        print("  -> attention is close to an arg-max on these inputs: 16-bit q / k / v alone move the fp32 result by > 1e-3 (DESIGN.md section 2)")
    if args.no_gpu or not torch.cuda.is_available():
        print("(no HIP device: native part skipped)"); return
    from stabletts_amd.flow_matching import CFMDecoder
    g = {k: v.cuda() for k, v in inp.items() if k != "lengths"}
    kwg = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
    for dt in ("f16", "bf16"):
        d = CFMDecoder(M, M, C, M, F_, 4, n_layers, 3, 0.1, G, operand_dtype=dt, check_finite=False)
        d.estimator.load_state_dict(sd)
        d = d.cuda()
        with torch.no_grad():
            one = d.estimator(t.cuda(), g["z"], g["mask"], g["mu"], g["c"]).cpu()
            out = d(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kwg, z=g["z"]).cpu()
        nonfinite = d.estimator.engine().output_nonfinite(torch.cuda.current_stream().cuda_stream)
        e1 = float((one - ref1).abs().max() / ref1.abs().max())
        disp = float((out - ref).abs().max() / (ref - inp["z"]).abs().max())
        print(f"native, {dt} operands: one evaluation {e1:.2e}, solve displacement {disp:.2e}, finite {bool(torch.isfinite(out).all()) and not nonfinite}"
              + ("   <- meets 1e-3" if max(e1, disp) <= 1e-3 else ""))


if __name__ == "__main__":
    main()
