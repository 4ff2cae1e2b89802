#!/usr/bin/env python
"""Which 16-bit operand class carries how much of the one-evaluation error (CPU, fp32 oracle only; nothing of the product path).

The oracle is run with the inputs AND weights of ONE class of convolutions rounded to f16 (fp32 accumulation, as the MFMA kernels
do) and everything else in fp32; the attention operands q, k, v likewise.  Weights with adaLN gates of O(1) (ada_std 0.15) by default.
    python tools/operand_attribution.py [ada_std] [qk_factor]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle import estimator_oracle as eo
from oracle.inputs import make_inputs

torch.set_num_threads(min(16, os.cpu_count() or 1))
ada = float(sys.argv[1]) if len(sys.argv) > 1 else 0.15
qk = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
r16 = lambda x: x.half().float()
sd = oracle.make_state_dict(1234, ada_std=ada)
for i in range(6):
    for nm in ("q", "k"):
        sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] * qk
inp = make_inputs(2, 1000, seed=81, lengths=[1000, 655])
t = torch.tensor(0.5)


def cls_of(name):
    if "attn.conv_o" in name: return "out-proj"
    if "attn.conv_" in name: return "q/k/v projection"
    if "mlp.conv_1" in name: return "FFN conv_1"
    if "mlp.conv_2" in name: return "FFN conv_2"
    if name.startswith("lsc_layers"): return "long-skip convs"
    if name.startswith("cond_proj"): return "cond prenet"
    if name.startswith("in_proj"): return "in_proj"
    if name.startswith("final_proj"): return "final_proj"
    return None


by_id = {id(v): cls_of(k) for k, v in sd.items() if k.endswith(".weight") and cls_of(k)}
real_conv1d = F.conv1d
active = set()


def conv1d(x, w, b=None, *a, **kw):
    c = by_id.get(id(w))
    if c in active:
        x, w = r16(x), r16(w)
    elif (c, "x") in active:      # only the activation operand rounded (the weight as a hi + lo pair: exact to 2^-22)
        x = r16(x)
    elif (c, "w") in active:
        w = r16(w)
    return real_conv1d(x, w, b, *a, **kw)


eo.F.conv1d = conv1d
with torch.inference_mode():
    ref = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
    rel = lambda o: float((o - ref).abs().max() / ref.abs().max())
    classes = ["q/k/v projection", "out-proj", "FFN conv_1", "FFN conv_2", "long-skip convs", "in_proj", "final_proj", "cond prenet"]
    print(f"ada_std {ada}, q/k x{qk}, B=2 x T=1000: one-evaluation error with ONE operand class in f16 (everything else fp32)")
    tot = 0.0
    for c in classes:
        active.clear(); active.add(c)
        e = rel(oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"]))
        tot += e * e
        print(f"  {c:18s} {e:.2e}", flush=True)
    for c in ("out-proj", "q/k/v projection", "FFN conv_1", "FFN conv_2"):
        for side in ("x", "w"):
            active.clear(); active.add((c, side))
            e = rel(oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"]))
            print(f"  {c:18s} only the {'activation' if side == 'x' else 'weight'} operand in f16: {e:.2e}", flush=True)
    active.clear()
    f = lambda q, k, v: dict(q=r16(q), k=r16(k), v=r16(v))
    e = rel(oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], qkv_subst=[f] * 6))
    tot += e * e
    print(f"  {'attention q, k, v':18s} {e:.2e}")
    active.update(classes)
    e = rel(oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], qkv_subst=[f] * 6))
    print(f"  {'ALL of them':18s} {e:.2e}   (root-sum-square of the rows: {tot ** 0.5:.2e})")
    # the native path: in_proj / final_proj take split-precision operands (exact to ~2^-22)
    native = [c for c in classes if c not in ("in_proj", "final_proj")]
    active.clear(); active.update(native)
    e = rel(oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], qkv_subst=[f] * 6))
    print(f"  as the native path rounds (in_proj / final_proj split): {e:.2e}")
    active.discard("out-proj"); active.add(("out-proj", "x"))
    e = rel(oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], qkv_subst=[f] * 6))
    print(f"  ... with the out-proj WEIGHT as a hi + lo pair: {e:.2e}")
    active.discard(("out-proj", "x"))
    e = rel(oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], qkv_subst=[f] * 6))
    print(f"  ... with both out-proj operands split: {e:.2e}")
