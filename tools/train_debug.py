#!/usr/bin/env python
"""Bring-up diagnostic of the native training path (run on the GPU box): compares the forward activations, every
captured backward intermediate and every parameter gradient of ONE compute_loss step with torch autograd through
the fp32 oracle.  Prints a table; the first stage with a large error localises a bug."""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle.inputs import make_inputs  # noqa: E402
from stabletts_amd.flow_matching import CFMDecoder  # noqa: E402


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
    B, T, lengths = 2, 44, [44, 29]
    if len(sys.argv) > 2:
        B, T = int(sys.argv[2]), int(sys.argv[3]); lengths = [T] + [max(1, T * 2 // 3)] * (B - 1)
    torch.set_num_threads(16)
    sd = oracle.make_state_dict(1234)
    inp = make_inputs(B, T, seed=31, lengths=lengths)
    x1 = make_inputs(B, T, seed=32)["z"]
    g = torch.Generator().manual_seed(7)
    t_rand = torch.rand(B, 1, 1, generator=g)
    z = torch.randn(B, 128, T, generator=g)
    # ---- oracle with autograd, taps retained
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    mu = inp["mu"].clone().requires_grad_(True)
    c = inp["c"].clone().requires_grad_(True)
    taps = {}
    t = 1 - torch.cos(t_rand * 0.5 * torch.pi)
    y = (1 - (1 - 1e-4) * t) * z + t * x1
    u = x1 - (1 - 1e-4) * z
    pred = oracle.decoder_forward(p, t.squeeze(), y, inp["mask"], mu, c, taps=taps)
    for v in taps.values():
        if v.requires_grad:
            v.retain_grad()
    loss = torch.nn.functional.mse_loss(pred, u, reduction="sum") / (inp["mask"].sum() * 128)
    loss.backward()
    # ---- native
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda().eval()
    eng = dec.estimator.engine()
    eng.debug_capture(True)
    mug = inp["mu"].cuda().requires_grad_(True)
    cg = inp["c"].cuda().requires_grad_(True)
    lossn, yn = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), mug, cg, t_rand=t_rand.cuda(), z=z.cuda())
    lossn.backward()
    torch.cuda.synchronize()
    out = {"loss": [float(lossn.detach()), float(loss.detach())]}
    gscale = float(eng.debug_fetch("g.scale")[0])
    print("gradient scale", gscale)
    _fetch = eng.debug_fetch
    eng.debug_fetch = lambda name: _fetch(name) / gscale if name.startswith("g.") else _fetch(name)
    tm = lambda v: v.detach().transpose(1, 2).contiguous().numpy()     # noqa: E731
    rows = []
    for i in range(6):
        for nm in ("x1", "x2", "x3"):
            rows.append((f"fwd t{i}.{nm}", rel(eng.debug_fetch(f"t{i}.{nm}").reshape(B, T, 256), tm(taps[f"b{i}.{nm}"]))))
    rows.append(("g.x3_5", rel(eng.debug_fetch("g.x3_5").reshape(B, T, 256), tm(taps["b5.x3"].grad))))
    for i in range(5, -1, -1):
        rows.append((f"g.x2_{i}", rel(eng.debug_fetch(f"g.x2_{i}").reshape(B, T, 256), tm(taps[f"b{i}.x2"].grad))))
        rows.append((f"g.dattn_{i}", rel(eng.debug_fetch(f"g.dattn_{i}").reshape(B, T, 256), tm(taps[f"b{i}.attn"].grad))))
        rows.append((f"g.dq_{i}", rel(eng.debug_fetch(f"g.dq_{i}").reshape(B, 4, T, 64) / 8.0, taps[f"b{i}.q"].grad.numpy())))
        rows.append((f"g.dk_{i}", rel(eng.debug_fetch(f"g.dk_{i}").reshape(B, 4, T, 64) * math.log(2.0), taps[f"b{i}.k"].grad.numpy())))
        rows.append((f"g.dv_{i}", rel(eng.debug_fetch(f"g.dv_{i}").reshape(B, 4, T, 64), taps[f"b{i}.v"].grad.numpy())))
        rows.append((f"g.x1_{i}", rel(eng.debug_fetch(f"g.x1_{i}").reshape(B, T, 256), tm(taps[f"b{i}.x1"].grad))))
        ref = taps["h0"].grad if i == 0 else taps[f"b{i - 1}.x3"].grad
        rows.append((f"g.xin_{i}", rel(eng.debug_fetch(f"g.xin_{i}").reshape(B, T, 256), tm(ref))))
    def where(name, got, ref):
        got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
        err = np.abs(got - ref)
        idx = np.unravel_index(err.argmax(), err.shape)
        print(f"{name}: max err {err.max():.3e} at {idx} got {got[idx]:.4e} ref {ref[idx]:.4e}; max|ref| {np.abs(ref).max():.3e}")
        # error profile along the time axis (axis -2 for [B,H,T,64], axis 1 for [B,T,C])
        ax = tuple(i for i in range(err.ndim) if i != (err.ndim - 2 if err.ndim == 4 else 1))
        prof = err.max(axis=ax)
        print("   per-t max err:", " ".join(f"{v:.1e}" for v in prof))
    where("dq_5", eng.debug_fetch("g.dq_5").reshape(B, 4, T, 64) / 8.0, taps["b5.q"].grad.numpy())
    where("dk_5", eng.debug_fetch("g.dk_5").reshape(B, 4, T, 64) * math.log(2.0), taps["b5.k"].grad.numpy())
    where("dv_5", eng.debug_fetch("g.dv_5").reshape(B, 4, T, 64), taps["b5.v"].grad.numpy())
    where("x2_4", eng.debug_fetch("g.x2_4").reshape(B, T, 256), tm(taps["b4.x2"].grad))
    where("xin_5", eng.debug_fetch("g.xin_5").reshape(B, T, 256), tm(taps["b4.x3"].grad))
    where("x1_5", eng.debug_fetch("g.x1_5").reshape(B, T, 256), tm(taps["b5.x1"].grad))
    eng.debug_capture(False)
    for k, v in rows:
        print(f"{k:14s} {v:.3e}")
    print("grad mu", rel(mug.grad.cpu().numpy(), mu.grad.numpy()), "grad c", rel(cg.grad.cpu().numpy(), c.grad.numpy()))
    worst = []
    for (name, q) in dec.estimator.named_parameters():
        r = rel(q.grad.cpu().numpy(), p[name].grad.numpy())
        worst.append((r, name))
    worst.sort(reverse=True)
    print("parameter gradients, worst first:")
    for r, name in worst[:40]:
        print(f"  {r:.3e}  {name}")
    print("median", float(np.median([r for r, _ in worst])))
    out["rows"] = rows; out["params"] = worst
    print("LOSS native/oracle", out["loss"])


if __name__ == "__main__":
    main()
