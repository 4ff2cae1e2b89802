#!/usr/bin/env python
"""Developer diagnostic: per-block gradient tensors of the native training backward (debug capture) against the oracle's
autograd evaluated at the native forward's own q, k, v (oracle.attention(subst=...)).  python tools/grad_diag.py [dtype] [B] [T]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    lens = [T, int(T * 0.873), int(T * 0.655), int(T * 0.512)][:B] + [T] * max(0, B - 4)
    ada, qk = float(os.environ.get("GRAD_DIAG_ADA", "0.02")), float(os.environ.get("GRAD_DIAG_QK", "1"))
    sd = oracle.make_state_dict(1234, ada_std=ada)         # GRAD_DIAG_ADA=0.15 GRAD_DIAG_QK=6: the "trained-like" weights of the tests
    for i in range(6):
        for nm in ("q", "k"):
            sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd[f"blocks.{i}.block.attn.conv_{nm}.weight"] * qk
    print(f"weights: ada_std {ada}, q/k projections x{qk}")
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda().eval()
    inp = make_inputs(B, T, seed=81, lengths=lens)
    x1 = make_inputs(B, T, seed=82)["z"]
    g0 = torch.Generator().manual_seed(19)
    t_rand = torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)
    eng = dec.estimator.engine()
    eng.debug_capture(True)
    loss, _ = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda(), t_rand=t_rand.cuda(), z=z.cuda())
    loss.backward()
    torch.cuda.synchronize()
    H, Tp = 4, (T + 63) // 64 * 64
    tt = np.arange(Tp); pos = (tt & ~12) | ((tt & 4) << 1) | ((tt & 8) >> 1)
    print(f"dtype {dt} B={B} T={T} loss {float(loss.detach()):.6f} gradient scale per block " +
          ", ".join(f"{float(eng.debug_fetch(f'g.scale_{i}')[0]):g}" for i in range(5, -1, -1)))
    subst = []
    for i in range(6):
        qn = eng.debug_fetch(f"t{i}.q").reshape(B, H, T, 64) * (8.0 / math.log2(math.e))
        kn = eng.debug_fetch(f"t{i}.k").reshape(B, H, T, 64)
        vn = eng.debug_fetch(f"t{i}.vt").reshape(B, H, 64, Tp)[..., pos][..., :T].transpose(0, 1, 3, 2)
        subst.append({k_: torch.from_numpy(np.ascontiguousarray(v_)) for k_, v_ in (("q", qn), ("k", kn), ("v", vn))})
    pr = {k_: v_.clone().requires_grad_(True) for k_, v_ in sd.items()}
    taps = {}
    t = 1 - torch.cos(t_rand * 0.5 * torch.pi)
    y = (1 - (1 - 1e-4) * t) * z + t * x1
    u = x1 - (1 - 1e-4) * z
    pred = oracle.decoder_forward(pr, t.squeeze(), y, inp["mask"], inp["mu"], inp["c"], taps=taps, qkv_subst=subst)
    keep = {}
    for i in range(6):
        for nm in ("attn", "q", "k", "v", "x1", "x2", "x3"):
            tn = taps[f"b{i}.{nm}"]
            tn.retain_grad(); keep[(i, nm)] = tn
    l2 = torch.nn.functional.mse_loss(pred, u, reduction="sum") / (inp["mask"].sum() * 128)
    l2.backward()
    params = dict(dec.estimator.named_parameters())
    for i in range(5, -1, -1):
        scale2 = float(eng.debug_fetch(f"g.scale_{i}")[0])          # block start .. LayerNorm-2 backward (g.x2)
        scale = float(eng.debug_fetch(f"g.scale_a{i}")[0])          # attention part (d attn, dq, dk, dv, g.x1)
        da = eng.debug_fetch(f"g.dattn_{i}").reshape(B, T, H * 64) / scale                      # time-major
        ra = keep[(i, "attn")].grad.permute(0, 2, 1).numpy()
        row = [f"block {i}: d attn {rel(da, ra):.2e} (max |ref| {np.abs(ra).max():.2e}, scaled max {np.abs(ra).max() * scale:.2e}, "
               f"frac of scaled |d attn| below f16 normal {float((np.abs(ra) * scale < 6.1e-5).mean()):.3f})"]
        # native accumulators: dq_acc = dS k, dk_acc = dS^T q_scaled (q_scaled = q log2(e)/8), dv; the oracle's gradients are
        # w.r.t. the unscaled post-RoPE q, k, v:  dL/dq = dq_acc / 8,  dL/dk = dk_acc ln 2,  dL/dv = dv
        for nm, fac in (("q", 1.0 / 8.0), ("k", math.log(2.0)), ("v", 1.0)):
            got = eng.debug_fetch(f"g.d{nm}_{i}").reshape(B, H, T, 64) / scale * fac
            row.append(f"d{nm} {rel(got, keep[(i, nm)].grad.numpy()):.2e}")
        for nm in ("x2", "x1"):
            got = eng.debug_fetch(f"g.{nm}_{i}").reshape(B, T, 256) / (scale2 if nm == "x2" else scale)
            row.append(f"d{nm} {rel(got, keep[(i, nm)].grad.permute(0, 2, 1).numpy()):.2e}")
        for nm in ("q", "k", "v", "o"):
            n = f"blocks.{i}.block.attn.conv_{nm}.weight"
            row.append(f"W{nm} {rel(params[n].grad.cpu().numpy(), pr[n].grad.numpy()):.2e}")
        print("; ".join(row))
    eng.debug_capture(False)


if __name__ == "__main__":
    main()
