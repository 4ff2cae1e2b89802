"""Which 16-bit operand carries the end-to-end error of the conv_q / conv_k weight gradients at config-5 frame counts?  (Round 6; CPU only.)
The oracle's autograd at B = 4 x T = 1000 ragged (the inputs of tests/test_gpu_training.py::size_case) is evaluated with one rounding at a time,
straight through (oracle.attention(subst=...) for q, k, v; a patched oracle.mha for the projections' input h1 and weights), each against the
un-rounded autograd:
    q, k, v rounded to f16 (the native path through round 5)   1.8e-1        q, k rounded, v exact   1.2e-3
    q, k exact, v rounded                                      1.8e-1        h1 rounded              1.1e-1        projection weights rounded   1.0e-3
-> split-precision q / k operands (built for inference, attention.hip) would buy nothing in training; v's 16-bit error does the damage, its own
rounding and the one it inherits from h1.  The training forward therefore computes v = W_v (h_hi + h_lo) and keeps it as a hi + lo operand pair
(ST_TRAIN_VLO, csrc/engine_train.cpp): 2.5e-1 -> 2.9e-2 on the hardware (profiles/r06_v_hi_lo_training.txt).
    python tools/train_qk_split_estimate.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from oracle.inputs import make_inputs  # noqa: E402


def r16(x):
    return x.detach().to(torch.float16).to(torch.float32)


def grads(sd, inp, x1, t_rand, z, subst):
    with torch.enable_grad():
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        loss, _ = oracle.compute_loss(pr, x1, inp["mask"], inp["mu"], inp["c"], t_rand, z, qkv_subst=None if subst is None else [subst] * 6)
        loss.backward()
    return {k: v.grad.numpy() for k, v in pr.items()}


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd = oracle.make_state_dict(1234)
    B, T, lens = 4, 1000, [1000, 873, 655, 512]
    inp = make_inputs(B, T, seed=81, lengths=lens)
    x1 = make_inputs(B, T, seed=82)["z"]
    g0 = torch.Generator().manual_seed(19)
    t_rand = torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)
    ref = grads(sd, inp, x1, t_rand, z, None)
    cases = {"q, k, v rounded to f16 (the native training path)": lambda q, k, v: {"q": r16(q), "k": r16(k), "v": r16(v)},
             "q, k rounded, v exact": lambda q, k, v: {"q": r16(q), "k": r16(k), "v": v.detach()},
             "q, k exact, v rounded (what split q / k operands would leave)": lambda q, k, v: {"q": q.detach(), "k": k.detach(), "v": r16(v)}}
    qk = [n for n in ref if ".attn.conv_q." in n or ".attn.conv_k." in n]
    # ... and the roundings UPSTREAM of q, k, v that hi + lo operand pairs of q, k, v cannot remove: the projections' 16-bit input h1 and weights
    import oracle.estimator_oracle as eo
    import torch.nn.functional as F
    orig_mha = eo.mha

    def ste16(x):
        return x + (r16(x) - x).detach()

    def mha_rounded(which):
        def mha(sd_, prefix, x, mask, n_heads=4, taps=None, drop=None, subst=None):
            xr = ste16(x) if "h1" in which else x
            w = (lambda n: ste16(sd_[prefix + n])) if "w" in which else (lambda n: sd_[prefix + n])
            q = F.conv1d(xr, w("conv_q.weight"), sd_[prefix + "conv_q.bias"])
            k = F.conv1d(xr, w("conv_k.weight"), sd_[prefix + "conv_k.bias"])
            v = F.conv1d(xr, w("conv_v.weight"), sd_[prefix + "conv_v.bias"])
            a, _ = eo.attention(q, k, v, mask, n_heads, drop, subst)
            return F.conv1d(a, sd_[prefix + "conv_o.weight"], sd_[prefix + "conv_o.bias"])
        return mha
    upstream = {"h1 (the q/k/v projections' input) rounded to f16, everything else fp32": ("h1",),
                "the q/k/v projection weights rounded to f16, everything else fp32": ("w",)}
    print(f"oracle autograd, B={B} x T={T} ragged {lens}; conv_q / conv_k gradients against the un-rounded oracle: worst max-norm error, min cosine; every other tensor: worst")
    for name, fn in cases.items():
        g = grads(sd, inp, x1, t_rand, z, fn)
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
        cos = lambda a, b: float((a.ravel().astype(np.float64) @ b.ravel().astype(np.float64)) / (np.linalg.norm(a.ravel().astype(np.float64)) * np.linalg.norm(b.ravel().astype(np.float64))))
        wq = max(rel(g[n], ref[n]) for n in qk); cq = min(cos(g[n], ref[n]) for n in qk)
        wo = max(rel(g[n], ref[n]) for n in ref if n not in qk)
        print(f"  {name}: q/k {wq:.2e}, cosine {cq:.6f}; others {wo:.2e}")
    for name, which in upstream.items():
        eo.mha = mha_rounded(which)
        try:
            g = grads(sd, inp, x1, t_rand, z, None)
        finally:
            eo.mha = orig_mha
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
        cos = lambda a, b: float((a.ravel().astype(np.float64) @ b.ravel().astype(np.float64)) / (np.linalg.norm(a.ravel().astype(np.float64)) * np.linalg.norm(b.ravel().astype(np.float64))))
        wq = max(rel(g[n], ref[n]) for n in qk); cq = min(cos(g[n], ref[n]) for n in qk)
        wo = max(rel(g[n], ref[n]) for n in ref if n not in qk)
        print(f"  {name}: q/k {wq:.2e}, cosine {cq:.6f}; others {wo:.2e}")


if __name__ == "__main__":
    main()
