#!/usr/bin/env python
"""CPU experiment (no GPU, nothing of the product path): would the FFN's k = 3 convolutions survive a Winograd F(2,3)
formulation along the frame axis under the 1e-3 parity bar?

F(2,3) computes two output frames from four input frames with 4 instead of 6 multiplications per (cout, cin) pair: the
FFN's MFMA work (half of a solve, power-limited) would fall by a third.  The price is numerical: the MFMA operands become
SUMS of two 16-bit activations (rounded to 16 bits again) and half-sums of three weights, and every output is a signed sum
of three products.  This script runs the fp32 oracle with the FFN convolutions replaced by emulations of
  direct : operands rounded to the 16-bit type, fp32 accumulation (what the shipped kernels do)
  wino   : F(2,3) with the transformed operands rounded to the 16-bit type, fp32 accumulation and output transform
and reports the one-evaluation and displacement errors of each against the plain fp32 oracle (everything outside the FFN
stays fp32, so the figures isolate the FFN's contribution; the shipped f16 path's TOTAL is 4.4e-4 / 4.2e-4).

    python tools/winograd_numerics.py [f16|bf16]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import estimator_oracle as eo  # noqa: E402
from oracle.inputs import make_inputs  # noqa: E402

DT = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]


def r16(x):
    return x.to(DT).float()


def conv_direct(x, w, b):
    return F.conv1d(r16(x), r16(w), b, padding=1)


def conv_wino(x, w, b, derived=False):
    """y[t] = sum_j w[:, :, j] x[t + j - 1]; tiles of two outputs (t = 2i, 2i + 1) from d0..d3 = x[2i - 1 .. 2i + 2]."""
    B, C, T = x.shape
    Te = T + (T & 1)
    xp = F.pad(r16(x), (1, 1 + Te - T))                       # xp[s] = x[s - 1]; the activations ARE 16-bit values
    d0, d1, d2, d3 = (xp[:, :, j:j + Te:2] for j in range(4))  # (B, C, Te / 2)
    g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
    u = [r16(g0), r16((g0 + g1 + g2) * 0.5), r16((g0 - g1 + g2) * 0.5), r16(g2)]
    if derived:     # three stored planes; the fourth from the 16-bit ones in 16-bit arithmetic: U2 = (U0 + U3) - U1
        u[2] = r16(r16(u[0] + u[3]) - u[1])
    v = [r16(d0 - d2), r16(d1 + d2), r16(d2 - d1), r16(d1 - d3)]
    m = [torch.einsum("oc,bct->bot", u[j], v[j]) for j in range(4)]
    y = torch.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]], dim=3).reshape(B, -1, Te)[:, :, :T]
    return y + b[None, :, None]


F43_BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                       [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
F43_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                      [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float32)
F43_AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                      dtype=torch.float32)


def conv_wino43(x, w, b):
    """F(4,3) (Lavin & Gray's matrices): 6 instead of 12 multiplications per four output frames."""
    B, C, T = x.shape
    Te = (T + 3) // 4 * 4
    xp = F.pad(r16(x), (1, 1 + Te - T))
    d = torch.stack([xp[:, :, j:j + Te:4] for j in range(6)], dim=0)          # (6, B, C, Te / 4)
    v = r16(torch.einsum("ij,jbct->ibct", F43_BT, d))
    u = r16(torch.einsum("ij,ocj->ioc", F43_G, w))
    m = torch.einsum("ioc,ibct->ibot", u, v)
    y = torch.einsum("ki,ibot->botk", F43_AT, m).reshape(B, -1, Te)[:, :, :T]
    return y + b[None, :, None]


def make_ffn(conv):
    def ffn(sd, prefix, x, mask, k=3, taps=None, drop=None):
        h = conv(x * mask, sd[prefix + "conv_1.weight"], sd[prefix + "conv_1.bias"])
        h = F.silu(h)
        h = conv(h * mask, sd[prefix + "conv_2.weight"], sd[prefix + "conv_2.bias"])
        return h * mask
    return ffn


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = oracle.make_state_dict(1234)
    plain = eo.ffn
    with torch.inference_mode():
        # self-check of the algebra in fp64-free form: un-rounded F(2,3) equals the direct convolution to fp32 rounding
        x = torch.randn(2, 256, 37)
        w = torch.randn(1024, 256, 3) * 0.02
        b = torch.randn(1024)
        global r16
        keep, r16 = r16, (lambda t: t)
        err = (conv_wino(x, w, b) - F.conv1d(x, w, b, padding=1)).abs().max().item()
        err43 = (conv_wino43(x, w, b) - F.conv1d(x, w, b, padding=1)).abs().max().item()
        r16 = keep
        assert err < 1e-4 and err43 < 1e-3, (err, err43)

        nfe_in = make_inputs(3, 257, seed=14, lengths=[257, 130, 64])
        t = torch.tensor(0.3)
        c1 = make_inputs(1, 500, seed=0)
        for ada_std in (0.02, 0.15):                    # parity weights / O(1) adaLN gates (test_strong_gates_ada_std_015)
            sd = oracle.make_state_dict(1234, ada_std=ada_std)
            eo.ffn = plain
            nfe_ref = oracle.decoder_forward(sd, t, nfe_in["z"], nfe_in["mask"], nfe_in["mu"], nfe_in["c"])
            c1_ref = oracle.cfm_forward(sd, c1["mu"], c1["mask"], 10, c1["z"], c1["c"], "euler", None)
            for name, conv in (("direct", conv_direct), ("F(2,3)", conv_wino), ("F(2,3)d", lambda x, w, b: conv_wino(x, w, b, True)), ("F(4,3)", conv_wino43)):
                eo.ffn = make_ffn(conv)
                out = oracle.decoder_forward(sd, t, nfe_in["z"], nfe_in["mask"], nfe_in["mu"], nfe_in["c"])
                o1 = oracle.cfm_forward(sd, c1["mu"], c1["mask"], 10, c1["z"], c1["c"], "euler", None)
                one = float((out - nfe_ref).abs().max() / nfe_ref.abs().max())
                rms = float((out - nfe_ref).pow(2).mean().sqrt() / nfe_ref.pow(2).mean().sqrt())
                disp = float((o1 - c1_ref).abs().max() / (c1_ref - c1["z"]).abs().max())
                print(f"ada_std {ada_std}  {name:7s} one_nfe_rel {one:.2e} (rms {rms:.2e})   config1_displacement_rel {disp:.2e}")
        eo.ffn = plain


if __name__ == "__main__":
    main()
