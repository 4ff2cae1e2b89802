"""Generates tests/golden/loss_grads.npz: gradients of the CFM training loss through the REAL reference modules
(/root/reference/models/flow_matching.py CFMDecoder.compute_loss, dropout off = eval mode) for the seeded case of
tests/golden/reference_outputs.npz ("loss_*": B=2, T=44).  They pin the oracle's backward arithmetic (autograd
through oracle.compute_loss) for the native backward pass that SURVEY.md section 8f ranks first; test
infrastructure only.  Run where /root/reference is mounted:

    python oracle/make_golden_grads.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "loss_grads.npz")
FULL = ["final_proj.bias", "in_proj.bias", "blocks.0.block.adaLN_modulation.2.bias", "blocks.5.time_fusion.film.bias",
        "time_mlp.layer.2.bias", "lsc_layers.1.bias", "blocks.3.block.attn.conv_q.bias"]      # stored in full


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not mounted at {REF}")
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    from oracle.make_golden import _install_torchdiffeq_standin
    _install_torchdiffeq_standin()
    from models.flow_matching import CFMDecoder               # reference, unmodified
    from oracle.weights import DecoderConfig, make_state_dict
    from oracle.inputs import make_inputs
    cfg = DecoderConfig()
    dec = CFMDecoder(cfg.noise_channels, cfg.cond_channels, cfg.hidden_channels, cfg.out_channels, cfg.filter_channels,
                     cfg.n_heads, cfg.n_layers, cfg.kernel_size, cfg.p_dropout, cfg.gin_channels).eval()
    dec.estimator.load_state_dict(make_state_dict(1234, cfg), strict=True)
    inp = make_inputs(2, 44, seed=31, lengths=[44, 29])
    x1 = make_inputs(2, 44, seed=32)["z"]
    mu = inp["mu"].clone().requires_grad_(True)
    c = inp["c"].clone().requires_grad_(True)
    torch.manual_seed(7)                                       # same draws as reference_outputs.npz loss_t_rand / loss_z
    loss, _ = dec.compute_loss(x1, inp["mask"], mu, c)
    loss.backward()
    res = {"loss_value": loss.detach().reshape(1), "grad_mu": mu.grad, "grad_c": c.grad}
    names, norms = [], []
    for name, p in dec.estimator.named_parameters():
        names.append(name)
        norms.append(float(p.grad.double().norm()))
        if name in FULL:
            res["grad." + name] = p.grad
    res["grad_norms"] = torch.tensor(norms, dtype=torch.float64)
    np.savez_compressed(OUT, names=np.array(names), **{k: v.detach().numpy() for k, v in res.items()})
    print("loss", float(loss), "params", len(names), "|grad| range", min(norms), max(norms))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
