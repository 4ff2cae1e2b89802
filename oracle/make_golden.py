"""Generate tests/golden/*.npz from the REAL reference modules (build container only).

Run:  python -m oracle.make_golden        (needs /root/reference mounted)

The reference's models/estimator.py, models/diffusion_transformer.py and
models/flow_matching.py are imported unmodified from /root/reference (never copied).
models/flow_matching.py imports torchdiffeq, which is absent offline: a stand-in
module exposing a fixed-grid ``odeint`` (euler / midpoint / rk4-3/8, grid = t) is
registered in sys.modules first, so the real CFMDecoder.forward / cfg_wrapper /
compute_loss code runs around it.  Outputs are stored as small fp32 fixtures; inputs
and weights are regenerated from seeds by oracle.weights / oracle.inputs, so the GPU
box (which has no /root/reference) needs only the committed .npz files.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("STABLETTS_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_torchdiffeq_standin():
    def odeint(func, y0, t, method=None, rtol=None, atol=None, **_):
        ys = [y0]
        y = y0
        for i in range(len(t) - 1):
            t0, t1 = t[i], t[i + 1]
            dt = t1 - t0
            if method == "euler":
                y = y + dt * func(t0, y)
            elif method == "midpoint":
                f0 = func(t0, y)
                y = y + dt * func(t0 + 0.5 * dt, y + f0 * (0.5 * dt))
            elif method == "rk4":
                k1 = func(t0, y)
                k2 = func(t0 + dt / 3, y + dt * k1 / 3)
                k3 = func(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
                k4 = func(t1, y + dt * (k1 - k2 + k3))
                y = y + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
            else:
                raise NotImplementedError(method)
            ys.append(y)
        return torch.stack(ys)

    m = types.ModuleType("torchdiffeq")
    m.odeint = odeint
    sys.modules["torchdiffeq"] = m


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not mounted at {REF}")
    sys.path.insert(0, REF)
    _install_torchdiffeq_standin()
    from models.estimator import Decoder                      # noqa: E402  (reference, unmodified)
    from models.flow_matching import CFMDecoder               # noqa: E402
    from models import diffusion_transformer as rdt            # noqa: E402

    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from oracle.weights import DecoderConfig, make_state_dict, make_cfg_params
    from oracle.inputs import make_inputs

    torch.set_num_threads(os.cpu_count())
    os.makedirs(OUT, exist_ok=True)
    cfg = DecoderConfig()
    sd = make_state_dict(1234, cfg)
    dec = CFMDecoder(cfg.noise_channels, cfg.cond_channels, cfg.hidden_channels, cfg.out_channels,
                     cfg.filter_channels, cfg.n_heads, cfg.n_layers, cfg.kernel_size, cfg.p_dropout,
                     cfg.gin_channels).eval()
    missing = dec.estimator.load_state_dict(sd, strict=True)
    assert isinstance(dec.estimator, Decoder)
    print("reference estimator loaded:", missing, sum(p.numel() for p in dec.estimator.parameters()), "params")
    fs, fc = make_cfg_params(4321, cfg)
    est = dec.estimator
    res = {}

    with torch.inference_mode():
        # ---- g1: one NFE, scalar t, ragged mask
        inp = make_inputs(2, 70, seed=11, lengths=[70, 51])
        res["nfe_scalar_t"] = est(torch.tensor(0.3), inp["z"], inp["mask"], inp["mu"], inp["c"])
        # ---- g2: one NFE, batched t (training-style)
        inp = make_inputs(3, 40, seed=12, lengths=[40, 33, 17])
        tb = torch.tensor([0.05, 0.5, 0.93])
        res["nfe_batched_t"] = est(tb, inp["z"], inp["mask"], inp["mu"], inp["c"])
        # ---- g3: sub-modules of block 2 on seeded tensors
        inp = make_inputs(2, 37, seed=13, lengths=[37, 20])
        g = torch.Generator().manual_seed(5)
        xs = torch.randn(2, 256, 37, generator=g)
        blk = est.blocks[2].block
        res["sub_ffn"] = blk.mlp(xs, inp["mask"])
        m = inp["mask"]
        am = m.unsqueeze(1) * m.unsqueeze(-1)
        am = torch.zeros_like(am).masked_fill(am == 0, -torch.finfo(xs.dtype).max)
        res["sub_mha"] = blk.attn(xs, am)
        res["sub_block"] = blk(xs, inp["c"], m)
        res["sub_wrapper"] = est.blocks[2](xs, inp["c"], torch.randn(1, 256, generator=g), m)
        xr = torch.randn(2, 4, 37, 64, generator=g)
        res["sub_rope"] = rdt.RotaryPositionalEmbeddings(32.0)(xr)
        res["sub_temb"] = est.time_embeddings(torch.tensor([0.0, 0.123, 1.0]))
        res["sub_condproj"] = est.cond_proj(inp["mu"])

    # ---- g4..g6: full solves through the REAL CFMDecoder.forward (z drawn inside under a seed)
    def solve(name, B, T, lengths, n, solver, cfg_strength, seed):
        inp = make_inputs(B, T, seed=seed, lengths=lengths)
        torch.manual_seed(seed)
        z = torch.randn_like(inp["mu"])          # what flow_matching.py:45 will draw (temperature 1)
        torch.manual_seed(seed)
        kw = None if cfg_strength is None else dict(fake_speaker=fs, fake_content=fc, cfg_strength=cfg_strength)
        out = dec(inp["mu"], inp["mask"], n, 1.0, inp["c"], solver, kw)
        res[name] = out
        res[name + "_z"] = z

    solve("solve_euler_cfg", 2, 64, [64, 45], 4, "euler", 3.0, 21)
    solve("solve_euler_nocfg", 1, 50, [50], 5, "euler", None, 22)
    solve("solve_midpoint", 1, 48, [48], 3, "midpoint", None, 23)
    solve("solve_rk4_cfg", 2, 33, [33, 30], 2, "rk4", 2.0, 24)

    # ---- g7: compute_loss with the draws recorded
    inp = make_inputs(2, 44, seed=31, lengths=[44, 29])
    x1 = make_inputs(2, 44, seed=32)["z"]
    torch.manual_seed(7)
    t_rand = torch.rand([2, 1, 1])
    zz = torch.randn_like(x1)
    torch.manual_seed(7)
    with torch.no_grad():
        loss, y = dec.compute_loss(x1, inp["mask"], inp["mu"], inp["c"])
    res["loss_value"] = loss.reshape(1)
    res["loss_y"] = y
    res["loss_t_rand"] = t_rand
    res["loss_z"] = zz

    np.savez_compressed(os.path.join(OUT, "reference_outputs.npz"),
                        **{k: v.detach().numpy().astype(np.float32) for k, v in res.items()})
    for k, v in res.items():
        print(f"{k:22s} {tuple(v.shape)}  absmax={float(v.abs().max()):.4f}")
    print("wrote", os.path.join(OUT, "reference_outputs.npz"))


if __name__ == "__main__":
    main()
