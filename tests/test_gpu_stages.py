"""GPU parity, one check per row of SURVEY.md section 8(a): every function on the path is compared with the fp32
oracle's intermediate ("tap") of the same name after ONE estimator evaluation, through the C ABI's debug capture
(st_debug_capture / st_debug_fetch).  Two operand types x three shapes: a short ragged batch (128-wide tiles,
first-generation QKV epilogue), a multi-tile ragged batch, and T=1000 with the 256x256-tile kernels forced
(ST_BIG_MIN_BLOCKS=0: the production kernels of BASELINE config 2, which a 2-utterance batch would not select).
Tolerances are per stage, relative to the stage's own max magnitude: 16-bit operand rounding for GEMM outputs,
tighter for fp32 residual-stream tensors.
"""
import math

import numpy as np
import pytest
import torch

import oracle
from oracle.inputs import make_inputs

pytestmark = pytest.mark.gpu

# stage -> (bf16 tolerance, f16 tolerance); names follow oracle.decoder_forward's taps
TOL = {
    "cond": (1.5e-2, 2e-3),     # a6  Decoder.cond_proj (3 convs + SiLU), 16-bit operand of in_proj
    "h0": (1.5e-2, 2e-3),       # a7  in_proj output after FiLM/LN of block 0 (a8, a9 prologue)
    "x1": (6e-3, 8e-4),         # a8  FiLM output = residual stream entering the block (fp32)
    "h1": (2e-2, 3e-3),         # a9  LayerNorm + adaLN modulate (16-bit operand of the QKV GEMM)
    "q": (2e-2, 3e-3),          # a10/a11 q projection + RoPE (pre-scaled by log2e/8)
    "k": (2e-2, 3e-3),          # a10/a11 k projection + RoPE
    "vt": (2e-2, 3e-3),         # a10 v projection (transposed, PV key order)
    "attn": (2e-2, 3e-3),       # a10 softmax(QK^T + mask) V, valid query rows
    "x2": (6e-3, 8e-4),         # a9  x + gate_msa * out_proj(attn)
    "h2": (2e-2, 3e-3),         # a9  LayerNorm2 + modulate
    "u": (2e-2, 3e-3),          # a12 FFN conv_1 + SiLU (* mask)
    "x3": (8e-3, 1e-3),         # a12 x + gate_mlp * conv_2(u)
    "lsc": (8e-3, 1e-3),        # a13 long-skip conv output
    "out": (1e-2, 1.5e-3),      # a14 final_proj (* mask) = a4 Decoder.forward
}


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def _tm(x):   # (B, C, T) -> (B, T, C)
    return x.transpose(1, 2).contiguous().numpy()


CASES = {
    "short_ragged": (2, 70, [70, 51], 11, None),
    "multi_tile": (2, 300, [300, 171], 12, None),
    "t1000_big_tiles": (2, 1000, [1000, 731], 13, "0"),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_every_stage_of_one_evaluation(sd, monkeypatch, dt, case):
    from stabletts_amd.flow_matching import CFMDecoder
    B, T, lengths, seed, big = CASES[case]
    if big is not None:
        monkeypatch.setenv("ST_BIG_MIN_BLOCKS", big)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    inp = make_inputs(B, T, seed=seed, lengths=lengths)
    t = torch.tensor(0.37)
    taps = {}
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
    eng = dec.estimator.engine()
    eng.debug_capture(True)
    try:
        out = dec.estimator(t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
        torch.cuda.synchronize()
        valid = inp["mask"][:, 0].bool().numpy()
        Tp = (T + 63) // 64 * 64
        ti = 0 if dt == "bf16" else 1
        worst = {}

        def check(stage, name, want, only_valid=False, got=None):
            got = eng.debug_fetch(name) if got is None else got
            got = np.asarray(got).reshape(want.shape)
            if only_valid:
                got, want = got[valid], want[valid]
            r = _rel(got, want)
            worst[name] = r
            assert r <= TOL[stage][ti], f"{name}: rel {r:.3e} > {TOL[stage][ti]:.1e}"

        check("cond", "cond", _tm(taps["cond"]))
        check("h0", "h0", _tm(taps["h0"]))
        tt = np.arange(Tp)
        pos = (tt & ~12) | ((tt & 4) << 1) | ((tt & 8) >> 1)          # PV key order inside each group of 16
        for i in range(6):
            b = f"b{i}."
            if i >= 3:
                try:   # not materialised when the long-skip conv carries the fused FiLM + LayerNorm epilogue
                    got = eng.debug_fetch(f"lsc{i - 3}")
                except Exception:
                    got = None
                if got is not None:
                    check("lsc", f"lsc{i - 3}", _tm(taps[f"lsc{i - 3}"]), got=got)
            check("x1", b + "x1", _tm(taps[b + "x1"]))
            check("h1", b + "h1", _tm(taps[b + "h1"]))
            check("q", b + "q", (taps[b + "q"] * (math.log2(math.e) / 8.0)).numpy())
            check("k", b + "k", taps[b + "k"].numpy())
            vt = eng.debug_fetch(b + "vt").reshape(B, 4, 64, Tp)[..., pos]      # back to frame order
            assert not vt[..., T:].any(), "vT frames [T, Tp) must be zero"
            check("vt", b + "vt", taps[b + "v"].numpy().transpose(0, 1, 3, 2), got=vt[..., :T])
            check("attn", b + "attn", _tm(taps[b + "attn"]), only_valid=True)
            check("x2", b + "x2", _tm(taps[b + "x2"]))
            check("h2", b + "h2", _tm(taps[b + "h2"]))
            check("u", b + "u", _tm(taps[b + "u"]))
            check("x3", b + "x3", _tm(taps[b + "x3"]))
        r = _rel(out.cpu().numpy(), ref.numpy())
        assert r <= TOL["out"][ti], f"out: rel {r:.3e}"
        pad = out.cpu()[~inp["mask"].bool().expand_as(out)]
        assert pad.numel() == 0 or float(pad.abs().max()) == 0.0       # estimator.py:138: output * mask
    finally:
        eng.debug_capture(False)


# ---------------------------------------------------------------- shape sweep: tile / halo / tail boundaries
SWEEP_T = [1, 2, 31, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 383, 511, 512, 513, 767]


@pytest.mark.parametrize("big", [False, True])
def test_one_evaluation_across_tile_boundaries(sd, monkeypatch, big):
    """One estimator evaluation (f16 operands, tight gate) for every T around the 64-key attention tiles, the
    128/256-frame GEMM tiles and the k=3 halo, ragged lengths [T, T//2 or 1], against the oracle.  big=True forces
    the 256x256-tile kernels wherever they are legal (ST_BIG_MIN_BLOCKS=0), big=False the 128-wide family
    (ST_BIG_MIN_BLOCKS huge)."""
    from stabletts_amd.flow_matching import CFMDecoder
    monkeypatch.setenv("ST_BIG_MIN_BLOCKS", "0" if big else "1000000000")
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    t = torch.tensor(0.61)
    worst = 0.0
    for T in SWEEP_T:
        inp = make_inputs(2, T, seed=100 + T, lengths=[T, max(1, T // 2)])
        with torch.inference_mode():
            ref = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
        out = dec.estimator(t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
        assert torch.isfinite(out).all(), f"T={T}"
        r = _rel(out.numpy(), ref.numpy())
        worst = max(worst, r)
        assert r <= 2e-3, f"T={T}: rel {r:.3e}"
        pad = out[~inp["mask"].bool().expand_as(out)]
        assert pad.numel() == 0 or float(pad.abs().max()) == 0.0, f"T={T}: padded frames must be exactly zero"
