"""The synthesise() chain of models/model.py:79-105 on native kernels end to end, on a real MI355X:
TextEncoder -> duration/alignment -> mu_y -> CFM decoder (CFG), against the oracle's pieces chained the same way
(the duration predictor and the reference encoder are outside the path: their outputs logw / c are synthetic).
Plus a seeded fuzz of the decoder over random shapes, lengths, solvers and CFG settings.  Run with ``-m gpu``."""
import numpy as np
import pytest
import torch

import oracle
from oracle.align_oracle import length_regulate as oracle_length_regulate
from oracle.inputs import make_inputs
from oracle.make_golden_text_encoder import text_inputs

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_synthesise_chain_native_vs_oracle(sd, cfg_params):
    from stabletts_amd.alignment import length_regulate
    from stabletts_amd.flow_matching import CFMDecoder
    from stabletts_amd.text_encoder import TextEncoder
    enc_sd = oracle.make_text_encoder_state_dict(2468)
    B, Tx, xl = 3, 40, [40, 31, 12]
    tok, c, lens = text_inputs(B, Tx, xl, 5)
    rng = np.random.Generator(np.random.PCG64(9))
    # ---- oracle chain (fp32 CPU)
    with torch.inference_mode():
        ox, omu_x, omask = oracle.text_encoder_forward(enc_sd, tok, c, lens)
    logw = torch.from_numpy(rng.normal(1.0, 0.5, size=(B, 1, Tx)).astype(np.float32)) * omask        # stand-in for self.dp(x, x_mask, c)
    oal = oracle_length_regulate(logw.numpy(), omask.numpy(), omu_x.numpy(), 1.0)
    mu_y, y_mask = torch.from_numpy(oal["mu_y"]), torch.from_numpy(oal["y_mask"])
    z = torch.from_numpy(rng.standard_normal(mu_y.shape).astype(np.float32))
    fs, fc = cfg_params
    kw = dict(fake_speaker=fs, fake_content=fc, cfg_strength=3.0)
    oref = oracle.cfm_forward(sd, mu_y, y_mask, 6, z, c, "euler", kw)
    # ---- native chain
    enc = TextEncoder(401, 128, 256, 1024, 4, 3, 3, 0.1, 256, operand_dtype="f16")
    enc.load_state_dict(enc_sd)
    enc = enc.cuda()
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    x, mu_x, x_mask = enc(tok.cuda(), c.cuda(), lens.cuda())
    r = length_regulate(logw.cuda(), x_mask, mu_x, 1.0)
    assert np.array_equal(r["y_lengths"].cpu().numpy(), oal["y_lengths"])
    assert np.array_equal(r["attn"][:, 0].cpu().numpy(), oal["attn"])                 # same alignment (integer path exact)
    assert _rel(r["mu_y"].cpu(), mu_y) <= 2e-3                                        # the TextEncoder's 16-bit operands
    kwg = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
    out = dec(r["mu_y"], r["y_mask"], 6, 1.0, c.cuda(), "euler", kwg, z=z.cuda()).cpu()
    assert torch.isfinite(out).all()
    assert float((out - oref).abs().max() / (oref - z).abs().max()) <= 3e-3           # encoder + decoder errors chained
    assert _rel(out, oref) <= 1e-3
    # ---- ... and the vocoder on the decoder's mel (api.py:76): native Vocos vs the numpy oracle on the ORACLE's mel
    # (random decoder weights leave the mel ~ noise z: values of a few units, fine as a vocoder input)
    import types
    from oracle import vocos_oracle as vo
    from stabletts_amd.vocos import Vocos
    vc = vo.VocosConfig
    voc = Vocos(types.SimpleNamespace(input_channels=vc.input_channels, dim=vc.dim, intermediate_dim=vc.intermediate_dim,
                                      num_layers=vc.num_layers), types.SimpleNamespace(n_fft=vc.n_fft, hop_length=vc.hop_length))
    vsd = vo.make_vocos_state_dict(77)
    voc.load_state_dict({k: torch.from_numpy(v) for k, v in vsd.items()})
    audio = voc.cuda()(out.cuda()).cpu().numpy()
    aref = vo.vocos_forward(vsd, oref.numpy())
    assert audio.shape == (B, out.shape[2] * 512)
    aerr = float(np.abs(audio - aref).max() / np.abs(aref).max())
    print(f"chain: mel {tuple(out.shape)} -> audio {audio.shape}, waveform vs oracle chain {aerr:.2e}")
    assert aerr <= 2.5e-3                                                              # decoder error + f16 vocoder operands (measured 8.6e-4)


def test_decoder_fuzz_random_shapes_solvers_cfg(sd, cfg_params):
    """Seeded fuzz: random batch sizes, lengths (ragged, with occasional tiny items), step counts,
    fixed-grid solvers and CFG strengths, f16 operands, against the oracle on the same inputs."""
    from stabletts_amd.flow_matching import CFMDecoder
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    rng = np.random.Generator(np.random.PCG64(2025))
    fs, fc = cfg_params
    worst = 0.0
    for trial in range(10):
        B = int(rng.integers(1, 6))
        T = int(rng.choice([3, 17, 64, 65, 130, 200, 257, 300]))
        lengths = [T] + [int(rng.integers(1, T + 1)) for _ in range(B - 1)]
        rng.shuffle(lengths)
        lengths = [int(v) for v in lengths]
        if max(lengths) != T:
            lengths[0] = T
        solver = str(rng.choice(["euler", "midpoint", "rk4"]))
        n = int(rng.integers(1, 5))
        cfg = None if rng.random() < 0.3 else float(rng.choice([1.0, 2.0, 3.5]))
        inp = make_inputs(B, T, seed=300 + trial, lengths=lengths)
        kw = None if cfg is None else dict(fake_speaker=fs, fake_content=fc, cfg_strength=cfg)
        ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], n, inp["z"], inp["c"], solver, kw)
        kwg = None if cfg is None else dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=cfg)
        out = dec(inp["mu"].cuda(), inp["mask"].cuda(), n, 1.0, inp["c"].cuda(), solver, kwg, z=inp["z"].cuda()).cpu()
        d = float((out - ref).abs().max() / (ref - inp["z"]).abs().max())
        worst = max(worst, d)
        assert d <= 1e-3, (trial, B, T, lengths, solver, n, cfg, d)          # north_star's bar; measured worst 4.6e-4
        pad = ~inp["mask"].bool().expand_as(out)
        assert torch.equal(out[pad], inp["z"][pad])
    print(f"fuzz worst displacement error {worst:.2e}")
