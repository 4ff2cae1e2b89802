"""Parity of the native gfx950 path (through the C ABI / ctypes shim) against the oracle and the
committed reference fixtures.  Needs a real MI355X: run with ``-m gpu``.

Gates (fixed noise, max over the whole tensor):
  * one vector-field evaluation: max|v - v_ref| / max|v_ref|;
  * full ODE solve: the DISPLACEMENT metric max|out - ref| / max|ref - z| -- what the decoder actually computes.
    (max|out - ref| / max|ref| is also asserted, but at random weights the mel is ~95 % the input noise z, so that
    number alone would tolerate a 2 % error of the decoder's own contribution.)
  north_star's bar is 1e-3.  **f16 operands -- the DEFAULT of every module and the dtype bench.py headlines -- meet
  it** on both metrics: gates 7e-4 (measured 2-4.4e-4), 1e-3 with O(1) adaLN gates (measured 6.1e-4).  **bf16 operands
  (opt-in) do not**: 8 mantissa bits give ~4e-3 per evaluation / displacement
  (2^-9 per operand rounding through ~40 GEMMs; the split-precision operands of in_proj / final_proj remove only
  the un-gated part); their gates below are regression guards, not a claim of meeting 1e-3.
"""
import math

import numpy as np
import pytest
import torch

import oracle
from oracle.inputs import make_inputs

pytestmark = pytest.mark.gpu

NFE_TOL = {"bf16": 1e-2, "f16": 7e-4}
MEL_TOL = 1e-3
DISP_TOL = {"bf16": 1e-2, "f16": 7e-4}


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def _disp(out, ref, z):
    return float((out.double() - ref.double()).abs().max() / (ref.double() - z.double()).abs().max())


@pytest.fixture(scope="module")
def decoders(sd):
    from stabletts_amd.flow_matching import CFMDecoder
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    out = {}
    for dt in ("bf16", "f16"):
        d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
        d.estimator.load_state_dict(sd)
        out[dt] = d.to("cuda:0")
    return out


def _cfg(cfg_params, strength, cuda):
    fs, fc = cfg_params
    if strength is None:
        return None
    if cuda:
        fs, fc = fs.cuda(), fc.cuda()
    return dict(fake_speaker=fs, fake_content=fc, cfg_strength=strength)


def _solve(dec, inp, n, solver, kw, z):
    return dec(inp["mu"].cuda(), inp["mask"].cuda(), n, 1.0, inp["c"].cuda(), solver, kw, z=z.cuda()).cpu()


# ---------------------------------------------------------------- native library is what runs
def test_native_library_loaded(decoders):
    import ctypes
    from stabletts_amd import _lib
    assert isinstance(_lib.load(), ctypes.CDLL)
    eng = decoders["bf16"].estimator.engine()
    assert eng.num_params() == 116 and eng.device_bytes() > 30e6      # packed 16-bit weights (the fp32 tensors stay torch's)
    maps = open("/proc/self/maps").read()
    assert "libstabletts_hip.so" in maps


# ---------------------------------------------------------------- one evaluation vs oracle
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,T,lengths,seed", [(2, 70, [70, 51], 11), (3, 257, [257, 130, 64], 14), (1, 1, [1], 15)])
def test_one_nfe_scalar_t(decoders, sd, dt, B, T, lengths, seed):
    inp = make_inputs(B, T, seed=seed, lengths=lengths)
    t = torch.tensor(0.3)
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
    out = decoders[dt].estimator(t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
    assert _rel(out, ref) <= NFE_TOL[dt]
    pad = ~inp["mask"].bool().expand_as(out)
    assert float(out[pad].abs().max()) == 0.0 if pad.any() else True     # estimator.py:138


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_one_nfe_batched_t_vs_reference_fixture(decoders, golden, dt):
    """Training-style per-item t (flow_matching.py:99) against the REAL reference's output."""
    inp = make_inputs(3, 40, seed=12, lengths=[40, 33, 17])
    tb = torch.tensor([0.05, 0.5, 0.93])
    out = decoders[dt].estimator(tb.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
    assert _rel(out, torch.from_numpy(golden["nfe_batched_t"])) <= NFE_TOL[dt]


# ---------------------------------------------------------------- solves vs the reference fixtures
CASES = [
    ("solve_euler_cfg", 2, 64, [64, 45], 4, "euler", 3.0, 21),
    ("solve_euler_nocfg", 1, 50, [50], 5, "euler", None, 22),
    ("solve_midpoint", 1, 48, [48], 3, "midpoint", None, 23),
    ("solve_rk4_cfg", 2, 33, [33, 30], 2, "rk4", 2.0, 24),
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("name,B,T,lengths,n,solver,cfg,seed", CASES)
def test_solve_vs_reference_fixture(decoders, cfg_params, golden, dt, name, B, T, lengths, n, solver, cfg, seed):
    inp = make_inputs(B, T, seed=seed, lengths=lengths)
    z = torch.from_numpy(golden[name + "_z"])
    ref = torch.from_numpy(golden[name])
    out = _solve(decoders[dt], inp, n, solver, _cfg(cfg_params, cfg, True), z)
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= MEL_TOL
    assert _disp(out, ref, z) <= DISP_TOL[dt]


# ---------------------------------------------------------------- BASELINE config 1 (B=1, T=500, n=10, CFG off)
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_config1_b1_t500_euler10(decoders, sd, dt):
    inp = make_inputs(1, 500, seed=0)
    ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 10, inp["z"], inp["c"], "euler", None)
    out = _solve(decoders[dt], inp, 10, "euler", None, inp["z"])
    assert _rel(out, ref) <= MEL_TOL
    assert _disp(out, ref, inp["z"]) <= DISP_TOL[dt]


# ---------------------------------------------------------------- a paragraph-length utterance (T beyond every BASELINE size)
def test_long_utterances_beyond_the_benchmark_length(decoders, sd, cfg_params):
    """api.py synthesises whole paragraphs: T = 4000 frames (46 s of audio at hop 512 / 44.1 kHz) next to a shorter item -- 63 key tiles
    per attention row, RoPE table grown past its first size, the frame-tile / part rules at a length no benchmark exercises.  One
    evaluation and a short CFG solve against the fp32 oracle at the usual gates."""
    inp = make_inputs(2, 4000, seed=91, lengths=[4000, 2771])
    t = torch.tensor(0.45)
    with torch.inference_mode():
        ref1 = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
        ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 2, inp["z"], inp["c"], "euler", _cfg(cfg_params, 3.0, False))
    for dt in ("bf16", "f16"):
        out1 = decoders[dt].estimator(t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
        assert _rel(out1, ref1) <= NFE_TOL[dt], (dt, _rel(out1, ref1))
        out = _solve(decoders[dt], inp, 2, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
        assert torch.isfinite(out).all()
        assert _rel(out, ref) <= MEL_TOL and _disp(out, ref, inp["z"]) <= DISP_TOL[dt], (dt, _rel(out, ref), _disp(out, ref, inp["z"]))
        assert torch.equal(out[1, :, 2771:], inp["z"][1, :, 2771:])      # the velocity is exactly zero past the mask: the noise stays
        print(f"[{dt}] T=4000: one evaluation {_rel(out1, ref1):.2e}; 2-step CFG solve mel {_rel(out, ref):.2e}, displacement {_disp(out, ref, inp['z']):.2e}")


# ---------------------------------------------------------------- many short utterances (item count beyond every BASELINE size)
def test_many_short_utterances_in_one_batch(decoders, sd, cfg_params):
    """B = 200 ragged utterances of <= 96 frames (400 CFG-doubled items: per-item tables, list strides and grid rules at an item count
    no benchmark reaches), one evaluation and a 2-step CFG solve against the fp32 oracle."""
    inp = make_inputs(200, 96, seed=92, ragged=True)
    t = torch.tensor(0.6)
    with torch.inference_mode():
        ref1 = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
        ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 2, inp["z"], inp["c"], "euler", _cfg(cfg_params, 3.0, False))
    for dt in ("bf16", "f16"):
        out1 = decoders[dt].estimator(t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
        assert _rel(out1, ref1) <= NFE_TOL[dt], (dt, _rel(out1, ref1))
        out = _solve(decoders[dt], inp, 2, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
        assert torch.isfinite(out).all()
        assert _rel(out, ref) <= MEL_TOL and _disp(out, ref, inp["z"]) <= DISP_TOL[dt], (dt, _rel(out, ref), _disp(out, ref, inp["z"]))
        print(f"[{dt}] B=200 x T<=96: one evaluation {_rel(out1, ref1):.2e}; 2-step CFG solve displacement {_disp(out, ref, inp['z']):.2e}")


# ---------------------------------------------------------------- long ODE (BASELINE config 3)
def test_long_ode_50_steps_state_drift(decoders, sd, cfg_params):
    inp = make_inputs(1, 128, seed=3)
    kw = _cfg(cfg_params, 3.0, False)
    ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 50, inp["z"], inp["c"], "euler", kw)
    for dt in ("bf16", "f16"):
        out = _solve(decoders[dt], inp, 50, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
        assert _rel(out, ref) <= MEL_TOL
        assert _disp(out, ref, inp["z"]) <= DISP_TOL[dt], dt


@pytest.mark.parametrize("solver,n", [("euler", 50), ("rk4", 12)])
def test_config3_at_size_long_ode(decoders, sd, cfg_params, solver, n):
    """BASELINE config 3 at size: B=8 x T=1000 (ragged), CFG 3.0, n=50 Euler (100 evaluations) and rk4-3/8 n=12 (96
    evaluations) -- fp32-state drift over a long ODE with the production 256x256-tile kernels.  Utterances are
    independent, so two rows (the longest and a ragged one) are checked against the oracle run on those alone."""
    inp = make_inputs(8, 1000, seed=70, ragged=True)
    lens = inp["mask"][:, 0].sum(-1)
    rows = [int(lens.argmax()), int(lens.argmin())]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    ref = oracle.cfm_forward(sd, sub["mu"], sub["mask"], n, sub["z"], sub["c"], solver, _cfg(cfg_params, 3.0, False))
    for dt in ("bf16", "f16"):
        out = _solve(decoders[dt], inp, n, solver, _cfg(cfg_params, 3.0, True), inp["z"])
        assert torch.isfinite(out).all()
        pad = ~inp["mask"].bool().expand_as(out)
        assert torch.equal(out[pad], inp["z"][pad])
        got = out[rows]
        assert _rel(got, ref) <= MEL_TOL, dt
        assert _disp(got, ref, sub["z"]) <= DISP_TOL[dt], dt


def test_config3_at_the_benchmarked_batch(decoders, sd, cfg_params):
    """BASELINE config 3 exactly as `bench.py --n-timesteps 50` runs it -- B = 32 x T = 1000 (all-ones mask), CFG 3.0, n = 50 Euler, the
    default two-part solve on the big-grid kernels -- f16 operands: the first and the last utterance against the oracle run on those
    two alone (utterances are independent; 100 oracle evaluations of a 2-row batch).  Gates as for the B = 8 case."""
    inp = make_inputs(32, 1000, seed=0)
    rows = [0, 31]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    ref = oracle.cfm_forward(sd, sub["mu"], sub["mask"], 50, sub["z"], sub["c"], "euler", _cfg(cfg_params, 3.0, False))
    out = _solve(decoders["f16"], inp, 50, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
    assert torch.isfinite(out).all()
    got = out[rows]
    mel, disp = _rel(got, ref), _disp(got, ref, sub["z"])
    print(f"config 3 at B=32 x T=1000, n=50 euler, f16: mel {mel:.2e}, displacement {disp:.2e}")
    assert mel <= MEL_TOL and disp <= DISP_TOL["f16"]


# ---------------------------------------------------------------- full-size (BASELINE config 2) properties
@pytest.fixture(scope="module")
def c2(decoders, cfg_params):
    inp = make_inputs(32, 1000, seed=0, ragged=True)
    kw = _cfg(cfg_params, 3.0, True)
    out = _solve(decoders["f16"], inp, 10, "euler", kw, inp["z"])      # f16 = the shipping default
    return inp, out


def test_c2_padding_is_exactly_zero_and_finite(c2):
    inp, out = c2
    assert torch.isfinite(out).all()
    pad = ~inp["mask"].bool().expand_as(out)
    assert pad.any()
    # padded frames: the vector field is exactly 0 there (estimator.py:138) so the state stays z
    assert torch.equal(out[pad], inp["z"][pad])


def test_c2_deterministic_and_batch_permutation_equivariant(decoders, cfg_params, c2):
    inp, out = c2
    kw = _cfg(cfg_params, 3.0, True)
    again = _solve(decoders["f16"], inp, 10, "euler", kw, inp["z"])
    assert torch.equal(again, out)                                    # bitwise repeatable
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(0))
    pin = {k: v[perm] for k, v in inp.items()}
    pout = _solve(decoders["f16"], pin, 10, "euler", kw, pin["z"])
    assert torch.equal(pout, out[perm])                               # utterances are independent units


def test_c2_batch_rows_match_oracle(sd, cfg_params, c2):
    """The production configuration itself (B=32 x T=1000, 256x256 tiles, 10 Euler steps, CFG 3.0): utterances are
    independent units, so rows of the batched native solve are checked against the oracle run on those utterances
    alone (the longest one and a ragged one)."""
    inp, out = c2
    lens = inp["mask"][:, 0].sum(-1)
    rows = [int(lens.argmax()), int(lens.argmin())]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    ref = oracle.cfm_forward(sd, sub["mu"], sub["mask"], 10, sub["z"], sub["c"], "euler", _cfg(cfg_params, 3.0, False))
    got = out[rows]
    assert _rel(got, ref) <= MEL_TOL
    assert _disp(got, ref, sub["z"]) <= DISP_TOL["f16"]


def test_c2_all_ones_mask_rows_match_oracle(decoders, sd, cfg_params):
    """BASELINE config 2 exactly as benchmarked (B=32 x T=1000, ALL-ONES mask, n=10 Euler, CFG 3.0), both operand
    types: two rows against the oracle run on those utterances alone."""
    inp = make_inputs(32, 1000, seed=0)
    rows = [3, 29]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    ref = oracle.cfm_forward(sd, sub["mu"], sub["mask"], 10, sub["z"], sub["c"], "euler", _cfg(cfg_params, 3.0, False))
    for dt in ("bf16", "f16"):
        out = _solve(decoders[dt], inp, 10, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
        got = out[rows]
        assert _rel(got, ref) <= MEL_TOL, dt
        assert _disp(got, ref, sub["z"]) <= DISP_TOL[dt], dt


def test_mask_with_interior_zeros(decoders, sd):
    """Non-prefix masks (holes): the reference builds its attention mask from mask x mask^T and multiplies frames by
    the mask (diffusion_transformer.py:107-108), so any 0/1 pattern is legal.  One evaluation + exact zeros."""
    inp = make_inputs(3, 300, seed=61, lengths=[300, 300, 211])
    mask = inp["mask"].clone()
    mask[0, 0, 40:75] = 0; mask[0, 0, 128] = 0; mask[0, 0, 0] = 0            # holes incl. frame 0 and a tile boundary
    mask[1, 0, 64:128] = 0                                                    # one whole 64-key tile masked
    mask[2, 0, 0:70] = 0                                                      # FIRST key tile fully masked, valid keys later
    inp["mask"] = mask
    inp["mu"] = inp["mu"] * mask
    t = torch.tensor(0.45)
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd, t, inp["z"], mask, inp["mu"], inp["c"])
    for dt in ("bf16", "f16"):
        out = decoders[dt].estimator(t.cuda(), inp["z"].cuda(), mask.cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
        assert _rel(out, ref) <= NFE_TOL[dt], dt
        assert float(out[~mask.bool().expand_as(out)].abs().max()) == 0.0


def test_length_one_and_zero_items_inside_a_long_batch(decoders, sd, cfg_params):
    """A length-1 utterance and a fully padded (length-0) row inside a T=1000 batch: per-row kv_end / n_full
    bookkeeping, tiles without a valid key, exact zeros on padding."""
    inp = make_inputs(4, 1000, seed=62, lengths=[1000, 1, 517, 0])
    t = torch.tensor(0.8)
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
    for dt in ("bf16", "f16"):
        out = decoders[dt].estimator(t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
        assert torch.isfinite(out).all()
        assert _rel(out[:3], ref[:3]) <= NFE_TOL[dt], dt
        assert float(out[3].abs().max()) == 0.0 and float(out[1, :, 1:].abs().max()) == 0.0
    ref2 = oracle.cfm_forward(sd, inp["mu"][:3], inp["mask"][:3], 2, inp["z"][:3], inp["c"][:3], "euler", _cfg(cfg_params, 3.0, False))
    out2 = _solve(decoders["f16"], inp, 2, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
    assert _disp(out2[:3], ref2, inp["z"][:3]) <= DISP_TOL["f16"]
    assert torch.equal(out2[3], inp["z"][3])


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_strong_gates_ada_std_015(dt):
    """The parity weights draw the adaLN-Zero output layers N(0, 0.02): gates ~0.02-0.2, so attention / FFN errors
    enter the residual stream attenuated.  A trained checkpoint has gates of O(1).  This case uses std 0.15 (gates
    and scales ~1) so the attention and FFN branches carry full weight in the end-to-end gate (measured numbers are
    recorded in DESIGN.md: 6.1e-4 f16 / 5.1e-3 bf16 -- the same as with weak gates, i.e. the f16 configuration keeps
    meeting north_star's 1e-3 when every branch contributes un-attenuated)."""
    from stabletts_amd.flow_matching import CFMDecoder
    sd2 = oracle.make_state_dict(1234, ada_std=0.15)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
    dec.estimator.load_state_dict(sd2)
    dec = dec.cuda()
    inp = make_inputs(2, 300, seed=63, lengths=[300, 233])
    t = torch.tensor(0.3)
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd2, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
    out = dec.estimator(t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
    r = _rel(out, ref)
    print(f"ada_std=0.15 one-NFE rel err [{dt}]: {r:.3e}")
    assert r <= {"bf16": 1e-2, "f16": 1e-3}[dt]


def test_c2_items_match_oracle_at_full_length(decoders, sd, cfg_params):
    """Two utterances at T=1000 (ragged) with CFG, 2 Euler steps, against the oracle."""
    inp = make_inputs(2, 1000, seed=7, lengths=[1000, 731])
    ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 2, inp["z"], inp["c"], "euler", _cfg(cfg_params, 3.0, False))
    for dt in ("bf16", "f16"):
        out = _solve(decoders[dt], inp, 2, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
        assert _rel(out, ref) <= MEL_TOL
        assert _disp(out, ref, inp["z"]) <= DISP_TOL[dt]


def test_padded_batch_equals_unpadded_up_to_pad_leak(decoders, sd):
    """the native path reproduces the reference's pad-leak -- same utterance padded vs
    unpadded differs exactly as it does in the oracle (and not more)."""
    a = make_inputs(1, 96, seed=9)
    b = {k: v.clone() for k, v in a.items()}
    padn = 32
    for k in ("mu", "z"):
        b[k] = torch.nn.functional.pad(a[k], (0, padn))
    b["z"][..., 96:] = make_inputs(1, padn, seed=10)["z"]
    b["mask"] = torch.nn.functional.pad(a["mask"], (0, padn))
    t = torch.tensor(0.5)
    with torch.inference_mode():
        ref_b = oracle.decoder_forward(sd, t, b["z"], b["mask"], b["mu"], b["c"])
    out_b = decoders["f16"].estimator(t, b["z"].cuda(), b["mask"].cuda(), b["mu"].cuda(), b["c"].cuda()).cpu()
    assert _rel(out_b, ref_b) <= NFE_TOL["f16"]


@pytest.fixture(scope="module")
def dopri5_case(sd, cfg_params):
    """One oracle dopri5 solve (~430 CFG evaluations on the CPU), shared by both operand types."""
    inp = make_inputs(2, 33, seed=24, lengths=[33, 30])
    ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 10, inp["z"], inp["c"], "dopri5", _cfg(cfg_params, 2.0, False))
    return inp, ref


@pytest.mark.parametrize("dt,solver", [("bf16", None), ("f16", "dopri5")])
def test_adaptive_dopri5_vs_oracle(decoders, cfg_params, dopri5_case, dt, solver):
    """The reference default solver (solver=None -> torchdiffeq dopri5, rtol=atol=1e-5, flow_matching.py:54).
    Step acceptance is a discrete decision on a 16-bit-operand vector field, so native and oracle may take
    different steps; both must land within the solve tolerance of each other."""
    inp, ref = dopri5_case
    out = _solve(decoders[dt], inp, 10, solver, _cfg(cfg_params, 2.0, True), inp["z"])
    st = decoders[dt].estimator.engine().last_solve_stats()
    assert st["nfe"] >= 14 and st["steps"] >= 2 and st["nfe"] == 2 + 6 * st["steps"]
    assert torch.isfinite(out).all()
    # ~370-430 evaluations instead of 10-20: operand rounding accumulates, so the gates are wider than for the
    # 10-step solves (measured on MI355X: mel 8e-4 / 1e-4, displacement 1.8e-2 / 4.8e-3 for bf16 / f16; native and
    # oracle took the identical step sequence, 61 and 71 steps with 4 rejections each)
    assert _rel(out, ref) <= {"bf16": 2e-3, "f16": 1e-3}[dt]
    assert float((out - ref).abs().max() / (ref - inp["z"]).abs().max()) <= {"bf16": 3e-2, "f16": 8e-3}[dt]
    pad = ~inp["mask"].bool().expand_as(out)
    if pad.any():
        assert torch.equal(out[pad], inp["z"][pad])      # the field is exactly 0 on padded frames for every stage


@pytest.mark.parametrize("solver,stages,use_cfg", [("bosh3", 3, False), ("fehlberg2", 2, True), ("adaptive_heun", 1, False)])
def test_other_adaptive_solvers_vs_oracle(decoders, sd, cfg_params, solver, stages, use_cfg):
    """torchdiffeq's other explicit adaptive pairs offered by the reference's web UI (webui.py:110), native through
    the same controller as dopri5 (rtol = atol = 1e-5, flow_matching.py:54).  Tiny problem: the second-order pairs
    need ~1000 steps at that tolerance.  f16 operands; gates as for dopri5."""
    inp = make_inputs(1, 12, seed=31, lengths=[12])
    stats = {}
    kw = _cfg(cfg_params, 2.0, False) if use_cfg else None
    ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], 10, inp["z"], inp["c"], solver, kw)
    out = _solve(decoders["f16"], inp, 10, solver, _cfg(cfg_params, 2.0, True) if use_cfg else None, inp["z"])
    st = decoders["f16"].estimator.engine().last_solve_stats()
    assert st["steps"] >= 2 and st["nfe"] == 2 + stages * st["steps"]
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= 1e-3
    assert float((out - ref).abs().max() / (ref - inp["z"]).abs().max()) <= 8e-3


@pytest.mark.parametrize("use_cfg,n", [(False, 10), (True, 6)])
def test_implicit_adams_vs_oracle(decoders, sd, cfg_params, use_cfg, n):
    """torchdiffeq's 'implicit_adams' (the last of the methods webui.py:110 offers), native: Runge-Kutta start-up steps,
    Adams-Bashforth predictor, functional iteration of the Adams-Moulton corrector with the max-norm convergence test on the
    device.  vs oracle.odeint_implicit_adams (restated, parity unpinned) on the same inputs; f16 operands.  The corrector's
    iteration count is a discrete decision on a 16-bit-operand field: the evaluation counts may differ by a few."""
    inp = make_inputs(2, 40, seed=37, lengths=[40, 29])
    kw = _cfg(cfg_params, 2.0, False) if use_cfg else None
    stats = {}
    t_span = oracle.linspace_f32(n)
    if use_cfg:
        f = lambda t, x: oracle.cfg_wrapper(sd, t, x, inp["mask"], inp["mu"], inp["c"], kw["fake_speaker"], kw["fake_content"], kw["cfg_strength"])
    else:
        f = lambda t, x: oracle.decoder_forward(sd, t, x, inp["mask"], inp["mu"], inp["c"])
    with torch.inference_mode():
        ref = oracle.odeint_implicit_adams(f, inp["z"], t_span, stats=stats)
        assert torch.equal(ref, oracle.cfm_forward(sd, inp["mu"], inp["mask"], n, inp["z"], inp["c"], "implicit_adams", kw))
    out = _solve(decoders["f16"], inp, n, "implicit_adams", _cfg(cfg_params, 2.0, True) if use_cfg else None, inp["z"])
    st = decoders["f16"].estimator.engine().last_solve_stats()
    assert st["steps"] == n and abs(st["nfe"] - stats["nfe"]) <= n, (st, stats)
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= 1e-3
    assert float((out - ref).abs().max() / (ref - inp["z"]).abs().max()) <= 8e-3
    pad = ~inp["mask"].bool().expand_as(out)
    assert torch.equal(out[pad], inp["z"][pad])


def test_attention_rescale_branch_with_peaky_scores(sd):
    """Online-softmax rescale path: with q/k projections scaled 6x the row maxima keep growing across key tiles
    by more than the deferred-rescale threshold, so the (otherwise rare) rescale branch of attention.hip runs on
    most tiles.  f16 operands (score rounding error scales with |score|); vs the fp32 oracle with the same weights."""
    from stabletts_amd.flow_matching import CFMDecoder
    sd2 = {k: v.clone() for k, v in sd.items()}
    for i in range(6):
        for nm in ("q", "k"):
            sd2[f"blocks.{i}.block.attn.conv_{nm}.weight"] *= 6.0
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd2)
    dec = dec.cuda()
    inp = make_inputs(2, 300, seed=17, lengths=[300, 201])
    t = torch.tensor(0.6)
    taps = {}
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd2, t, inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
    # the scores really are peaky: log2-domain row maxima far above the deferral threshold of 6
    smax = (taps["b0.q"] @ taps["b0.k"].transpose(-1, -2)).amax(-1) * (math.log2(math.e) / 8.0)
    assert float(smax.median()) > 12.0
    out = dec.estimator(t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= 4e-3


@pytest.mark.parametrize("ada,qk,gate", [(0.15, 1.0, 1e-3), (0.02, 6.0, 2e-3), (0.15, 6.0, None)])
def test_attention_re_reference_path_at_size(ada, qk, gate):
    """Regression (round 4): the inference attention kernel re-references a row's softmax (O, l rescaled, tile redone) only when a
    lane's partial row sum leaves f16's comfortable range -- never with the seeded weights, on a few rows per wave with adaLN gates
    of O(1), on most tiles with 6x q / k projections.  That path returned row sums that missed their first term (an inline-asm
    v_add_f32 scheduled one instruction behind the v_exp_f32 producing its input: a transcendental-use hazard hipcc does not see
    inside asm), i.e. outputs scaled by up to 4x -- at B = 4 x T = 1000 ragged (16 key tiles), which the T = 300 peaky-score test
    did not reach.  One evaluation vs the fp32 oracle, and the attention output itself vs an fp64 softmax on the native q, k, v.
    (Both changes together make the random network chaotic -- rounding the fp32 oracle's own q, k, v to f16 moves its gradients by
    13 %, tests/test_gpu_training.py -- so that case gates the attention kernel and finiteness only; its end-to-end 0.14 is printed.)
    Measured on MI355X (round 5, profiles/r05_parity_trained.txt): one evaluation 7.8e-4 with O(1) gates (gate 1e-3 = north_star's bar),
    1.2e-3 with 6x q / k (gate 2e-3: there the f16 rounding of q and k ALONE moves the fp32 oracle by 9e-4,
    tools/qk_rounding_sensitivity.py -- the scores' row maxima are ~90-170, the softmax is an arg-max)."""
    from stabletts_amd.flow_matching import CFMDecoder
    B, T, lens = 4, 1000, [1000, 873, 655, 512]
    sd2 = oracle.make_state_dict(1234, ada_std=ada)
    for i in range(6):
        for nm in ("q", "k"):
            sd2[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd2[f"blocks.{i}.block.attn.conv_{nm}.weight"] * qk
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd2)
    dec = dec.cuda()
    inp = make_inputs(B, T, seed=81, lengths=lens)
    t = torch.tensor(0.5)
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd2, t, inp["z"], inp["mask"], inp["mu"], inp["c"])
    eng = dec.estimator.engine()
    eng.debug_capture(True)
    try:
        out = dec.estimator(t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
        torch.cuda.synchronize()
        H, Tp = 4, (T + 63) // 64 * 64
        tt = np.arange(Tp); pos = (tt & ~12) | ((tt & 4) << 1) | ((tt & 8) >> 1)
        q = eng.debug_fetch("b0.q").reshape(B, H, T, 64).astype(np.float64)          # scaled by log2(e) / 8
        k = eng.debug_fetch("b0.k").reshape(B, H, T, 64).astype(np.float64)
        v = eng.debug_fetch("b0.vt").reshape(B, H, 64, Tp)[..., pos][..., :T].transpose(0, 1, 3, 2).astype(np.float64)
        got = eng.debug_fetch("b0.attn").reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    finally:
        eng.debug_capture(False)
    m = inp["mask"][:, 0].double().numpy()
    worst = 0.0
    for b in range(B):
        S = q[b] @ k[b].transpose(0, 2, 1) + (1 - m[b])[None, None, :] * (-1e30)
        P = np.exp2(S - S.max(-1, keepdims=True)); P /= P.sum(-1, keepdims=True)
        want = P @ v[b]
        worst = max(worst, float(np.abs(got[b][:, :lens[b]] - want[:, :lens[b]]).max() / np.abs(want[:, :lens[b]]).max()))
    r = _rel(out, ref)
    print(f"ada_std {ada}, q/k x{qk}: attention (block 0) vs fp64 softmax on the native q, k, v {worst:.2e}; one evaluation vs the oracle {r:.2e}")
    assert torch.isfinite(out).all()
    assert worst <= 2e-3
    assert gate is None or r <= gate


def _trained_like(ada, qk):
    sd2 = oracle.make_state_dict(1234, ada_std=ada)
    for i in range(6):
        for nm in ("q", "k"):
            sd2[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd2[f"blocks.{i}.block.attn.conv_{nm}.weight"] * qk
    return sd2


@pytest.mark.parametrize("mode", ["default", "winograd"])
def test_default_engine_with_trained_like_weights_at_benchmark_size(cfg_params, monkeypatch, mode):
    """The shipped default on trial where the error budget is tightest (round-4 review, item 1): adaLN gates of O(1) -- what a
    released checkpoint has; the reference zero-initialises them only at init, models/estimator.py:98-101 -- at B = 32 x T = 1000
    RAGGED, n = 10 Euler, CFG 3.0, on an engine created with NO overrides and no capture: the fused big-grid FFN, the
    weight-stationary q/k/v and out-projection kernels, two solve parts.  Two rows (longest, shortest) against the fp32 oracle run
    on those utterances alone; gates = north_star's bar: ONE evaluation <= 1e-3 and solve displacement <= 1e-3.
    Measured (MI355X, profiles/r05_parity_trained.txt): default (direct fused FFN) 7.3e-4 / 4.1e-4; the opt-in Winograd FFN
    (ST_FUSED_FFN=3) 8.8e-4 / 3.7e-4 -- which is why it is opt-in: the numbers picked the default."""
    from stabletts_amd.flow_matching import CFMDecoder
    sd2 = _trained_like(0.15, 1.0)
    if mode == "winograd":
        monkeypatch.setenv("ST_FUSED_FFN", "3")
    else:
        monkeypatch.delenv("ST_FUSED_FFN", raising=False)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    assert dec.check_finite == (mode == "winograd")      # half the f16 range for the FFN intermediate: the guard is on by itself
    dec.estimator.load_state_dict(sd2)
    dec = dec.cuda(); dec.estimator.engine()
    monkeypatch.delenv("ST_FUSED_FFN", raising=False)
    inp = make_inputs(32, 1000, seed=0, ragged=True)
    lens = inp["mask"][:, 0].sum(-1)
    rows = [int(lens.argmax()), int(lens.argmin())]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    t = torch.tensor(0.5)
    with torch.inference_mode():
        ref1 = oracle.decoder_forward(sd2, t, sub["z"], sub["mask"], sub["mu"], sub["c"])
        ref = oracle.cfm_forward(sd2, sub["mu"], sub["mask"], 10, sub["z"], sub["c"], "euler", _cfg(cfg_params, 3.0, False))
    one = dec.estimator(t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()[rows]
    out = _solve(dec, inp, 10, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
    pad = ~inp["mask"].bool().expand_as(out)
    assert torch.isfinite(out).all() and torch.equal(out[pad], inp["z"][pad])
    e1, disp, mel = _rel(one, ref1), _disp(out[rows], ref, sub["z"]), _rel(out[rows], ref)
    print(f"trained-like (ada_std 0.15), B=32 x T=1000 ragged, {mode} engine: one evaluation {e1:.3e}, solve displacement {disp:.3e}, mel {mel:.3e}")
    assert e1 <= 1e-3 and disp <= 1e-3 and mel <= 1e-3


def test_trained_like_weights_with_peaky_attention_exceed_the_bar_and_why(cfg_params):
    """O(1) gates AND 3x q / k projections: the scores' row maxima are ~80-200 (tools/qk_rounding_sensitivity.py), the softmax is an
    arg-max, and rounding q, k, v to f16 -- nothing else -- already moves the fp32 ORACLE by 1.7e-3 per evaluation.  f16 attention
    operands cannot meet 1e-3 there whatever the kernels do (split q / k operands would: 3x the QK^T MFMAs); DESIGN.md section 2
    states it.  Regression guard on the default engine at benchmark size: one evaluation 3.9e-3, displacement 1.07e-3 measured."""
    from stabletts_amd.flow_matching import CFMDecoder
    sd2 = _trained_like(0.15, 3.0)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    dec.estimator.load_state_dict(sd2)
    dec = dec.cuda()
    inp = make_inputs(32, 1000, seed=0, ragged=True)
    lens = inp["mask"][:, 0].sum(-1)
    rows = [int(lens.argmax()), int(lens.argmin())]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    t = torch.tensor(0.5)
    with torch.inference_mode():
        ref1 = oracle.decoder_forward(sd2, t, sub["z"], sub["mask"], sub["mu"], sub["c"])
        ref = oracle.cfm_forward(sd2, sub["mu"], sub["mask"], 10, sub["z"], sub["c"], "euler", _cfg(cfg_params, 3.0, False))
    one = dec.estimator(t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()[rows]
    out = _solve(dec, inp, 10, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
    e1, disp = _rel(one, ref1), _disp(out[rows], ref, sub["z"])
    print(f"trained-like + 3x q/k, B=32 x T=1000 ragged, default engine: one evaluation {e1:.3e}, solve displacement {disp:.3e}")
    assert torch.isfinite(out).all() and e1 <= 8e-3 and disp <= 2.5e-3


def _oracle_max_lse(sd2, t, inp):
    """max over blocks, items, heads and VALID query rows of logsumexp_k(q k / sqrt(d) + mask) in the fp32 oracle (natural units)."""
    taps = {}
    with torch.inference_mode():
        oracle.decoder_forward(sd2, t, inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
    m = inp["mask"][:, 0].bool()
    best = -1e30
    for i in range(6):
        q, k = taps[f"b{i}.q"].double(), taps[f"b{i}.k"].double()
        s_ = q @ k.transpose(-1, -2) / 8.0
        s_ = s_.masked_fill(~m[:, None, None, :], -1e30)
        lse = torch.logsumexp(s_, dim=-1)                      # (B, H, T)
        best = max(best, float(lse[m[:, None, :].expand_as(lse)].max()))
    return best


@pytest.mark.parametrize("ada,qk", [(0.02, 1.0), (0.15, 1.0), (0.15, 3.0)])
def test_attention_statistic_matches_the_oracle(ada, qk):
    """st_attention_stats: the largest log-sum-exp of any valid attention row of the calls since the last query, against the same quantity
    of the fp32 oracle's q, k (a ragged batch: padded rows and masked keys must not enter).  It is what tells a serving loop that a
    checkpoint sits in the arg-max regime (values: ~10 at the seeded weights, 80-200 with O(1) gates and 3x q / k)."""
    from stabletts_amd.flow_matching import CFMDecoder
    sd2 = _trained_like(ada, qk) if (ada, qk) != (0.02, 1.0) else oracle.make_state_dict(1234)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    dec.estimator.load_state_dict(sd2)
    dec = dec.cuda()
    inp = make_inputs(3, 520, seed=17, lengths=[520, 401, 77])
    t = torch.tensor(0.5)
    eng = dec.estimator.engine()
    stream = torch.cuda.current_stream().cuda_stream
    assert eng.attention_stats(stream) == float("-inf")                 # nothing has run yet
    dec.estimator(t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    got = eng.attention_stats(stream)
    want = _oracle_max_lse(sd2, t, inp)
    print(f"ada_std {ada}, q/k x{qk}: max log-sum-exp native {got:.3f}, oracle {want:.3f}")
    assert abs(got - want) <= 0.02 * abs(want) + 0.05
    assert eng.attention_stats(stream) == float("-inf")                 # the query resets the statistic


@pytest.mark.parametrize("ada,qk", [(0.15, 1.0), (0.15, 3.0)])
def test_split_precision_attention_operands_at_benchmark_size(cfg_params, ada, qk):
    """attention_precision='split' (q and k as hi + lo operand pairs, scores from three products; st_set_option) measured where 16-bit
    q / k operands cost the most -- O(1) adaLN gates, and additionally 3x q / k projections (softmax = arg-max, score maxima 80-200) --
    on the engine as shipped at B = 32 x T = 1000 ragged, n = 10 Euler, CFG 3.0: one evaluation and the solve displacement with and
    without, against the fp32 oracle.  The split engine must not be worse anywhere; what it reaches in the arg-max regime is printed and
    recorded in DESIGN.md section 2 (the remaining error there is the other 16-bit operands, amplified by the saturated softmax)."""
    from stabletts_amd.flow_matching import CFMDecoder
    sd2 = _trained_like(ada, qk)
    inp = make_inputs(32, 1000, seed=0, ragged=True)
    lens = inp["mask"][:, 0].sum(-1)
    rows = [int(lens.argmax()), int(lens.argmin())]
    sub = {k: v[rows] for k, v in inp.items() if k != "lengths"}
    t = torch.tensor(0.5)
    with torch.inference_mode():
        ref1 = oracle.decoder_forward(sd2, t, sub["z"], sub["mask"], sub["mu"], sub["c"])
        ref = oracle.cfm_forward(sd2, sub["mu"], sub["mask"], 10, sub["z"], sub["c"], "euler", _cfg(cfg_params, 3.0, False))
    res = {}
    for mode in ("16bit", "split"):
        dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, attention_precision=mode)
        dec.estimator.load_state_dict(sd2)
        dec = dec.cuda()
        eng = dec.estimator.engine()
        assert eng.get_option("attention_precision") == (1 if mode == "split" else 0)
        one = dec.estimator(t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()[rows]
        lse = eng.attention_stats(torch.cuda.current_stream().cuda_stream)
        out = _solve(dec, inp, 10, "euler", _cfg(cfg_params, 3.0, True), inp["z"])
        pad = ~inp["mask"].bool().expand_as(out)
        assert torch.isfinite(out).all() and torch.equal(out[pad], inp["z"][pad])
        res[mode] = (_rel(one, ref1), _disp(out[rows], ref, sub["z"]), lse)
        del dec
    print(f"ada_std {ada}, q/k x{qk}, B=32 x T=1000 ragged: one evaluation / displacement  16-bit q,k {res['16bit'][0]:.3e} / {res['16bit'][1]:.3e}   "
          f"split q,k {res['split'][0]:.3e} / {res['split'][1]:.3e}   (max log-sum-exp {res['16bit'][2]:.1f})")
    assert res["split"][0] <= 1.05 * res["16bit"][0] and res["split"][1] <= 1.05 * res["16bit"][1]
    if qk == 1.0:
        assert res["split"][0] <= 1e-3 and res["split"][1] <= 1e-3


def test_attention_precision_auto_switches_in_the_arg_max_regime(cfg_params):
    """CFMDecoder(attention_precision='auto'): the first solve runs with 16-bit q / k operands, the engine's statistic (max log-sum-exp
    > Decoder.AUTO_SPLIT_LSE) says the softmax is an arg-max, the decoder warns, switches its engine to split operands and repeats the
    solve -- the result is bit-identical to a decoder constructed with attention_precision='split'; with the seeded weights it stays
    on the 16-bit kernel and equals the default decoder bit for bit."""
    import warnings
    from stabletts_amd.flow_matching import CFMDecoder
    inp = make_inputs(4, 300, seed=23, lengths=[300, 255, 190, 64])
    kw = _cfg(cfg_params, 3.0, True)
    for sd2, expect_split in ((_trained_like(0.15, 3.0), True), (oracle.make_state_dict(1234), False)):
        outs = {}
        for mode in ("auto", "split", "16bit"):
            dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, attention_precision=mode)
            dec.estimator.load_state_dict(sd2)
            dec = dec.cuda()
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                outs[mode] = _solve(dec, inp, 3, "euler", kw, inp["z"])
                outs[mode + "2"] = _solve(dec, inp, 3, "euler", kw, inp["z"])
            if mode == "auto":
                assert (len([x for x in w if "arg-max" in str(x.message)]) == 1) == expect_split
                assert dec.estimator.engine().get_option("attention_precision") == int(expect_split)
        assert torch.equal(outs["auto"], outs["split" if expect_split else "16bit"])
        assert torch.equal(outs["auto2"], outs["auto"])
        assert not torch.equal(outs["split"], outs["16bit"])


def test_ffn_intermediate_in_the_upper_half_of_f16_range(sd, monkeypatch):
    """|u| in (32,752, 65,504): inside f16's range, so the DEFAULT engine (direct fused FFN) must stay exact -- and it is what runs
    unless the caller opts into the Winograd kernel, whose transformed operands are sums of two rows and overflow there; that
    engine must say so by itself (check_finite defaults to on with ST_FUSED_FFN=3), not return NaN silently.  conv_1 of block 0 is
    scaled up (conv_2 down by the same factor, so the residual stream keeps its magnitude)."""
    from stabletts_amd.flow_matching import CFMDecoder
    inp = make_inputs(2, 504, seed=91, lengths=[504, 377])
    t = torch.tensor(0.4)
    taps = {}
    with torch.inference_mode():
        oracle.decoder_forward(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
    u = taps["b0.u"]                                   # (B, F, T), masked
    alpha = 48000.0 / float(u.max())             # (SiLU is sub-linear near 0: the scaled maximum lands a little higher)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["blocks.0.block.mlp.conv_1.weight"] *= alpha; sd2["blocks.0.block.mlp.conv_1.bias"] *= alpha
    sd2["blocks.0.block.mlp.conv_2.weight"] /= alpha
    taps = {}
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd2, t, inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
    u = taps["b0.u"]
    pair = u[..., 1:] + u[..., :-1]
    assert 40000 < float(u.abs().max()) < 60000
    assert float(pair[..., 0::2].max()) > 70000 and float(pair[..., 1::2].max()) > 70000      # a sum of two rows overflows, either pairing
    args = (t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    for mode in ("default", "3"):
        monkeypatch.setenv("ST_BIG_MIN_BLOCKS", "1"); monkeypatch.setenv("ST_SMALL_GRID", "0")
        if mode == "3":
            monkeypatch.setenv("ST_FUSED_FFN", "3")
        dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
        dec.estimator.load_state_dict(sd2)
        dec = dec.cuda(); dec.estimator.engine()
        for k in ("ST_BIG_MIN_BLOCKS", "ST_SMALL_GRID", "ST_FUSED_FFN"):
            monkeypatch.delenv(k, raising=False)
        if mode == "default":
            out = dec.estimator(*args).cpu()
            r = _rel(out, ref)
            print(f"max |u| {float(u.abs().max()):.0f}: default engine one evaluation vs the oracle {r:.2e}")
            assert torch.isfinite(out).all() and r <= NFE_TOL["f16"]
            z2 = dec(inp["mu"].cuda(), inp["mask"].cuda(), 2, 1.0, inp["c"].cuda(), "euler", None, z=inp["z"].cuda())
            assert torch.isfinite(z2).all() and not dec.check_finite
        else:
            assert dec.check_finite
            with pytest.raises(FloatingPointError):
                dec(inp["mu"].cuda(), inp["mask"].cuda(), 2, 1.0, inp["c"].cuda(), "euler", None, z=inp["z"].cuda())



def test_cfg_strength_one_equals_cond_branch(decoders, cfg_params):
    inp = make_inputs(2, 80, seed=4, lengths=[80, 61])
    a = _solve(decoders["f16"], inp, 3, "euler", _cfg(cfg_params, 1.0, True), inp["z"])
    b = _solve(decoders["f16"], inp, 3, "euler", None, inp["z"])
    assert _rel(a, b) <= 1e-5          # u + 1*(c-u) == c up to fp32 rounding


def test_temperature_scales_noise(decoders):
    inp = make_inputs(1, 40, seed=6)
    d = decoders["f16"]
    a = d(inp["mu"].cuda(), inp["mask"].cuda(), 2, 0.5, inp["c"].cuda(), "euler", None, z=inp["z"].cuda())
    b = d(inp["mu"].cuda(), inp["mask"].cuda(), 2, 1.0, inp["c"].cuda(), "euler", None, z=(inp["z"] * 0.5).cuda())
    assert torch.equal(a, b)
    c = d(inp["mu"].cuda(), inp["mask"].cuda(), 2, 1.0, inp["c"].cuda(), "euler")      # internal noise
    assert c.shape == a.shape and torch.isfinite(c).all()


def test_n_mels_80_variant(cfg_params):
    """north_star quotes mel=80; the reference default is 128. Channel padding path (n_feats < 128)."""
    from stabletts_amd.flow_matching import CFMDecoder
    cfg80 = oracle.DecoderConfig(noise_channels=80, cond_channels=80, out_channels=80)
    sd80 = oracle.make_state_dict(99, cfg80)
    dec = CFMDecoder(80, 80, 256, 80, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd80)
    dec = dec.cuda()
    inp = make_inputs(2, 50, seed=8, lengths=[50, 31], n_feats=80)
    ref = oracle.cfm_forward(sd80, inp["mu"], inp["mask"], 3, inp["z"], inp["c"], "euler", None)
    out = _solve(dec, inp, 3, "euler", None, inp["z"])
    assert _rel(out, ref) <= MEL_TOL


def test_parameter_update_is_picked_up(sd):
    from stabletts_amd.flow_matching import CFMDecoder
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    inp = make_inputs(1, 32, seed=2)
    t = torch.tensor(0.2)
    args = (t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    a = dec.estimator(*args)
    with torch.no_grad():
        dec.estimator.final_proj.bias.add_(1.0)
    b = dec.estimator(*args)
    assert float((b - a).mean()) == pytest.approx(1.0, abs=1e-3)


def test_compute_loss_forward_vs_reference_fixture(decoders, golden):
    """CFMDecoder.compute_loss forward (flow_matching.py:69-100) against the REAL reference's value, same draws."""
    inp = make_inputs(2, 44, seed=31, lengths=[44, 29])
    x1 = make_inputs(2, 44, seed=32)["z"]
    want = float(golden["loss_value"][0])
    for dt, tol in (("bf16", 3e-3), ("f16", 5e-4)):
        with torch.no_grad():
            loss, y = decoders[dt].compute_loss(x1.cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda(),
                                                t_rand=torch.from_numpy(golden["loss_t_rand"]).cuda(),
                                                z=torch.from_numpy(golden["loss_z"]).cuda())
        assert abs(float(loss) - want) <= tol * want, (dt, float(loss), want)
        assert _rel(y.cpu(), torch.from_numpy(golden["loss_y"])) <= 1e-6


def test_error_behaviour(decoders):
    d = decoders["bf16"]
    inp = make_inputs(2, 16, seed=1)
    with pytest.raises(ValueError):
        d(inp["mu"].cuda(), inp["mask"][:1].cuda(), 2, 1.0, inp["c"].cuda(), "euler")
    with pytest.raises(NotImplementedError):
        d(inp["mu"].cuda(), inp["mask"].cuda(), 2, 1.0, inp["c"].cuda(), "explicit_adams")
    from stabletts_amd._lib import NativeError
    with pytest.raises(NativeError):
        d(inp["mu"].cuda(), inp["mask"].cuda(), 0, 1.0, inp["c"].cuda(), "euler")
    # maximum size: 2*B*T*filter must stay below 2^31 (32-bit row indexing); rejected before anything is allocated
    eng = d.estimator.engine()
    fake = torch.zeros(1, device="cuda")
    with pytest.raises(NativeError, match="too large"):
        eng.lib.st_cfm_solve.restype  # noqa: B018  (binding exists)
        eng._check(eng.lib.st_cfm_solve(eng.handle, fake.data_ptr(), fake.data_ptr(), fake.data_ptr(), fake.data_ptr(),
                                        2, 0, 0, 0.0, None, None, fake.data_ptr(), 1024, 2048, None))


def test_non_native_solver_runs_torchdiffeq_controller_over_native_estimator(decoders, cfg_params, monkeypatch):
    """solver names without a native controller (torchdiffeq methods outside webui.py:110's list, e.g. explicit_adams) hand the time stepping to
    torchdiffeq while every vector-field evaluation stays native.  torchdiffeq is not installed here, so a
    stand-in module whose ``odeint`` is the fixed-grid Euler rule checks the plumbing (call signature of
    flow_matching.py:54, CFG wrapper, trajectory[-1]): the result must equal the fused native euler solve."""
    import sys
    import types
    calls = {}

    def odeint(fn, y0, t, method=None, rtol=None, atol=None):
        calls.update(method=method, rtol=rtol, atol=atol, nfe=0)
        ys, y = [y0], y0
        for i in range(len(t) - 1):
            y = y + (t[i + 1] - t[i]) * fn(t[i], y)
            calls["nfe"] += 1
            ys.append(y)
        return torch.stack(ys)

    fake = types.ModuleType("torchdiffeq")
    fake.odeint = odeint
    monkeypatch.setitem(sys.modules, "torchdiffeq", fake)
    d = decoders["f16"]
    fs, fc = cfg_params
    kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=2.0)
    inp = {k: v.cuda() for k, v in make_inputs(2, 60, seed=41, lengths=[60, 37]).items() if k != "lengths"}
    for cfg in (None, kw):
        ref = d(inp["mu"], inp["mask"], 4, 0.8, inp["c"], "euler", cfg, z=inp["z"])
        out = d(inp["mu"], inp["mask"], 4, 0.8, inp["c"], "explicit_adams", cfg, z=inp["z"])
        assert calls == dict(method="explicit_adams", rtol=1e-5, atol=1e-5, nfe=4)
        assert _rel(out.cpu(), ref.cpu()) <= 2e-4


@pytest.mark.parametrize("solver,n", [("euler", 6), ("rk4", 2)])
def test_hip_graph_replay_is_bitwise_identical(decoders, cfg_params, monkeypatch, solver, n):
    """ST_HIP_GRAPH=1: the fixed-grid solve body is captured into a HIP graph the second time a solve signature is
    seen and replayed afterwards.  Replays must equal the eager launches bit for bit, also when the caller's input
    tensors (content and addresses) change between calls."""
    d = decoders["bf16"]
    kw = _cfg(cfg_params, 2.5, True)
    cases = [make_inputs(3, 90, seed=50 + i, lengths=[90, 64, 33]) for i in range(4)]
    monkeypatch.delenv("ST_HIP_GRAPH", raising=False)
    eager = [_solve(d, c, n, solver, kw, c["z"]) for c in cases]
    monkeypatch.setenv("ST_HIP_GRAPH", "1")
    for rep in range(2):
        for c, ref in zip(cases, eager):          # call 1 eager (warm), call 2 captures, later calls replay
            out = _solve(d, c, n, solver, kw, c["z"])
            assert torch.equal(out, ref)
