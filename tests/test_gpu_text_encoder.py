"""GPU parity of the native TextEncoder through the C ABI (st_text_encoder_forward):
against the outputs of the REAL reference module (tests/golden/text_encoder_outputs.npz, written by
oracle/make_golden_text_encoder.py) and against the fp32 oracle on other seeded inputs.
Tolerances: relative to the tensor's max magnitude; x is the fp32 residual stream fed by 16-bit-operand GEMMs,
mu_x one more 16-bit-operand GEMM on top (same bars as the decoder's stage tests)."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle.make_golden_text_encoder import CASES, text_inputs

pytestmark = pytest.mark.gpu

# measured on MI355X (round 5, proj on split-precision operands): f16 x 1.1e-4 / mu_x 5.9e-5, bf16 1.1e-3 / 5.8e-4
TOL_X = {"bf16": 3e-3, "f16": 3e-4}
TOL_MU = {"bf16": 2e-3, "f16": 3e-4}


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


@pytest.fixture(scope="module")
def enc_sd():
    return oracle.make_text_encoder_state_dict(2468)


@pytest.fixture(scope="module")
def encoders(enc_sd):
    from stabletts_amd.text_encoder import TextEncoder
    out = {}
    for dt in ("bf16", "f16"):
        m = TextEncoder(401, 128, 256, 1024, 4, 3, 3, 0.1, 256, operand_dtype=dt)
        m.load_state_dict(enc_sd)
        out[dt] = m.cuda()
    return out


def test_state_dict_layout_matches_reference(enc_sd, encoders):
    sd = encoders["bf16"].state_dict()
    assert set(sd) == set(enc_sd)
    assert all(tuple(sd[k].shape) == tuple(enc_sd[k].shape) for k in sd)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", list(CASES))
def test_vs_reference_fixture(encoders, dt, case):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "text_encoder_outputs.npz"))
    B, T, lengths, seed = CASES[case]
    tok, c, lens = text_inputs(B, T, lengths, seed)
    x, mu_x, mask = encoders[dt](tok.cuda(), c.cuda(), lens.cuda())
    assert np.array_equal(mask.cpu().numpy(), g[case + "_mask"])
    assert _rel(x.cpu().numpy(), g[case + "_x"]) <= TOL_X[dt]
    ex, em = _rel(x.cpu().numpy(), g[case + "_x"]), _rel(mu_x.cpu().numpy(), g[case + "_mu_x"])
    print(f"text encoder {dt} {case}: x {ex:.2e} mu_x {em:.2e}")
    assert em <= TOL_MU[dt]
    pad = ~mask.bool().expand_as(mu_x)
    assert float(mu_x[pad].abs().max()) == 0.0          # proj(x) * x_mask (text_encoder.py:42)
    assert float(x[~mask.bool().expand_as(x)].abs().max()) == 0.0


def test_vs_oracle_other_inputs_and_determinism(encoders, enc_sd):
    tok, c, lens = text_inputs(4, 250, [250, 180, 97, 1], 77)
    with torch.inference_mode():
        rx, rmu, rmask = oracle.text_encoder_forward(enc_sd, tok, c, lens)
    for dt in ("bf16", "f16"):
        x, mu_x, mask = encoders[dt](tok.cuda(), c.cuda(), lens.cuda())
        assert torch.equal(mask.cpu(), rmask)
        assert _rel(x.cpu().numpy(), rx.numpy()) <= TOL_X[dt]
        assert _rel(mu_x.cpu().numpy(), rmu.numpy()) <= TOL_MU[dt]
        x2, mu2, _ = encoders[dt](tok.cuda(), c.cuda(), lens.cuda())
        assert torch.equal(x2, x) and torch.equal(mu2, mu_x)      # bitwise repeatable


def test_handle_kinds_are_not_interchangeable(encoders):
    from stabletts_amd._lib import NativeError
    eng = encoders["bf16"].engine()
    z = torch.zeros(1, 128, 8, device="cuda")
    with pytest.raises(NativeError):
        eng.cfm_solve(z, torch.ones(1, 1, 8, device="cuda"), z, torch.zeros(1, 256, device="cuda"), 2, 0, False, 0.0,
                      None, None, torch.empty_like(z), torch.cuda.current_stream().cuda_stream)
    with pytest.raises(ValueError):
        encoders["bf16"](torch.zeros(2, 5, dtype=torch.long), torch.zeros(1, 256), torch.tensor([5, 5]))
