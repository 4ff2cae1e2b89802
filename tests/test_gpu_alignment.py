"""Native duration -> alignment -> mu_y (stabletts_amd.alignment, st_durations / st_align / st_generate_path) against
fixtures produced by the REAL reference helpers (models/model.py:17-27,85-95): bit-exact.  Run with ``-m gpu``."""
import os

import numpy as np
import pytest
import torch

from oracle.make_golden_align import CASES, align_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "align_outputs.npz"))


@pytest.mark.parametrize("name", list(CASES))
def test_length_regulate_bit_exact(gold, name):
    from stabletts_amd.alignment import length_regulate
    B, Tx, xl, M, ls, seed = CASES[name]
    logw, mu_x, x_mask, _ = align_inputs(B, Tx, xl, M, seed)
    r = length_regulate(logw.cuda(), x_mask.cuda(), mu_x.cuda(), ls)
    assert np.array_equal(r["w_ceil"].cpu().numpy(), gold[name + "_w_ceil"])
    assert np.array_equal(r["y_lengths"].cpu().numpy(), gold[name + "_y_lengths"])
    assert np.array_equal(r["y_mask"].cpu().numpy(), gold[name + "_y_mask"])
    assert np.array_equal(r["attn"][:, 0].cpu().numpy().astype(np.uint8), gold[name + "_attn"])
    assert np.array_equal(r["mu_y"].cpu().numpy(), gold[name + "_mu_y"])       # a gather of fp32 values: exact
    r2 = length_regulate(logw.cuda(), x_mask.cuda(), mu_x.cuda(), ls, return_attn=False)
    assert r2["attn"] is None and torch.equal(r2["mu_y"], r["mu_y"])


@pytest.mark.parametrize("name", list(CASES))
def test_generate_path_drop_in_bit_exact(gold, name):
    """generate_path(duration, mask) with the reference's own arguments (w_ceil, x_mask x y_mask)."""
    from stabletts_amd.alignment import generate_path
    B, Tx, xl, M, ls, seed = CASES[name]
    _, _, x_mask, _ = align_inputs(B, Tx, xl, M, seed)
    w_ceil = torch.from_numpy(gold[name + "_w_ceil"])[:, 0]
    y_mask = torch.from_numpy(gold[name + "_y_mask"])
    attn_mask = (x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1)
    path = generate_path(w_ceil.cuda(), attn_mask.cuda())
    assert path.dtype == attn_mask.dtype
    assert np.array_equal(path.cpu().numpy().astype(np.uint8), gold[name + "_attn"])


def test_alignment_feeds_the_native_decoder(sd, cfg_params):
    """TextEncoder-shaped mu_x -> native length regulation -> native CFM solve: the (mu_y, y_mask) pair produced
    natively drives CFMDecoder.forward exactly like the reference-produced pair."""
    from stabletts_amd.alignment import length_regulate
    from stabletts_amd.flow_matching import CFMDecoder
    B, Tx, xl, M, ls, seed = CASES["basic"]
    logw, mu_x, x_mask, _ = align_inputs(B, Tx, xl, M, seed)
    r = length_regulate(logw.cuda(), x_mask.cuda(), mu_x.cuda(), ls)
    g = np.load(os.path.join(ROOT, "tests", "golden", "align_outputs.npz"))
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    c = torch.randn(B, 256, generator=torch.Generator().manual_seed(0)).cuda()
    z = torch.randn(r["mu_y"].shape, generator=torch.Generator().manual_seed(1)).cuda()
    a = dec(r["mu_y"], r["y_mask"], 3, 1.0, c, "euler", None, z=z)
    b = dec(torch.from_numpy(g["basic_mu_y"]).cuda(), torch.from_numpy(g["basic_y_mask"]).cuda(), 3, 1.0, c, "euler", None, z=z)
    assert torch.equal(a, b)
