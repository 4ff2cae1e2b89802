"""GPU parity of the native Vocos vocoder against the fixtures of the REAL reference module
(tests/golden/vocos_outputs.npz) and the numpy oracle."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import vocos_oracle as vo
from oracle.make_golden_vocos import CASES, SD_SEED

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vocos_outputs.npz")
# max |audio - ref| / max |ref|; hidden: max |h - ref| / max |ref| of the final LayerNorm output
# f16 measured 6.6e-4 .. 7.3e-4 with [W_hi | W_lo] pointwise weights (round 6, profiles/r06_vocos_split_weights.txt; 6.9e-4 .. 9.7e-4
# with only the head split, round 5; 8.8e-4 .. 1.13e-3 before): the 1e-3 bar
TOL_AUDIO = {"f16": 1e-3, "bf16": 1e-2}
TOL_HIDDEN = {"f16": 1.5e-3, "bf16": 8e-3}


def _cfgs():
    c = vo.VocosConfig
    return (types.SimpleNamespace(input_channels=c.input_channels, dim=c.dim, intermediate_dim=c.intermediate_dim, num_layers=c.num_layers),
            types.SimpleNamespace(n_fft=c.n_fft, hop_length=c.hop_length))


@pytest.fixture(scope="module", params=["f16", "bf16"])
def voc(request):
    from stabletts_amd.vocos import Vocos
    m = Vocos(*_cfgs(), operand_dtype=request.param)
    sd = {k: torch.from_numpy(v) for k, v in vo.make_vocos_state_dict(SD_SEED).items()}
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0")


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("name", list(CASES))
def test_vocos_matches_reference_fixtures(voc, name):
    gold = np.load(GOLD)
    B, T, seed = CASES[name]
    mel = torch.from_numpy(vo.make_mel(B, T, seed)).cuda()
    eng = voc.engine()
    eng.debug_capture(True)
    audio = voc(mel).cpu().numpy()
    hid = eng.debug_fetch("voc.hidden").reshape(B, T, -1)
    eng.debug_capture(False)
    assert audio.shape == gold[name + ".audio"].shape
    eh, ea = _rel(hid, gold[name + ".hidden"]), _rel(audio, gold[name + ".audio"])
    print(f"{voc.operand_dtype} {name}: hidden {eh:.2e} audio {ea:.2e}")
    assert eh < TOL_HIDDEN[voc.operand_dtype]
    assert ea < TOL_AUDIO[voc.operand_dtype]


def test_istft_head_alone_is_fp32_exact(voc):
    """The ISTFT kernels (spectrum, radix-4 inverse FFT, overlap-add, envelope) are fp32: fed the oracle's own head
    projection they must reproduce the oracle's waveform to fp32 accuracy.  Checked through the captured head output:
    audio == istft(head_out) computed by the oracle."""
    B, T, seed = 2, 61, 11
    mel = torch.from_numpy(vo.make_mel(B, T, seed)).cuda()
    eng = voc.engine()
    eng.debug_capture(True)
    audio = voc(mel).cpu().numpy()
    ho = eng.debug_fetch("voc.head_out").reshape(B, T, 2, -1)[..., :1025].astype(np.float64)     # planes -> 1025 bins
    eng.debug_capture(False)
    mag = np.minimum(np.exp(ho[:, :, 0]), 1e2)
    S = (mag * (np.cos(ho[:, :, 1]) + 1j * np.sin(ho[:, :, 1]))).transpose(0, 2, 1)
    sd = vo.make_vocos_state_dict(SD_SEED)
    ref = vo.istft_same(S, sd["head.istft.window"].astype(np.float64), 2048, 512)
    assert _rel(audio, ref) < 2e-5


def test_vocos_long_batch_vs_oracle(voc):
    """A batch at bench-like length against the fp64 oracle (row-flattened GEMMs: utterance boundaries must not leak
    through the depthwise convolution or the overlap-add)."""
    B, T, seed = 3, 300, 5
    mel_np = vo.make_mel(B, T, seed)
    sd = vo.make_vocos_state_dict(SD_SEED)
    ref = vo.vocos_forward(sd, mel_np)
    audio = voc(torch.from_numpy(mel_np).cuda()).cpu().numpy()
    e = _rel(audio, ref)
    print(f"{voc.operand_dtype} B=3 T=300: audio {e:.2e}")
    assert e < TOL_AUDIO[voc.operand_dtype]
    # utterances are independent: item 1 alone gives the same waveform
    solo = voc(torch.from_numpy(mel_np[1:2]).cuda()).cpu().numpy()
    assert np.array_equal(solo[0], audio[1]) or _rel(solo[0], audio[1]) < 1e-6


def test_vocos_rejects_wrong_inputs(voc):
    with pytest.raises(ValueError):
        voc(torch.zeros(1, 128, 8))                       # CPU tensor
    with pytest.raises(ValueError):
        voc(torch.zeros(1, 80, 8, device="cuda"))         # wrong n_mels


def test_vocoder_is_inference_only_and_says_so():
    """the native vocoder has no backward.  Its parameters do not require grad (plain ``voc(mel)`` works in
    any grad mode and returns a graph-less tensor); asking for gradients -- a mel that requires grad, or parameters
    switched to requires_grad -- raises instead of silently training nothing."""
    import types
    from stabletts_amd.vocos import Vocos
    c = vo.VocosConfig
    voc = Vocos(types.SimpleNamespace(input_channels=c.input_channels, dim=c.dim, intermediate_dim=c.intermediate_dim, num_layers=c.num_layers),
                types.SimpleNamespace(n_fft=c.n_fft, hop_length=c.hop_length))
    voc.load_state_dict({k: torch.from_numpy(v) for k, v in vo.make_vocos_state_dict(77).items()})
    voc = voc.cuda()
    assert not any(p.requires_grad for p in voc.parameters())
    mel = torch.randn(1, c.input_channels, 20, device="cuda")
    with torch.enable_grad():
        out = voc(mel)
        assert not out.requires_grad and torch.isfinite(out).all()
        with pytest.raises(NotImplementedError, match="inference-only"):
            voc(mel.clone().requires_grad_(True))
        voc.backbone.embed.weight.requires_grad_(True)
        with pytest.raises(NotImplementedError, match="inference-only"):
            voc(mel)
