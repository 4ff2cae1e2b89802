"""CPU restatement (numpy) of the reference's Vocos vocoder forward: mel -> waveform (test infrastructure).

Follows vocoders/vocos/models/backbone.py:50-56 (VocosBackbone.forward), module.py:33-46 (ConvNeXtBlock.forward),
head.py:39-72 (ISTFT.forward, padding="same") and head.py:93-117 (ISTFTHead.forward), with config.py:4-19 (MelConfig:
n_fft 2048, hop 512) and config.py:46-50 (VocosConfig: 128 -> 512, intermediate 1536, 8 layers).
Pinned against outputs of the REAL modules by oracle/make_golden_vocos.py -> tests/golden/vocos_outputs.npz
(tests/test_oracle_golden.py).  NOT part of the product: stabletts_amd/ never imports this.
"""
import numpy as np
from scipy.special import erf


class VocosConfig:
    """config.py:46-50 + the two MelConfig fields the head uses (config.py:6,8)."""
    input_channels = 128
    dim = 512
    intermediate_dim = 1536
    num_layers = 8
    n_fft = 2048
    hop_length = 512


def make_vocos_state_dict(seed, cfg=VocosConfig, dtype=np.float32):
    """Seeded, non-degenerate stand-in for a trained checkpoint, keyed like ``Vocos.state_dict()`` (model.py:11-15):
    every LayerNorm has non-trivial affine parameters, layer scales vary per channel, biases are non-zero."""
    rng = np.random.Generator(np.random.PCG64(seed))
    C, F, L, M, N = cfg.dim, cfg.intermediate_dim, cfg.num_layers, cfg.input_channels, cfg.n_fft
    n = lambda *s, std=1.0: (rng.standard_normal(s) * std).astype(dtype)
    sd = {}
    sd["backbone.embed.weight"] = n(C, M, 7, std=1.0 / np.sqrt(7 * M))
    sd["backbone.embed.bias"] = n(C, std=0.1)
    sd["backbone.norm.weight"] = (1.0 + n(C, std=0.1)).astype(dtype)
    sd["backbone.norm.bias"] = n(C, std=0.1)
    for i in range(L):
        p = f"backbone.convnext.{i}."
        sd[p + "dwconv.weight"] = n(C, 1, 7, std=1.0 / np.sqrt(7))
        sd[p + "dwconv.bias"] = n(C, std=0.1)
        sd[p + "norm.weight"] = (1.0 + n(C, std=0.1)).astype(dtype)
        sd[p + "norm.bias"] = n(C, std=0.1)
        sd[p + "pwconv1.weight"] = n(F, C, std=1.0 / np.sqrt(C))
        sd[p + "pwconv1.bias"] = n(F, std=0.1)
        sd[p + "pwconv2.weight"] = n(C, F, std=1.0 / np.sqrt(F))
        sd[p + "pwconv2.bias"] = n(C, std=0.1)
        sd[p + "gamma"] = ((1.0 / L) * (1.0 + n(C, std=0.3))).astype(dtype) * 4.0
    sd["backbone.final_layer_norm.weight"] = (1.0 + n(C, std=0.1)).astype(dtype)
    sd["backbone.final_layer_norm.bias"] = n(C, std=0.1)
    sd["head.out.weight"] = n(N + 2, C, std=1.0 / np.sqrt(C))
    sd["head.out.bias"] = n(N + 2, std=0.3)
    k = np.arange(N, dtype=np.float64)
    sd["head.istft.window"] = (0.5 - 0.5 * np.cos(2.0 * np.pi * k / N)).astype(dtype)   # torch.hann_window (periodic), head.py:28
    return sd


def make_mel(B, T, seed, M=128):
    """log-mel-like input (B, M, T): values in roughly [-11, 2] like utils/audio.py's log(clamp(mel, 1e-5))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.clip(rng.standard_normal((B, M, T)) * 2.0 - 4.0, -11.5, 3.0).astype(np.float32)


def conv1d_same(x, w, b, groups=1):
    """nn.Conv1d(k, padding=k//2[, groups]) on (B, C, T)."""
    B, Cin, T = x.shape
    Cout, Cg, K = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (K // 2, K // 2)))
    cols = np.stack([xp[:, :, j:j + T] for j in range(K)], axis=-1)          # (B, Cin, T, K)
    if groups == 1:
        y = np.einsum("bctk,ock->bot", cols, w)
    else:
        assert groups == Cin == Cout and Cg == 1
        y = np.einsum("bctk,ck->bct", cols, w[:, 0])
    return y + b[None, :, None]


def layer_norm(x, w, b, eps=1e-6):
    """nn.LayerNorm over the LAST axis (biased variance)."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def gelu(x):
    """nn.GELU() (exact erf form)."""
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def backbone_forward(sd, mel, cfg=VocosConfig, dtype=np.float64, taps=None):
    """backbone.py:50-56.  mel (B, input_channels, T) -> (B, T, dim)."""
    g = lambda k: sd[k].astype(dtype)
    x = conv1d_same(mel.astype(dtype), g("backbone.embed.weight"), g("backbone.embed.bias"))                  # :51
    x = layer_norm(x.transpose(0, 2, 1), g("backbone.norm.weight"), g("backbone.norm.bias")).transpose(0, 2, 1)   # :52
    if taps is not None: taps["embed"] = x.transpose(0, 2, 1).copy()
    for i in range(cfg.num_layers):                                                                           # :53-54
        p = f"backbone.convnext.{i}."
        res = x                                                                                               # module.py:34
        h = conv1d_same(x, g(p + "dwconv.weight"), g(p + "dwconv.bias"), groups=cfg.dim)                      # :35
        h = layer_norm(h.transpose(0, 2, 1), g(p + "norm.weight"), g(p + "norm.bias"))                       # :36-37
        h = gelu(h @ g(p + "pwconv1.weight").T + g(p + "pwconv1.bias"))                                       # :38-39
        h = h @ g(p + "pwconv2.weight").T + g(p + "pwconv2.bias")                                             # :40
        h = g(p + "gamma") * h                                                                                # :41-42
        x = res + h.transpose(0, 2, 1)                                                                        # :43-45
        if taps is not None: taps[f"block{i}"] = x.transpose(0, 2, 1).copy()
    return layer_norm(x.transpose(0, 2, 1), g("backbone.final_layer_norm.weight"), g("backbone.final_layer_norm.bias"))   # :55


def istft_same(spec, window, n_fft, hop):
    """head.py:39-72 with padding == "same", win_length == n_fft.  spec (B, n_fft/2+1, T) complex -> (B, T*hop)."""
    B, N, T = spec.shape
    pad = (n_fft - hop) // 2                                                  # :48
    frames = np.fft.irfft(spec, n_fft, axis=1) * window[None, :, None]        # :56-57 (norm="backward")
    out_len = (T - 1) * hop + n_fft                                           # :60
    y = np.zeros((B, out_len), frames.dtype)
    env = np.zeros(out_len, frames.dtype)
    for t in range(T):                                                        # :61-69: fold == overlap-add
        y[:, t * hop:t * hop + n_fft] += frames[:, :, t]
        env[t * hop:t * hop + n_fft] += window ** 2
    y, env = y[:, pad:out_len - pad], env[pad:out_len - pad]
    assert (env > 1e-11).all()                                                # :72
    return y / env                                                            # :73


def head_forward(sd, x, cfg=VocosConfig, dtype=np.float64, taps=None):
    """head.py:93-117.  x (B, T, dim) -> audio (B, T*hop)."""
    o = (x @ sd["head.out.weight"].astype(dtype).T + sd["head.out.bias"].astype(dtype)).transpose(0, 2, 1)   # :103
    half = o.shape[1] // 2
    mag, p = o[:, :half], o[:, half:]                                         # :104
    mag = np.minimum(np.exp(mag), 1e2)                                        # :105-106
    S = mag * (np.cos(p) + 1j * np.sin(p))                                    # :108-115
    if taps is not None: taps["head_out"] = o.transpose(0, 2, 1).copy()
    return istft_same(S, sd["head.istft.window"].astype(dtype), cfg.n_fft, cfg.hop_length)                   # :116


def vocos_forward(sd, mel, cfg=VocosConfig, dtype=np.float64, taps=None):
    """model.py:17-20: audio = head(backbone(mel))."""
    return head_forward(sd, backbone_forward(sd, mel, cfg, dtype, taps), cfg, dtype, taps)
