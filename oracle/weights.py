"""Seeded weight factory for the CFM decoder estimator (test infrastructure).

Produces a state_dict with exactly the 116 tensors / names / shapes of the
reference's ``decoder.estimator.*`` (reference
models/estimator.py:65-101, models/diffusion_transformer.py:10-96), drawn from a
numpy PCG64 stream so the values do not depend on the torch version.

Distributions follow the reference's effective initialisation (PyTorch default
Conv1d/Linear init = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias;
xavier_uniform for conv_q/k/v, models/diffusion_transformer.py:54-56) EXCEPT the
adaLN-Zero output layers, which the reference zero-initialises
(models/estimator.py:98-101): with zeros every gate is 0 and each DiT block is the
identity, so a parity test would never exercise attention/FFN.  They are drawn
N(0, 0.02) instead.
"""
from dataclasses import dataclass
import math

import numpy as np
import torch


@dataclass(frozen=True)
class DecoderConfig:
    """Constructor arguments of reference CFMDecoder (models/flow_matching.py:12)."""
    noise_channels: int = 128
    cond_channels: int = 128
    hidden_channels: int = 256
    out_channels: int = 128
    filter_channels: int = 1024
    n_heads: int = 4
    n_layers: int = 6
    kernel_size: int = 3
    p_dropout: float = 0.1
    gin_channels: int = 256


def _uniform(rng, shape, bound):
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def _conv(rng, sd, name, cout, cin, k, xavier=False):
    fan_in = cin * k
    if xavier:
        bound = math.sqrt(6.0 / (cin * k + cout * k))
    else:
        bound = 1.0 / math.sqrt(fan_in)
    sd[name + ".weight"] = _uniform(rng, (cout, cin, k), bound)
    sd[name + ".bias"] = _uniform(rng, (cout,), 1.0 / math.sqrt(fan_in))


def _linear(rng, sd, name, cout, cin):
    bound = 1.0 / math.sqrt(cin)
    sd[name + ".weight"] = _uniform(rng, (cout, cin), bound)
    sd[name + ".bias"] = _uniform(rng, (cout,), bound)


def make_state_dict(seed: int = 1234, cfg: DecoderConfig = DecoderConfig(), ada_std: float = 0.02):
    """state_dict of reference ``Decoder`` (models/estimator.py:65) with seeded values."""
    rng = np.random.Generator(np.random.PCG64(seed))
    C, F, M, G = cfg.hidden_channels, cfg.filter_channels, cfg.noise_channels, cfg.gin_channels
    K = cfg.kernel_size
    sd = {}
    _linear(rng, sd, "time_mlp.layer.0", F, C)
    _linear(rng, sd, "time_mlp.layer.2", C, F)
    _conv(rng, sd, "in_proj", C, C + M, 1)
    for i in range(cfg.n_layers):
        p = f"blocks.{i}."
        _conv(rng, sd, p + "time_fusion.film", 2 * C, C, 1)
        for nm in ("q", "k", "v"):
            _conv(rng, sd, p + f"block.attn.conv_{nm}", C, C, 1, xavier=True)
        _conv(rng, sd, p + "block.attn.conv_o", C, C, 1)
        _conv(rng, sd, p + "block.mlp.conv_1", F, C, K)
        _conv(rng, sd, p + "block.mlp.conv_2", C, F, K)
        if G != C:
            _linear(rng, sd, p + "block.adaLN_modulation.0", C, G)
        sd[p + "block.adaLN_modulation.2.weight"] = torch.from_numpy(
            (rng.standard_normal((6 * C, C)) * ada_std).astype(np.float32))
        sd[p + "block.adaLN_modulation.2.bias"] = torch.from_numpy(
            (rng.standard_normal((6 * C,)) * ada_std).astype(np.float32))
    _conv(rng, sd, "final_proj", cfg.out_channels, C, 1)
    _conv(rng, sd, "cond_proj.0", F, cfg.cond_channels, K)
    _conv(rng, sd, "cond_proj.2", F, F, K)
    _conv(rng, sd, "cond_proj.4", C, F, K)
    for i in range(cfg.n_layers // 2):
        _conv(rng, sd, f"lsc_layers.{i}", C, 2 * C, K)
    return sd


def make_cfg_params(seed: int = 4321, cfg: DecoderConfig = DecoderConfig(), std: float = 0.1):
    """Non-zero stand-ins for StableTTS.fake_speaker / fake_content (models/model.py:43-44)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fs = torch.from_numpy((rng.standard_normal((1, cfg.gin_channels)) * std).astype(np.float32))
    fc = torch.from_numpy((rng.standard_normal((1, cfg.cond_channels, 1)) * std).astype(np.float32))
    return fs, fc


@dataclass(frozen=True)
class TextEncoderConfig:
    """Constructor arguments of reference TextEncoder (models/text_encoder.py:9) as StableTTS builds it
    (models/model.py:36; config.py ModelConfig; text/symbols.py: 401 symbols)."""
    n_vocab: int = 401
    out_channels: int = 128
    hidden_channels: int = 256
    filter_channels: int = 1024
    n_heads: int = 4
    n_layers: int = 3
    kernel_size: int = 3
    p_dropout: float = 0.1
    gin_channels: int = 256


def make_text_encoder_state_dict(seed: int = 2468, cfg: TextEncoderConfig = TextEncoderConfig(), ada_std: float = 0.02):
    """state_dict of reference ``TextEncoder`` (models/text_encoder.py:8-32) with seeded values: embedding
    N(0, hidden^-0.5) (:23), block convs / linears as in make_state_dict, adaLN-Zero output layers re-randomised
    N(0, ada_std) for the same reason (zero init, :30-32, would make every block the identity)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    C, F, G, K = cfg.hidden_channels, cfg.filter_channels, cfg.gin_channels, cfg.kernel_size
    sd = {}
    sd["emb.weight"] = torch.from_numpy((rng.standard_normal((cfg.n_vocab, C)) * C ** -0.5).astype(np.float32))
    for i in range(cfg.n_layers):
        p = f"encoder.{i}."
        for nm in ("q", "k", "v"):
            _conv(rng, sd, p + f"attn.conv_{nm}", C, C, 1, xavier=True)
        _conv(rng, sd, p + "attn.conv_o", C, C, 1)
        _conv(rng, sd, p + "mlp.conv_1", F, C, K)
        _conv(rng, sd, p + "mlp.conv_2", C, F, K)
        if G != C:
            _linear(rng, sd, p + "adaLN_modulation.0", C, G)
        sd[p + "adaLN_modulation.2.weight"] = torch.from_numpy((rng.standard_normal((6 * C, C)) * ada_std).astype(np.float32))
        sd[p + "adaLN_modulation.2.bias"] = torch.from_numpy((rng.standard_normal((6 * C,)) * ada_std).astype(np.float32))
    _conv(rng, sd, "proj", cfg.out_channels, C, 1)
    return sd
