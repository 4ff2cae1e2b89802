"""CPU restatement (numpy) of the duration -> alignment -> mu_y step of the reference (test infrastructure).

Follows models/model.py:85-95 (StableTTS.synthesise) and generate_path (:17-27), utils/mask.py (sequence_mask).
Pinned against outputs of the REAL helpers by oracle/make_golden_align.py -> tests/golden/align_outputs.npz
(tests/test_oracle_golden.py).  NOT part of the product: stabletts_amd/ never imports this.
"""
import numpy as np


def generate_path(duration, mask):
    """models/model.py:17-27.  duration (B, Tx) fp32, mask (B, Tx, Ty) -> path (B, Tx, Ty)."""
    b, t_x, t_y = mask.shape
    cum = np.cumsum(duration.astype(np.float32), axis=1, dtype=np.float32)           # :19 (sequential fp32)
    j = np.arange(t_y, dtype=np.float32)
    path = (j[None, None, :] < cum[:, :, None]).astype(mask.dtype)                    # :22-24 sequence_mask(cum, t_y)
    path = path - np.pad(path, ((0, 0), (1, 0), (0, 0)))[:, :-1]                      # :25
    return path * mask                                                                # :26


def length_regulate(logw, x_mask, mu_x, length_scale=1.0):
    """models/model.py:85-95: returns dict(w_ceil, y_lengths, y_mask, attn, mu_y)."""
    w = np.exp(logw.astype(np.float32)) * x_mask                                      # :85
    w_ceil = (np.ceil(w) * np.float32(length_scale)).astype(np.float32)               # :86
    y_lengths = np.maximum(w_ceil.sum(axis=(1, 2), dtype=np.float32), 1).astype(np.int64)   # :87
    t_y = int(y_lengths.max())                                                        # :88
    y_mask = (np.arange(t_y)[None, :] < y_lengths[:, None]).astype(np.float32)[:, None, :]  # :91
    attn_mask = x_mask[:, 0, :, None] * y_mask[:, 0, None, :]                         # :92
    attn = generate_path(w_ceil[:, 0], attn_mask)                                     # :93
    mu_y = np.einsum("bij,bmi->bmj", attn, mu_x.astype(np.float32))                   # :94-95
    return dict(w_ceil=w_ceil, y_lengths=y_lengths, y_mask=y_mask, attn=attn, mu_y=mu_y)
