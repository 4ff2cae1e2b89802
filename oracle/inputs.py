"""Seeded synthetic inputs for the CFM decoder path (test infrastructure).

Shapes follow the call site models/model.py:99-103: mu (B, n_feats, T),
mask (B, 1, T) float 0/1 from sequence_mask, c (B, gin), plus an explicit
noise tensor z (B, n_feats, T) (the reference draws it internally,
models/flow_matching.py:45).
"""
import numpy as np
import torch


def make_inputs(B, T, seed=0, lengths=None, n_feats=128, gin=256, ragged=False, min_frac=0.6):
    """Returns dict(mu, mask, c, z, lengths). numpy PCG64 stream => torch-version independent.

    lengths: explicit list, or ragged=True -> U{ceil(min_frac*T)..T} with the max forced to T.
    mu is multiplied by the mask, as models/model.py:95-96 effectively does (mu_y is built
    from attn, which is zero on padded frames).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = torch.from_numpy(rng.standard_normal((B, n_feats, T)).astype(np.float32))
    c = torch.from_numpy(rng.standard_normal((B, gin)).astype(np.float32))
    z = torch.from_numpy(rng.standard_normal((B, n_feats, T)).astype(np.float32))
    if lengths is None:
        if ragged:
            lo = int(np.ceil(min_frac * T))
            lengths = rng.integers(lo, T + 1, size=B)
            lengths[int(rng.integers(0, B))] = T
            lengths = [int(v) for v in lengths]
        else:
            lengths = [T] * B
    lengths = torch.tensor(lengths, dtype=torch.long)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).to(torch.float32).unsqueeze(1)
    mu = mu * mask
    return dict(mu=mu, mask=mask, c=c, z=z, lengths=lengths)
