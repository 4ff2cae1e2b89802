"""Generates tests/golden/text_encoder_outputs.npz from the REAL reference TextEncoder
(/root/reference/models/text_encoder.py, unmodified) with the seeded weights of
oracle/weights.make_text_encoder_state_dict.  Run where /root/reference is mounted:

    python oracle/make_golden_text_encoder.py

Test infrastructure: the GPU box never has the reference, only the committed vectors.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "text_encoder_outputs.npz")


def text_inputs(B, T, lengths, seed, n_vocab=401, gin=256):
    """Seeded phoneme ids in [1, n_vocab) interspersed with 0 (text/__init__.py intersperse), speaker vectors."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tok = rng.integers(1, n_vocab, size=(B, T)).astype(np.int64)
    tok[:, 0::2] = 0
    for b, L in enumerate(lengths):
        tok[b, L:] = 0
    c = rng.standard_normal((B, gin)).astype(np.float32)
    return torch.from_numpy(tok), torch.from_numpy(c), torch.tensor(lengths, dtype=torch.long)


CASES = {"small": (3, 37, [37, 25, 9], 21), "long": (2, 200, [200, 131], 22)}


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not mounted at {REF}")
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    from models.text_encoder import TextEncoder               # reference, unmodified
    from oracle.weights import TextEncoderConfig, make_text_encoder_state_dict
    cfg = TextEncoderConfig()
    sd = make_text_encoder_state_dict(2468, cfg)
    enc = TextEncoder(cfg.n_vocab, cfg.out_channels, cfg.hidden_channels, cfg.filter_channels, cfg.n_heads,
                      cfg.n_layers, cfg.kernel_size, cfg.p_dropout, cfg.gin_channels).eval()
    print("load:", enc.load_state_dict(sd, strict=True), sum(p.numel() for p in enc.parameters()), "params")
    res = {}
    with torch.inference_mode():
        for name, (B, T, lengths, seed) in CASES.items():
            tok, c, lens = text_inputs(B, T, lengths, seed)
            x, mu_x, mask = enc(tok, c, lens)
            res[name + "_x"], res[name + "_mu_x"], res[name + "_mask"] = x.numpy(), mu_x.numpy(), mask.numpy()
            print(name, x.shape, mu_x.shape, float(x.abs().max()), float(mu_x.abs().max()))
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
