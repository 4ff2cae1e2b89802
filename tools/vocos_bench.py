#!/usr/bin/env python
"""Throughput of the native Vocos vocoder (developer tool): mel (B, 128, T) -> audio, seconds of 44.1 kHz audio per
second and the per-class kernel times.  python tools/vocos_bench.py [B] [T] [dtype]"""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vocos_oracle as vo                 # noqa: E402  (seeded weights / inputs only)
from stabletts_amd.vocos import Vocos                 # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dt = sys.argv[3] if len(sys.argv) > 3 else "f16"
c = vo.VocosConfig
m = Vocos(types.SimpleNamespace(input_channels=c.input_channels, dim=c.dim, intermediate_dim=c.intermediate_dim, num_layers=c.num_layers),
          types.SimpleNamespace(n_fft=c.n_fft, hop_length=c.hop_length), operand_dtype=dt)
m.load_state_dict({k: torch.from_numpy(v) for k, v in vo.make_vocos_state_dict(77).items()})
m = m.cuda()
mel = torch.from_numpy(vo.make_mel(B, T, 3)).cuda()
for _ in range(3):
    m(mel)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    m(mel)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
eng = m.engine(); eng.profile_enable(True); m(mel); torch.cuda.synchronize()
pr = eng.profile_read(); eng.profile_enable(False)
flops = B * T * (2 * 896 * 512 + 8 * 4 * 512 * 1536 + 2 * 512 * 2050)
print(json.dumps({"B": B, "T": T, "dtype": dt, "ms": round(ms, 3), "frames_per_s": round(B * T / ms * 1e3),
                  "audio_s_per_s": round(B * T * 512 / 44100 / ms * 1e3, 1), "gemm_tflops": round(flops / ms / 1e9, 1),
                  "classes_ms": {k: round(v["total_ms"], 3) for k, v in pr.items() if v["launches"]}}))
