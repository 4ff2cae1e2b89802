"""A plain-C host (examples/cabi_solve.c: gcc, no Python / torch / C++) drives the C ABI end to end: engine
creation, parameter discovery (st_param_info), upload, st_finalize, one Euler + CFG solve on device buffers.
The same seeded weights and inputs are rebuilt here in numpy and sent through the Python shim: both hosts call the
same library, so the mel must agree to rounding of the printed checksums."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lcg_uniform(seed, n):
    """examples/cabi_solve.c: lcg_uniform() stream, vectorised by affine-map doubling (s -> A*s + C mod 2^32)."""
    a, c, mask = 1664525, 1013904223, 0xFFFFFFFF
    out = np.empty(n, dtype=np.uint64)
    s1 = (seed * a + c) & mask
    out[0] = s1
    filled, A, C = 1, a, c                    # (A, C): the map for `filled` steps
    while filled < n:
        m = min(filled, n - filled)
        out[filled:filled + m] = (out[:m] * np.uint64(A) + np.uint64(C)) & np.uint64(mask)
        A, C = (A * A) & mask, (A * C + C) & mask
        filled += m
    return ((out >> np.uint64(8)).astype(np.float32) * np.float32(2.0 / 16777216.0) - np.float32(1.0)).astype(np.float32)


def fnv1a(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def fill(n, seed, scale):
    return torch.from_numpy(lcg_uniform(seed, n) * np.float32(scale))


def test_c_host_matches_python_host():
    exe = os.path.join(ROOT, "examples", "cabi_solve")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build_c_example()
    B, T, n = 2, 96, 4
    r = subprocess.run([exe, str(B), str(T), str(n)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"params=(\d+) sum=(\S+) abs=(\S+) first=(\S+) last=(\S+) finite=1", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == 116
    c_sum, c_abs, c_first, c_last = (float(m.group(i)) for i in range(2, 6))

    from stabletts_amd.flow_matching import CFMDecoder
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    sd = {}
    for name, p in dec.estimator.state_dict().items():
        shape = tuple(p.shape)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        f = 1
        while (f + 1) * (f + 1) <= fan_in:
            f += 1
        scale = 0.02 if "adaLN_modulation.2" in name else 1.0 / f
        sd[name] = fill(int(np.prod(shape)), fnv1a(name), scale).reshape(shape)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    info = dict(dec.estimator.engine().param_info())          # st_param_info: what a non-Python host discovers
    assert info == {k: tuple(v.shape) for k, v in sd.items()}
    M, G = 128, 256
    mu = fill(B * M * T, 11, 1.0).reshape(B, M, T)
    z = fill(B * M * T, 12, 1.0).reshape(B, M, T)
    c = fill(B * G, 13, 1.0).reshape(B, G)
    fs = fill(G, 14, 0.1).reshape(1, G)
    fc = fill(M, 15, 0.1).reshape(1, M, 1)
    mask = torch.zeros(B, 1, T)
    for b in range(B):
        mask[b, 0, :T - (b * T) // (3 * B)] = 1.0
    kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
    out = dec(mu.cuda(), mask.cuda(), n, 1.0, c.cuda(), "euler", kw, z=z.cuda()).cpu().double()
    assert abs(float(out.sum()) - c_sum) <= 1e-6 * c_abs
    assert abs(float(out.abs().sum()) - c_abs) <= 1e-6 * c_abs
    assert abs(float(out.flatten()[0]) - c_first) <= 1e-6 * max(1.0, abs(c_first))
    assert abs(float(out.flatten()[-1]) - c_last) <= 1e-6 * max(1.0, abs(c_last))
