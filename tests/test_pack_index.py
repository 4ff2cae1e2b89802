"""CPU check of the weight-layout index maps the kernels and the packers share (csrc/common.h: ffn_stream_index, ffn_wino_index,
qkv_frag_index): compiled for the HOST with hipcc, no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weight_stream_and_fragment_maps_are_bijections(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "pack_index_check")
    src = os.path.join(ROOT, "tests", "host", "pack_index_check.cpp")
    r = subprocess.run([hipcc, "-x", "hip", "--cuda-host-only", "-O1", "-std=c++17", src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout + r.stderr


def test_fragment_layouts_are_bank_conflict_free():
    """tools/lds_bank_check.py: the 32x32x16 B-fragment read and the Winograd kernel's pair-interleaved raw rows under the
    ds_read_b128 lane-group model of MI355X_MICROARCH.md."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_bank_check", os.path.join(ROOT, "tools", "lds_bank_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    w32 = max(m.worst(lambda l: m.swz(base + tap + (l & 31), 2 * ks + (l >> 5))) for base in (0, 32, 64, 96) for tap in range(3) for ks in range(4))
    assert w32 == 1
    # the Winograd fused FFN's raw rows: lane i reads row 2 i + e; pair-interleaved storage is conflict-free, the plain layout is not
    for lay, want in ((m.swz, 2), (m.pair_interleaved, 1)):
        ww = max(m.worst(lambda l: lay(2 * (32 * b + (l & 31)) + e, 2 * ks + (l >> 5))) for b in range(2) for e in range(4) for ks in range(4))
        assert ww == want
    pos = {m.pair_interleaved(r, c) for r in range(136) for c in range(8)}
    assert len(pos) == 136 * 8 and max(pos) < 136 * 128
