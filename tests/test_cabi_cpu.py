"""C-ABI library: loads on a CPU-only box, exports every symbol include/stabletts_hip.h declares,
validates configurations like the reference constructor does, and fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from stabletts_amd.build import build
    build(verbose=False)
    from stabletts_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "stabletts_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(st_[a-z_0-9]+)\s*\(", hdr))
    from stabletts_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.st_abi_version() == 4


def _create(lib, **over):
    from stabletts_amd._lib import StConfig
    base = dict(noise_channels=128, hidden_channels=256, filter_channels=1024, n_heads=4, n_layers=6,
                kernel_size=3, gin_channels=256, operand_dtype=0)
    base.update(over)
    cfg = StConfig(**base)
    h = ctypes.c_void_p()
    rc = lib.st_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    return rc, h, lib.st_last_error(None).decode()


def test_reference_assertions_are_mirrored(lib):
    rc, _, msg = _create(lib, n_layers=5)
    assert rc == -1 and "estimator.py:92" in msg
    rc, _, msg = _create(lib, n_heads=3)
    assert rc == -1 and "diffusion_transformer.py:35" in msg
    rc, _, msg = _create(lib, hidden_channels=512, n_heads=8)
    assert rc == -4 and "hidden_channels" in msg          # valid in the reference, not built natively
    rc, _, msg = _create(lib, kernel_size=5)
    assert rc == -4


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
def test_no_gpu_fails_loudly(lib):
    rc, _, msg = _create(lib)
    assert rc == -2 and "device" in msg
    from stabletts_amd.flow_matching import CFMDecoder
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    mu = torch.zeros(1, 128, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(mu, torch.ones(1, 1, 8), 2, 1.0, torch.zeros(1, 256), "euler")


def test_shim_mirrors_reference_interface():
    import inspect
    from stabletts_amd.flow_matching import CFMDecoder
    import oracle
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    sd = dec.estimator.state_dict()
    ref = oracle.make_state_dict(1234)
    assert set(sd) == set(ref) and all(sd[k].shape == ref[k].shape for k in ref)
    assert dec.sigma_min == 1e-4
    params = list(inspect.signature(dec.forward).parameters)
    assert params[:7] == ["mu", "mask", "n_timesteps", "temperature", "c", "solver", "cfg_kwargs"]
    # adaLN-Zero init like the reference (estimator.py:98-101)
    assert float(sd["blocks.0.block.adaLN_modulation.2.weight"].abs().max()) == 0.0
    with pytest.raises(NotImplementedError):      # a torchdiffeq method without a native controller (every webui.py:110 method has one)
        dec(torch.zeros(1, 128, 8), torch.ones(1, 1, 8), 2, 1.0, torch.zeros(1, 256), "explicit_adams")
    from stabletts_amd import _lib
    assert set(_lib.SOLVERS) >= {"euler", "midpoint", "dopri5", "rk4", "implicit_adams", "bosh3", "fehlberg2", "adaptive_heun", None}
    with pytest.raises(AssertionError):
        CFMDecoder(128, 128, 256, 128, 1024, 4, 5, 3, 0.1, 256)       # n_layers % 2 (estimator.py:92)
    with pytest.raises(RuntimeError, match="no CPU fallback"):        # training path is native too: no CPU fallback either
        dec.compute_loss(torch.zeros(1, 128, 8), torch.ones(1, 1, 8), torch.zeros(1, 128, 8), torch.zeros(1, 256))


def test_install_registers_dropin_module():
    import sys
    import stabletts_amd
    saved = sys.modules.pop("models.flow_matching", None)
    try:
        m = stabletts_amd.install()
        assert sys.modules["models.flow_matching"] is m and hasattr(m, "CFMDecoder")
    finally:
        sys.modules.pop("models.flow_matching", None)
        if saved is not None:
            sys.modules["models.flow_matching"] = saved


def test_text_encoder_shim_mirrors_reference_interface():
    """models/text_encoder.py:9,34: constructor, forward(x, c, x_lengths), checkpoint key layout, adaLN-Zero init,
    and no CPU fallback."""
    import inspect
    import sys
    import oracle
    import stabletts_amd
    from stabletts_amd.text_encoder import TextEncoder
    enc = TextEncoder(401, 128, 256, 1024, 4, 3, 3, 0.1, 256)
    sd = enc.state_dict()
    ref = oracle.make_text_encoder_state_dict(2468)
    assert set(sd) == set(ref) and all(sd[k].shape == ref[k].shape for k in ref)
    assert list(inspect.signature(enc.forward).parameters) == ["x", "c", "x_lengths"]
    assert float(sd["encoder.0.adaLN_modulation.2.weight"].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.zeros(1, 5, dtype=torch.long), torch.zeros(1, 256), torch.tensor([5]))
    saved = {k: sys.modules.pop(k, None) for k in ("models.flow_matching", "models.text_encoder")}
    try:
        stabletts_amd.install(text_encoder=True)
        assert sys.modules["models.text_encoder"].TextEncoder is TextEncoder
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def test_vocos_shim_mirrors_reference_interface():
    """vocoders/vocos/models/model.py:11-20: constructor Vocos(vocos_config, mel_config), forward(x), the state_dict
    layout of the REAL module (names + shapes recorded by oracle/make_golden_vocos.py), the reference's init
    (backbone.py:44-48, module.py:27-31), and no CPU fallback."""
    import inspect
    import os
    import sys
    import types
    import numpy as np
    import stabletts_amd
    from stabletts_amd.vocos import Vocos
    v = Vocos(types.SimpleNamespace(input_channels=128, dim=512, intermediate_dim=1536, num_layers=8),
              types.SimpleNamespace(n_fft=2048, hop_length=512))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocos_outputs.npz"))
    ref = dict(zip(g["state_dict.names"].tolist(), g["state_dict.shapes"].tolist()))
    sd = v.state_dict()
    assert list(sd) == list(ref)                                                   # same keys, same order
    assert all(",".join(map(str, sd[k].shape)) == ref[k] for k in ref)
    assert list(inspect.signature(v.forward).parameters) == ["x"]
    assert float(sd["backbone.convnext.3.gamma"][0]) == pytest.approx(1 / 8)       # layer_scale_init_value = 1 / num_layers
    assert float(sd["backbone.embed.bias"].abs().max()) == 0.0 and abs(float(sd["backbone.embed.weight"].std()) - 0.02) < 2e-3
    assert torch.equal(sd["head.istft.window"], torch.hann_window(2048))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        v(torch.zeros(1, 128, 4))
    saved = sys.modules.pop("vocoders.vocos.models.model", None)
    try:
        stabletts_amd.install(vocoder=True)
        assert sys.modules["vocoders.vocos.models.model"].Vocos is Vocos
    finally:
        sys.modules.pop("vocoders.vocos.models.model", None)
        if saved is not None:
            sys.modules["vocoders.vocos.models.model"] = saved


def test_vocoder_config_limits(lib):
    """st_create_vocoder rejects what the native kernels are not built for (and never touches a GPU to do so)."""
    import ctypes
    from stabletts_amd._lib import StVocosConfig, ST_ERR_INVALID, ST_ERR_UNSUPPORTED
    h = ctypes.c_void_p()
    ok = dict(input_channels=128, dim=512, intermediate_dim=1536, num_layers=8, n_fft=2048, hop_length=512, operand_dtype=0)
    for bad, code in ((dict(dim=256), ST_ERR_UNSUPPORTED), (dict(n_fft=1024), ST_ERR_UNSUPPORTED),
                      (dict(hop_length=256), ST_ERR_UNSUPPORTED), (dict(num_layers=0), ST_ERR_INVALID),
                      (dict(operand_dtype=7), ST_ERR_INVALID), (dict(input_channels=100), ST_ERR_UNSUPPORTED)):
        cfg = StVocosConfig(**{**ok, **bad})
        assert lib.st_create_vocoder(ctypes.byref(cfg), 0, ctypes.byref(h)) == code, bad
        assert lib.st_last_error(None)


def test_winograd_opt_in_turns_the_finite_check_on(monkeypatch):
    """The FFN default is the direct fused kernel; the Winograd form (ST_FUSED_FFN=3, f16 only) halves the f16 range of the FFN
    intermediate, so a decoder created under it checks its output by itself (an explicit check_finite wins either way)."""
    from stabletts_amd.flow_matching import CFMDecoder
    args = (128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    monkeypatch.delenv("ST_FUSED_FFN", raising=False)
    assert CFMDecoder(*args).check_finite is False
    monkeypatch.setenv("ST_FUSED_FFN", "3")
    assert CFMDecoder(*args).check_finite is True
    assert CFMDecoder(*args, operand_dtype="bf16").check_finite is False      # bf16 engines never run the Winograd kernel
    assert CFMDecoder(*args, check_finite=False).check_finite is False
    monkeypatch.setenv("ST_FUSED_FFN", "1")
    assert CFMDecoder(*args).check_finite is False and CFMDecoder(*args, check_finite=True).check_finite is True


def test_bench_env_lists_cover_every_variable_the_engine_reads():
    """bench.py classifies the environment by explicit lists: every getenv("ST_*") of the engine sources (and the variables _lib.py /
    build.py read) must be in exactly one of them, so a new switch cannot slip through as 'neutral' by default."""
    import re
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    read = set()
    csrc = os.path.join(ROOT, "stabletts_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".cpp", ".h", ".hip")):
            read |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', open(os.path.join(csrc, f)).read()))
    for f in ("_lib.py", "build.py", "flow_matching.py", "estimator.py"):
        read |= set(re.findall(r'environ(?:\.get)?[\(\[]"((?:ST_|STABLETTS_)[A-Z_0-9]+)"', open(os.path.join(ROOT, "stabletts_amd", f)).read()))
    listed = set(bench.ENGINE_ENV) | set(bench.ENGINE_NEUTRAL_ENV)
    assert read <= listed, sorted(read - listed)
    assert not set(bench.ENGINE_ENV) & set(bench.ENGINE_NEUTRAL_ENV)


def test_bench_refuses_engine_changing_environment():
    """bench.py describes the library as shipped: with an ST_* variable that changes which kernels run (or STABLETTS_HIP_LIB) it
    refuses before touching the device; ST_SPLIT / ST_HIP_GRAPH only change how the same kernels are enqueued and pass."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if not k.startswith("ST_") and k != "STABLETTS_HIP_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env={**env, "ST_FUSED_FFN": "3"},
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to measure" in (r.stderr + r.stdout) and "ST_FUSED_FFN" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], env={**env, "ST_SPLIT": "1"},
                       capture_output=True, text=True, timeout=300)
    assert "refusing to measure" not in (r.stderr + r.stdout)       # (on a CPU-only box it then stops at "needs a HIP device")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
                       env={**env, "ST_BUILD_OUT": "/tmp/x.so", "ST_BUILD_DEFS": "-DX"}, capture_output=True, text=True, timeout=300)
    assert "refusing to measure" not in (r.stderr + r.stdout)       # build-only left-overs do not cost the harness its line
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a HIP device" in (r.stderr + r.stdout)


def test_micro_benchmarks_still_compile_against_the_kernel_headers():
    """pruning a kernel header silently broke tools/micro/ffn_bench.hip (it included the deleted ffn_fused16.h), i.e. the
    evidence tools DESIGN.md cites.  A front-end-only pass (hipcc -fsyntax-only, host + gfx950 device, ~8 s) over the micro-benchmarks
    that instantiate library kernels keeps them compiling."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    srcs = [os.path.join(ROOT, "tools", "micro", f) for f in ("ffn_bench.hip", "qkv_bench.hip", "ffn_wino_bench.hip")]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-Wno-unused-value", *srcs], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_design_md_numbers_come_from_the_committed_evidence():
    """DESIGN.md is docs/DESIGN.template.md with its tokens filled from profiles/r06_* by tools/fill_design.py: the end-state figures
    in the design document cannot drift from the committed bench lines / rocprofv3 summaries."""
    import subprocess
    import sys
    before = open(os.path.join(ROOT, "DESIGN.md")).read()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fill_design.py")], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    after = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert after == before, "DESIGN.md is stale: run python tools/fill_design.py"
    assert len(after.splitlines()) <= 400 and "\u27e8" not in after.encode("unicode_escape").decode()
