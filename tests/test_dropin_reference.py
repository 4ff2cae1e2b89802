"""Drop-in check against the REAL reference package (CPU, build container only: skipped where /root/reference is
not mounted, e.g. on the GPU box).  With ``stabletts_amd.install(text_encoder=True)`` the unmodified
``models/model.py`` must build ``StableTTS`` around the native ``CFMDecoder`` / ``TextEncoder`` and end up with
exactly the checkpoint key layout (names and shapes) of the all-reference model, so released checkpoints load.
Only import stand-ins are used for packages absent offline (numba for monotonic_align, torchdiffeq)."""
import importlib
import os
import sys
import types

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference not mounted")


def _purge():
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "monotonic_align"
              or k.startswith("monotonic_align.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]


def _build(native):
    """-> StableTTS instance built by the reference's own models/model.py."""
    import stabletts_amd
    _purge()
    if native:
        stabletts_amd.install(text_encoder=True)
    model_mod = importlib.import_module("models.model")
    return model_mod.StableTTS(401, 128, 256, 1024, 4, 3, 6, 3, 0.1, 256)


def test_reference_model_builds_around_native_modules(monkeypatch):
    class _Ty:                                                  # numba.int32[:, :, ::1] etc. in the signature
        def __getitem__(self, item):
            return self

        def __call__(self, *a, **k):
            return self

    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda f: f)                  # monotonic_align/core.py decorator (training only)
    numba.void = numba.int32 = numba.float32 = _Ty()
    tde = types.ModuleType("torchdiffeq")
    tde.odeint = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stand-in"))
    monkeypatch.setitem(sys.modules, "numba", numba)
    monkeypatch.setitem(sys.modules, "torchdiffeq", tde)
    monkeypatch.syspath_prepend(REF)
    try:
        ref = _build(native=False)
        ref_sd = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert type(ref.decoder).__module__ == "models.flow_matching"
        nat = _build(native=True)
        assert type(nat.decoder).__module__ == "stabletts_amd.flow_matching"
        assert type(nat.encoder).__module__ == "stabletts_amd.text_encoder"
        nat_sd = {k: tuple(v.shape) for k, v in nat.state_dict().items()}
        assert nat_sd == ref_sd                                 # same names, same shapes: checkpoints are interchangeable
        missing, unexpected = nat.load_state_dict(ref.state_dict(), strict=True)
        assert not missing and not unexpected
        assert nat.decoder.sigma_min == ref.decoder.sigma_min
    finally:
        _purge()
