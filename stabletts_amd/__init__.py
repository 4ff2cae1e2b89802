"""stabletts_amd -- MI355X (gfx950) native flow-matching mel decoder for StableTTS.

Public surface mirrors the reference's ``models/flow_matching.py``:

    from stabletts_amd.flow_matching import CFMDecoder

``install()`` registers that module as ``models.flow_matching`` so the reference's
``models/model.py:7`` (``from models.flow_matching import CFMDecoder``) picks it up unmodified;
``install(text_encoder=True)`` also registers ``stabletts_amd.text_encoder`` as ``models.text_encoder``
(``models/model.py:6``), the caller side of the path on the same block kernels.
"""
import sys

__all__ = ["install", "CFMDecoder", "TextEncoder"]


def install(text_encoder=False, vocoder=False):
    """Make ``models.flow_matching`` (and optionally ``models.text_encoder`` / ``vocoders.vocos.models.model``)
    resolve to the native drop-ins (call before importing models.model / api.get_vocoder)."""
    from . import flow_matching
    sys.modules["models.flow_matching"] = flow_matching
    if text_encoder:
        from . import text_encoder as te
        sys.modules["models.text_encoder"] = te
    if vocoder:
        from . import vocos
        sys.modules["vocoders.vocos.models.model"] = vocos      # api.py:26-28: from vocoders.vocos.models.model import Vocos
    return flow_matching


def __getattr__(name):
    if name == "CFMDecoder":
        from .flow_matching import CFMDecoder
        return CFMDecoder
    if name == "Vocos":
        from .vocos import Vocos
        return Vocos
    if name == "TextEncoder":
        from .text_encoder import TextEncoder
        return TextEncoder
    raise AttributeError(name)
