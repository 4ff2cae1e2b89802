"""stabletts_amd -- MI355X (gfx950) native flow-matching mel decoder for StableTTS.

Public surface mirrors the reference's ``models/flow_matching.py``:

    from stabletts_amd.flow_matching import CFMDecoder

``install()`` registers that module as ``models.flow_matching`` so the reference's
``models/model.py:7`` (``from models.flow_matching import CFMDecoder``) picks it up unmodified.
"""
import sys

__all__ = ["install", "CFMDecoder"]


def install():
    """Make ``models.flow_matching`` resolve to the native drop-in (call before importing models.model)."""
    from . import flow_matching
    sys.modules["models.flow_matching"] = flow_matching
    return flow_matching


def __getattr__(name):
    if name == "CFMDecoder":
        from .flow_matching import CFMDecoder
        return CFMDecoder
    raise AttributeError(name)
