"""Drop-in replacement for the reference's ``models/text_encoder.py`` (the caller side
of the hot path, on the same DiT block kernels).

``TextEncoder`` keeps the reference constructor (models/text_encoder.py:9), the ``forward(x, c, x_lengths)``
signature (:34) and the checkpoint key layout (``emb.weight``, ``encoder.<i>.attn.conv_q.weight``, ...,
``proj.bias``), but embedding, the ``n_layers`` DiTConVBlocks (adaLN-Zero, RoPE attention, conv-FFN) and the
output projection run as hand-written gfx950 kernels behind ``st_text_encoder_forward``
(include/stabletts_hip.h).  Inference only (no autograd graph); there is no PyTorch fallback.
"""
import torch
import torch.nn as nn

from . import _lib
from .estimator import DiTConVBlock, _param_key


class TextEncoder(nn.Module):
    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size,
                 p_dropout, gin_channels, operand_dtype="f16"):
        super().__init__()
        self.n_vocab, self.out_channels, self.hidden_channels = n_vocab, out_channels, hidden_channels
        self.filter_channels, self.n_heads, self.n_layers = filter_channels, n_heads, n_layers
        self.kernel_size, self.p_dropout, self.gin_channels = kernel_size, p_dropout, gin_channels
        self.operand_dtype = operand_dtype
        self.scale = self.hidden_channels ** 0.5

        self.emb = nn.Embedding(n_vocab, hidden_channels)
        nn.init.normal_(self.emb.weight, 0.0, hidden_channels ** -0.5)
        self.encoder = nn.ModuleList([DiTConVBlock(hidden_channels, filter_channels, n_heads, kernel_size, gin_channels)
                                      for _ in range(n_layers)])
        self.proj = nn.Conv1d(hidden_channels, out_channels, 1)
        self.initialize_weights()
        self._engine = None
        self._engine_key = None

    def initialize_weights(self):
        """adaLN-Zero (models/text_encoder.py:29-32)."""
        for block in self.encoder:
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)

    def __getstate__(self):
        st = self.__dict__.copy()      # the ctypes engine handle is per-process, never copied/pickled
        st["_engine"] = None
        st["_engine_key"] = None
        return st

    def _param_key(self):
        return _param_key(self)

    def sync_weights(self):
        """Force a weight re-upload at the next call (after writes through ``p.data`` that bypass the version counter)."""
        self._engine_key = None

    def _apply(self, fn, *a, **k):
        self._engine_key = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._engine_key = None
        return super()._load_from_state_dict(*a, **k)

    def engine(self):
        """The native handle bound to the device of the parameters, with weights in sync."""
        p0 = next(self.parameters())
        if p0.device.type != "cuda":
            raise RuntimeError("stabletts_amd: the text encoder runs only on a HIP device (move the module with "
                               ".to('cuda')); there is no CPU fallback")
        dev = p0.device.index if p0.device.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device != dev or self._engine.operand_dtype != self.operand_dtype:
            if self._engine is not None:
                self._engine.close()
            self._engine = _lib.Engine(self.out_channels, self.hidden_channels, self.filter_channels, self.n_heads,
                                       self.n_layers, self.kernel_size, self.gin_channels, self.operand_dtype, dev,
                                       text_encoder_vocab=self.n_vocab)
            self._engine_key = None
        key = self._param_key()
        if key != self._engine_key:
            with torch.no_grad():
                torch.cuda.synchronize(dev)
                self._engine.load_state_dict(self.state_dict())
            self._engine_key = key
        return self._engine

    def forward(self, x: torch.Tensor, c: torch.Tensor, x_lengths: torch.Tensor):
        """x: (B, T) phoneme ids, c: (B, gin) speaker vectors, x_lengths: (B,) ->
        (x (B, hidden, T), mu_x (B, out, T), x_mask (B, 1, T)) exactly as models/text_encoder.py:34-44."""
        if self.emb.weight.device.type != "cuda":
            self.engine()      # raises: no CPU fallback
        if torch.is_grad_enabled() and (c.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("the native TextEncoder is inference-only (no backward kernels): call it under "
                                      "torch.no_grad(), or train with the reference module and load its checkpoint")
        with torch.no_grad():
            return self._forward(x, c, x_lengths)

    def _forward(self, x, c, x_lengths):
        eng = self.engine()
        dev = self.emb.weight.device
        for name, t in (("x", x), ("c", c), ("x_lengths", x_lengths)):
            if t.device != dev:
                raise ValueError(f"{name} is on {t.device}, the encoder's parameters are on {dev}")
        if x.dim() != 2 or c.shape != (x.shape[0], self.gin_channels) or x_lengths.shape != (x.shape[0],):
            raise ValueError("shape mismatch: x (B,T) ids, c (B,gin), x_lengths (B,)")
        B, T = x.shape
        tok = x.detach().to(device=dev, dtype=torch.long).contiguous()
        lens = x_lengths.detach().to(device=dev, dtype=torch.long).contiguous()
        cc = c.detach().to(device=dev, dtype=torch.float32).contiguous()
        h = torch.empty(B, self.hidden_channels, T, device=dev, dtype=torch.float32)
        mu_x = torch.empty(B, self.out_channels, T, device=dev, dtype=torch.float32)
        mask = torch.empty(B, 1, T, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            eng.text_encoder_forward(tok, lens, cc, h, mu_x, mask, torch.cuda.current_stream(dev).cuda_stream)
        return h, mu_x, mask
