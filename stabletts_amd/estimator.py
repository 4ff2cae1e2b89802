"""Parameter container + native forward for the vector-field estimator.

Mirrors the module tree of the reference ``Decoder`` (models/estimator.py:65-96) and
``DiTConVBlock`` (models/diffusion_transformer.py:82-96) so that ``state_dict()`` has exactly the
116 ``decoder.estimator.*`` names/shapes of released checkpoints and
DDP / AdamW / load_state_dict see ordinary ``nn.Parameter``s.  The modules hold parameters only;
all arithmetic runs in libstabletts_hip.so (hand-written gfx950 kernels) -- there is no PyTorch
fallback for the forward pass.
"""
import torch
import torch.nn as nn

from . import _lib


def _param_key(module):
    """(storage identity, version counters) of the parameters.  The first changes when a tensor moves (``.to()``,
    ``.half()``, re-assignment), the second whenever autograd-visible code rewrites a weight in place (optimizer step,
    ``load_state_dict``).  Inference tensors (a module built or moved under torch.inference_mode) carry no version
    counter; they key on the address alone and need sync_weights() after an in-place update."""
    ptrs, vers = [], []
    for p in module.parameters():
        try:
            ver = p._version
        except RuntimeError:
            ver = -1
        ptrs.append((p.data_ptr(), p.dtype, p.is_contiguous()))
        vers.append(ver)
    return tuple(ptrs), tuple(vers)


class _ParamsOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the computation runs in the native HIP engine")


class FiLMLayer(_ParamsOnly):
    """models/estimator.py:20-33."""
    def __init__(self, in_channels, cond_channels):
        super().__init__()
        self.film = nn.Conv1d(cond_channels, in_channels * 2, 1)


class MultiHeadAttention(_ParamsOnly):
    """models/diffusion_transformer.py:32-56 (parameters and init only)."""
    def __init__(self, channels, out_channels, n_heads):
        super().__init__()
        assert channels % n_heads == 0
        self.conv_q = nn.Conv1d(channels, channels, 1)
        self.conv_k = nn.Conv1d(channels, channels, 1)
        self.conv_v = nn.Conv1d(channels, channels, 1)
        self.conv_o = nn.Conv1d(channels, out_channels, 1)
        for m in (self.conv_q, self.conv_k, self.conv_v):
            nn.init.xavier_uniform_(m.weight)


class FFN(_ParamsOnly):
    """models/diffusion_transformer.py:10-23."""
    def __init__(self, in_channels, out_channels, filter_channels, kernel_size):
        super().__init__()
        self.conv_1 = nn.Conv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.conv_2 = nn.Conv1d(filter_channels, out_channels, kernel_size, padding=kernel_size // 2)


class DiTConVBlock(_ParamsOnly):
    """models/diffusion_transformer.py:82-96; LayerNorms have no parameters."""
    def __init__(self, hidden_channels, filter_channels, num_heads, kernel_size, gin_channels):
        super().__init__()
        self.attn = MultiHeadAttention(hidden_channels, hidden_channels, num_heads)
        self.mlp = FFN(hidden_channels, hidden_channels, filter_channels, kernel_size)
        self.adaLN_modulation = nn.Sequential(
            nn.Linear(gin_channels, hidden_channels) if gin_channels != hidden_channels else nn.Identity(),
            nn.SiLU(),
            nn.Linear(hidden_channels, 6 * hidden_channels, bias=True))


class DitWrapper(_ParamsOnly):
    """models/estimator.py:8-18."""
    def __init__(self, hidden_channels, filter_channels, num_heads, kernel_size, gin_channels, time_channels):
        super().__init__()
        self.time_fusion = FiLMLayer(hidden_channels, time_channels)
        self.block = DiTConVBlock(hidden_channels, filter_channels, num_heads, kernel_size, gin_channels)


class TimestepEmbedding(_ParamsOnly):
    """models/estimator.py:51-62."""
    def __init__(self, in_channels, out_channels, filter_channels):
        super().__init__()
        self.layer = nn.Sequential(nn.Linear(in_channels, filter_channels), nn.SiLU(inplace=True),
                                   nn.Linear(filter_channels, out_channels))


class Decoder(nn.Module):
    """Same constructor as the reference Decoder (models/estimator.py:66); forward is native.

    forward(t, x, mask, mu, c) -> (B, out_channels, T): one vector-field evaluation
    (models/estimator.py:103-138).  Under torch.no_grad()/inference_mode it is the plain native launch sequence;
    when gradients are required it goes through stabletts_amd/autograd.py (native forward that keeps the
    activations + native backward), so DDP / AdamW see ordinary parameter gradients.
    """

    def __init__(self, noise_channels, cond_channels, hidden_channels, out_channels, filter_channels,
                 dropout=0.1, n_layers=1, n_heads=4, kernel_size=3, gin_channels=0, use_lsc=True,
                 operand_dtype="f16", attention_precision="16bit"):
        super().__init__()
        assert hidden_channels % 2 == 0, "SinusoidalPosEmb requires dim to be even"
        if not use_lsc:
            raise NotImplementedError("native estimator is built with the U-Net long-skip connections (use_lsc=True)")
        assert n_layers % 2 == 0
        if not (noise_channels == cond_channels == out_channels):
            raise NotImplementedError("native estimator expects noise == cond == out channels (n_mels)")
        self.noise_channels, self.cond_channels = noise_channels, cond_channels
        self.hidden_channels, self.out_channels = hidden_channels, out_channels
        self.filter_channels, self.n_layers, self.n_heads = filter_channels, n_layers, n_heads
        self.kernel_size, self.gin_channels, self.use_lsc = kernel_size, gin_channels, use_lsc
        self.operand_dtype = operand_dtype
        # "16bit": q, k, v enter the attention MFMAs as 16-bit operands (default).  "split": q and k as hi + lo operand pairs (three
        # QK^T products: +~50 % attention time) for checkpoints whose softmax is an arg-max.  "auto": starts at "16bit"; CFMDecoder
        # reads the engine's attention statistic after a solve and switches to "split" when the largest log-sum-exp exceeds
        # AUTO_SPLIT_LSE (one stream synchronisation per solve while still undecided, none afterwards).
        if attention_precision not in ("16bit", "split", "auto"):
            raise ValueError("attention_precision must be '16bit', 'split' or 'auto'")
        self.attention_precision = attention_precision
        self._attn_split = attention_precision == "split"
        self.p_dropout = float(dropout)        # train-mode dropout of the FFN activations and attention probabilities

        self.time_mlp = TimestepEmbedding(hidden_channels, hidden_channels, filter_channels)
        self.in_proj = nn.Conv1d(hidden_channels + noise_channels, hidden_channels, 1)
        self.blocks = nn.ModuleList([DitWrapper(hidden_channels, filter_channels, n_heads, kernel_size,
                                                gin_channels, hidden_channels) for _ in range(n_layers)])
        self.final_proj = nn.Conv1d(hidden_channels, out_channels, 1)
        self.cond_proj = nn.Sequential(
            nn.Conv1d(cond_channels, filter_channels, kernel_size, padding=kernel_size // 2), nn.SiLU(inplace=True),
            nn.Conv1d(filter_channels, filter_channels, kernel_size, padding=kernel_size // 2), nn.SiLU(inplace=True),
            nn.Conv1d(filter_channels, hidden_channels, kernel_size, padding=kernel_size // 2))
        self.n_lsc_layers = n_layers // 2
        self.lsc_layers = nn.ModuleList([nn.Conv1d(2 * hidden_channels, hidden_channels, kernel_size,
                                                   padding=kernel_size // 2) for _ in range(self.n_lsc_layers)])
        self.initialize_weights()
        self._engine = None
        self._engine_key = None        # storage identity of the parameters the engine is bound to
        self._engine_vers = None       # their version counters at the last (re)pack
        self._staging = None           # fp32 copies the engine reads when the parameters themselves are not fp32

    def initialize_weights(self):
        """adaLN-Zero (models/estimator.py:98-101)."""
        for block in self.blocks:
            nn.init.constant_(block.block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.block.adaLN_modulation[-1].bias, 0)

    def __getstate__(self):
        st = self.__dict__.copy()      # the ctypes engine handle is per-process, never copied/pickled
        st["_engine"] = None
        st["_engine_key"] = st["_engine_vers"] = st["_staging"] = None
        return st

    # ------------------------------------------------------------------ native engine plumbing
    def _param_key(self):
        return _param_key(self)

    def sync_weights(self):
        """Force the engine to re-pack its 16-bit weight copies at the next call.  Needed only after writes that bypass
        autograd's version counter (``p.data.copy_(ema)``, ``m.weight.data.normal_()``, as some EMA / weight-swap
        utilities do); in-place ops on the parameters themselves, optimizer steps, ``load_state_dict`` and
        ``.to()`` are detected automatically."""
        self._engine_vers = None

    def release_engine(self):
        """Destroy the native engine (device arena, packed weights, streams); the next call builds a fresh one.  For a process that is
        done with this module for a while -- e.g. a synthesis model kept next to a training run."""
        if self._engine is not None:
            self._engine.close()
        self._engine = None
        self._engine_key = None
        self._engine_vers = None

    def _apply(self, fn, *a, **k):            # .to() / .cuda() / .half(): storage changes
        self._engine_key = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._engine_key = None
        return super()._load_from_state_dict(*a, **k)

    def engine(self):
        """The st_engine bound to the device of the parameters, with weights in sync.

        The engine READS the fp32 parameters where torch keeps them (st_bind_param: no copy); what it owns are the
        packed 16-bit MFMA operand copies.  After an in-place update (every optimizer step of a training loop) those
        are re-packed by kernels on the current stream (st_repack): no allocation, no host copy, no synchronisation.
        Only a change of storage (first use, ``.to()``, dtype change) takes the slow path (bind + st_finalize)."""
        p0 = next(self.parameters())
        if p0.device.type != "cuda":
            raise RuntimeError("stabletts_amd: the estimator runs only on a HIP device (move the module with "
                               ".to('cuda')); there is no CPU fallback")
        dev = p0.device.index if p0.device.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device != dev or self._engine.operand_dtype != self.operand_dtype:
            if self._engine is not None:
                self._engine.close()
            self._engine = _lib.Engine(self.noise_channels, self.hidden_channels, self.filter_channels, self.n_heads,
                                       self.n_layers, self.kernel_size, self.gin_channels, self.operand_dtype, dev)
            self._engine_key = None
            if getattr(self, "_attn_split", False):
                self._engine.set_option("attention_precision", 1)
        key, vers = self._param_key()
        if key != self._engine_key:
            with torch.no_grad():
                named = list(self.named_parameters())
                if all(p.dtype == torch.float32 and p.is_contiguous() for _, p in named):
                    self._staging = None
                    bound = [(n, p.detach()) for n, p in named]
                else:       # e.g. a .half() module: the engine reads fp32 staging copies
                    self._staging = [p.detach().to(dtype=torch.float32).contiguous() for _, p in named]
                    bound = [(n, s) for (n, _), s in zip(named, self._staging)]
                self._engine.bind_parameters(bound)        # st_finalize synchronises the device: pending writes have landed
            self._engine_key, self._engine_vers = key, vers
        elif vers != self._engine_vers:
            with torch.no_grad(), torch.cuda.device(dev):
                if self._staging is not None:
                    for s, p in zip(self._staging, self.parameters()):
                        s.copy_(p)
                self._engine.repack(torch.cuda.current_stream(dev).cuda_stream)
            self._engine_vers = vers
        return self._engine

    AUTO_SPLIT_LSE = 50.0      # natural units; seeded / initialised weights give ~10, the arg-max regime of DESIGN.md section 2 80-200

    def set_attention_precision(self, mode):
        """'16bit' / 'split' (see __init__); applies to the live engine at once."""
        if mode not in ("16bit", "split"):
            raise ValueError("mode must be '16bit' or 'split'")
        self._attn_split = mode == "split"
        if self._engine is not None:
            self._engine.set_option("attention_precision", int(self._attn_split))

    def device(self):
        return next(self.parameters()).device

    def _prep(self, t, dev=None, name="input"):
        """fp32 contiguous view of a caller tensor ON THE ENGINE'S DEVICE.  Like the reference modules, a tensor on
        another device is an error (raw pointers cross the C ABI: a CPU tensor would otherwise fault the GPU)."""
        dev = self.device() if dev is None else dev
        if t.device != dev:
            raise ValueError(f"{name} is on {t.device}, the estimator's parameters are on {dev}")
        return t.detach().to(dtype=torch.float32).contiguous()

    def forward(self, t, x, mask, mu, c):
        dev = self.device()
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or any(
            torch.is_tensor(v) and v.requires_grad for v in (x, mu, c)))
        if needs_grad:
            from .autograd import estimator_apply
            return estimator_apply(self, t, x, mask, mu, c)
        with torch.no_grad():
            eng = self.engine()
            B, M, T = x.shape
            if not torch.is_tensor(t):
                t = torch.tensor([float(t)])
            t = self._prep(t.reshape(-1).to(dev), dev, "t")      # the time is a scalar (or B scalars): host values are fine
            if t.numel() not in (1, B):
                raise ValueError("t must be a scalar or have one entry per batch item")
            x, mu, c = self._prep(x, dev, "x"), self._prep(mu, dev, "mu"), self._prep(c, dev, "c")
            mask = self._prep(mask, dev, "mask")
            if mu.shape != x.shape or mask.shape != (B, 1, T) or c.shape != (B, self.gin_channels):
                raise ValueError("shape mismatch: x/mu (B,M,T), mask (B,1,T), c (B,gin)")
            out = torch.empty_like(x)
            with torch.cuda.device(dev):
                eng.estimator_forward(t, x, mu, mask, c, out, torch.cuda.current_stream(dev).cuda_stream)
            return out
