"""Drop-in for the reference's Vocos vocoder: ``Vocos(vocos_config, mel_config)(mel) -> audio``
(vocoders/vocos/models/model.py:11-20; used by api.py:26-31,76).

Same module tree as the reference (``backbone.embed``, ``backbone.norm``, ``backbone.convnext.<i>.{dwconv,norm,
pwconv1,pwconv2,gamma}``, ``backbone.final_layer_norm``, ``head.out``, ``head.istft.window``), so a released
``vocos.pt`` loads with ``load_state_dict`` unchanged.  The modules hold parameters only: the forward pass runs in
libstabletts_hip.so (st_vocos_forward); there is no PyTorch fallback.  Inference only, like the reference's use of it.
"""
import torch
import torch.nn as nn

from . import _lib
from .estimator import _ParamsOnly, _param_key


class ConvNeXtBlock(_ParamsOnly):
    """vocoders/vocos/models/module.py:16-31 (parameters and init only)."""
    def __init__(self, dim, intermediate_dim, layer_scale_init_value):
        super().__init__()
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, intermediate_dim)
        self.pwconv2 = nn.Linear(intermediate_dim, dim)
        if not layer_scale_init_value > 0:
            raise NotImplementedError("native ConvNeXt block is built with the layer scale (gamma)")
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim), requires_grad=True)


class VocosBackbone(_ParamsOnly):
    """vocoders/vocos/models/backbone.py:21-48."""
    def __init__(self, input_channels, dim, intermediate_dim, num_layers, layer_scale_init_value=None):
        super().__init__()
        self.input_channels = input_channels
        self.embed = nn.Conv1d(input_channels, dim, kernel_size=7, padding=3)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        layer_scale_init_value = layer_scale_init_value or 1 / num_layers
        self.convnext = nn.ModuleList([ConvNeXtBlock(dim, intermediate_dim, layer_scale_init_value) for _ in range(num_layers)])
        self.final_layer_norm = nn.LayerNorm(dim, eps=1e-6)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.constant_(m.bias, 0)


class ISTFT(_ParamsOnly):
    """vocoders/vocos/models/head.py:19-28 (the window buffer)."""
    def __init__(self, n_fft, hop_length, win_length, padding="same"):
        super().__init__()
        if padding != "same":
            raise NotImplementedError('native ISTFT implements padding="same" (the reference head\'s default)')
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.register_buffer("window", torch.hann_window(win_length))


class ISTFTHead(_ParamsOnly):
    """vocoders/vocos/models/head.py:86-91."""
    def __init__(self, dim, n_fft, hop_length, padding="same"):
        super().__init__()
        self.out = nn.Linear(dim, n_fft + 2)
        self.istft = ISTFT(n_fft=n_fft, hop_length=hop_length, win_length=n_fft, padding=padding)


class Vocos(nn.Module):
    """Same constructor as the reference: ``Vocos(VocosConfig(), MelConfig())`` -- any objects with the attributes
    input_channels / dim / intermediate_dim / num_layers and n_fft / hop_length (config.py:4-19,46-50)."""

    def __init__(self, vocos_config, mel_config, operand_dtype="f16"):
        super().__init__()
        self.cfg = dict(input_channels=int(vocos_config.input_channels), dim=int(vocos_config.dim),
                        intermediate_dim=int(vocos_config.intermediate_dim), num_layers=int(vocos_config.num_layers),
                        n_fft=int(mel_config.n_fft), hop_length=int(mel_config.hop_length))
        self.operand_dtype = operand_dtype
        c = self.cfg
        self.backbone = VocosBackbone(c["input_channels"], c["dim"], c["intermediate_dim"], c["num_layers"])
        self.head = ISTFTHead(c["dim"], c["n_fft"], c["hop_length"])
        # inference-only module (the reference only ever calls it under inference_mode, api.py:64,76):
        # its parameters do not ask for gradients, so a plain ``voc(mel)`` is legal in any grad mode; asking for them
        # (requires_grad_(True) or a mel that requires grad) raises in forward instead of silently training nothing
        self.requires_grad_(False)
        self._engine = None
        self._engine_key = None

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_engine"] = None
        st["_engine_key"] = None
        return st

    def sync_weights(self):
        """Force a re-read of the parameters (after writes that bypass autograd's version counter)."""
        self._engine_key = None

    def _apply(self, fn, *a, **k):
        self._engine_key = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._engine_key = None
        return super()._load_from_state_dict(*a, **k)

    def engine(self):
        p0 = next(self.parameters())
        if p0.device.type != "cuda":
            raise RuntimeError("stabletts_amd: the vocoder runs only on a HIP device (move the module with .to('cuda')); "
                               "there is no CPU fallback")
        dev = p0.device.index if p0.device.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device != dev or self._engine.operand_dtype != self.operand_dtype:
            if self._engine is not None:
                self._engine.close()
            self._engine = _lib.Engine(0, 0, 0, 0, 0, 0, 0, self.operand_dtype, dev, vocoder=self.cfg)
            self._engine_key = None
        key = (_param_key(self), self.head.istft.window.data_ptr(), self.head.istft.window._version)
        if key != self._engine_key:
            with torch.no_grad():
                torch.cuda.synchronize(dev)
                self._engine.load_state_dict(self.state_dict())
            self._engine_key = key
        return self._engine

    def forward(self, x):
        """mel (B, input_channels, T) -> audio (B, T * hop_length)  (model.py:17-20)."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("the native vocoder is inference-only (SURVEY 8f-4): it has no backward, so a mel or "
                                      "parameters that require grad would silently train nothing; run it under torch.no_grad()")
        dev = next(self.parameters()).device
        if x.device != dev:
            raise ValueError(f"mel is on {x.device}, the vocoder's parameters are on {dev}")
        if x.dim() != 3 or x.shape[1] != self.cfg["input_channels"]:
            raise ValueError("mel must be (B, input_channels, T)")
        with torch.no_grad():
            eng = self.engine()
            mel = x.detach().to(torch.float32).contiguous()
            B, _, T = mel.shape
            audio = torch.empty(B, T * self.cfg["hop_length"], device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                eng.vocos_forward(mel, audio, torch.cuda.current_stream(dev).cuda_stream)
            return audio
