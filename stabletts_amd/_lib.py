"""ctypes binding of libstabletts_hip.so (C ABI: include/stabletts_hip.h).

The library is built in-tree by ``python -m stabletts_amd.build`` (hipcc, gfx950).  There is
NO fallback: if the shared library is missing or fails to load, importing the native path raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STABLETTS_HIP_LIB") or os.path.join(HERE, "libstabletts_hip.so")   # env: developer A/B builds

ST_OK = 0
ST_ERR_INVALID, ST_ERR_HIP, ST_ERR_STATE, ST_ERR_UNSUPPORTED = -1, -2, -3, -4
ST_OPERAND_BF16, ST_OPERAND_F16 = 0, 1
ST_SOLVER_EULER, ST_SOLVER_MIDPOINT, ST_SOLVER_RK4, ST_SOLVER_DOPRI5 = 0, 1, 2, 3
ST_SOLVER_BOSH3, ST_SOLVER_FEHLBERG2, ST_SOLVER_ADAPTIVE_HEUN, ST_SOLVER_IMPLICIT_ADAMS = 4, 5, 6, 7
OPERAND_DTYPES = {"bf16": ST_OPERAND_BF16, "f16": ST_OPERAND_F16, "fp16": ST_OPERAND_F16}
# None is torchdiffeq's default method = dopri5 (models/flow_matching.py:54)
SOLVERS = {"euler": ST_SOLVER_EULER, "midpoint": ST_SOLVER_MIDPOINT, "rk4": ST_SOLVER_RK4,
           "dopri5": ST_SOLVER_DOPRI5, None: ST_SOLVER_DOPRI5, "bosh3": ST_SOLVER_BOSH3,
           "fehlberg2": ST_SOLVER_FEHLBERG2, "adaptive_heun": ST_SOLVER_ADAPTIVE_HEUN,
           "implicit_adams": ST_SOLVER_IMPLICIT_ADAMS}      # = every method the reference's webui.py:110 offers

# every symbol include/stabletts_hip.h declares
EXPORTS = [
    "st_abi_version", "st_create", "st_destroy", "st_last_error", "st_load_param", "st_num_params",
    "st_finalize", "st_bind_param", "st_repack", "st_train_serial", "st_estimator_forward", "st_cfm_solve", "st_output_status", "st_last_solve_stats", "st_debug_capture", "st_debug_fetch",
    "st_create_text_encoder", "st_text_encoder_forward", "st_param_info",
    "st_profile_enable", "st_profile_select", "st_profile_stride", "st_profile_num_classes", "st_profile_class_name", "st_profile_read",
    "st_device_bytes", "st_train_forward", "st_train_backward", "st_train_backward_part", "st_train_param_part", "st_train_grad_offset",
    "st_train_grad_numel", "st_param_grad", "st_param_grads_flat",
    "st_durations", "st_generate_path", "st_align", "st_create_vocoder", "st_vocos_forward",
    "st_cfm_loss_prep", "st_cfm_loss", "st_cfm_loss_backward", "st_cfm_loss_scratch_floats",
    "st_set_option", "st_get_option", "st_attention_stats",
]


class StConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "noise_channels", "hidden_channels", "filter_channels", "n_heads", "n_layers",
        "kernel_size", "gin_channels", "operand_dtype")]


class StVocosConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "input_channels", "dim", "intermediate_dim", "num_layers", "n_fft", "hop_length", "operand_dtype")]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libstabletts_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Loads the shared library (once).  torch is imported first so that the HIP runtime the
    library binds to (SONAME libamdhip64.so.7) is the one torch already loaded: a process must
    not hold two HIP runtimes, streams and device pointers are shared across the boundary."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so when the wheel bundles it)
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tl):
            ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -m stabletts_amd.build` "
                          "(the native HIP path has no fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    c_void_p, c_int, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.st_abi_version.restype = c_int
    lib.st_create.argtypes = [ctypes.POINTER(StConfig), c_int, ctypes.POINTER(c_void_p)]
    lib.st_create.restype = c_int
    lib.st_destroy.argtypes = [c_void_p]
    lib.st_destroy.restype = None
    lib.st_last_error.argtypes = [c_void_p]
    lib.st_last_error.restype = ctypes.c_char_p
    lib.st_load_param.argtypes = [c_void_p, ctypes.c_char_p, c_void_p, ctypes.POINTER(ctypes.c_int64), c_int]
    lib.st_load_param.restype = c_int
    lib.st_num_params.argtypes = [c_void_p]
    lib.st_num_params.restype = c_int
    lib.st_finalize.argtypes = [c_void_p]
    lib.st_finalize.restype = c_int
    lib.st_bind_param.argtypes = [c_void_p, ctypes.c_char_p, c_void_p, ctypes.POINTER(ctypes.c_int64), c_int]
    lib.st_bind_param.restype = c_int
    lib.st_repack.argtypes = [c_void_p, c_void_p]
    lib.st_repack.restype = c_int
    lib.st_train_serial.argtypes = [c_void_p]
    lib.st_train_serial.restype = ctypes.c_int64
    lib.st_estimator_forward.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int, c_int, c_void_p]
    lib.st_estimator_forward.restype = c_int
    lib.st_cfm_solve.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                 c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.st_cfm_solve.restype = c_int
    lib.st_output_status.argtypes = [c_void_p, c_void_p, ctypes.POINTER(c_int)]
    lib.st_output_status.restype = c_int
    lib.st_last_solve_stats.argtypes = [c_void_p] + [ctypes.POINTER(ctypes.c_int64)] * 3
    lib.st_last_solve_stats.restype = c_int
    lib.st_debug_capture.argtypes = [c_void_p, c_int]
    lib.st_debug_capture.restype = c_int
    lib.st_debug_fetch.argtypes = [c_void_p, ctypes.c_char_p, c_void_p, ctypes.c_int64]
    lib.st_debug_fetch.restype = ctypes.c_int64
    lib.st_profile_enable.argtypes = [c_void_p, c_int]
    lib.st_profile_enable.restype = c_int
    lib.st_profile_select.argtypes = [c_void_p, ctypes.c_uint64]
    lib.st_profile_select.restype = c_int
    lib.st_param_info.argtypes = [c_void_p, c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64)]
    lib.st_param_info.restype = c_int
    lib.st_create_text_encoder.argtypes = [ctypes.POINTER(StConfig), c_int, c_int, ctypes.POINTER(c_void_p)]
    lib.st_create_text_encoder.restype = c_int
    lib.st_text_encoder_forward.argtypes = [c_void_p] + [c_void_p] * 6 + [c_int, c_int, c_void_p]
    lib.st_text_encoder_forward.restype = c_int
    lib.st_profile_stride.argtypes = [c_void_p, c_int]
    lib.st_profile_stride.restype = c_int
    lib.st_profile_num_classes.restype = c_int
    lib.st_profile_class_name.argtypes = [c_int]
    lib.st_profile_class_name.restype = ctypes.c_char_p
    lib.st_profile_read.argtypes = [c_void_p, c_int, ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.st_profile_read.restype = c_int
    lib.st_device_bytes.argtypes = [c_void_p]
    lib.st_device_bytes.restype = ctypes.c_int64
    lib.st_train_forward.argtypes = [c_void_p] + [c_void_p] * 6 + [c_int, c_int, c_float, ctypes.c_uint64, c_void_p]
    lib.st_train_forward.restype = c_int
    lib.st_train_backward.argtypes = [c_void_p, ctypes.c_int64, c_int, c_int] + [c_void_p] * 4 + [c_void_p]
    lib.st_train_backward.restype = c_int
    lib.st_train_backward_part.argtypes = [c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.c_int64] + [c_void_p] * 3 + [c_void_p]
    lib.st_train_backward_part.restype = c_int
    lib.st_train_param_part.argtypes = [c_void_p, ctypes.c_char_p]
    lib.st_train_param_part.restype = c_int
    lib.st_train_grad_offset.argtypes = [c_void_p, ctypes.c_char_p]
    lib.st_train_grad_offset.restype = ctypes.c_int64
    lib.st_train_grad_numel.argtypes = [c_void_p]
    lib.st_train_grad_numel.restype = ctypes.c_int64
    lib.st_param_grad.argtypes = [c_void_p, ctypes.c_char_p, c_void_p, ctypes.c_int64, c_void_p]
    lib.st_param_grad.restype = c_int
    lib.st_param_grads_flat.argtypes = [c_void_p, c_void_p, ctypes.c_int64, c_void_p]
    lib.st_param_grads_flat.restype = c_int
    lib.st_durations.argtypes = [c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.st_durations.restype = c_int
    lib.st_generate_path.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.st_generate_path.restype = c_int
    lib.st_align.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.st_align.restype = c_int
    lib.st_cfm_loss_prep.argtypes = [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.st_cfm_loss_prep.restype = c_int
    lib.st_cfm_loss.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.st_cfm_loss.restype = c_int
    lib.st_cfm_loss_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.st_cfm_loss_backward.restype = c_int
    lib.st_cfm_loss_scratch_floats.argtypes = []
    lib.st_cfm_loss_scratch_floats.restype = c_int
    lib.st_create_vocoder.argtypes = [ctypes.POINTER(StVocosConfig), c_int, ctypes.POINTER(c_void_p)]
    lib.st_create_vocoder.restype = c_int
    lib.st_vocos_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.st_vocos_forward.restype = c_int
    lib.st_set_option.argtypes = [c_void_p, ctypes.c_char_p, c_int]
    lib.st_set_option.restype = c_int
    lib.st_get_option.argtypes = [c_void_p, ctypes.c_char_p, ctypes.POINTER(c_int)]
    lib.st_get_option.restype = c_int
    lib.st_attention_stats.argtypes = [c_void_p, c_void_p, ctypes.POINTER(c_float)]
    lib.st_attention_stats.restype = c_int
    if lib.st_abi_version() != 4:
        raise ImportError("libstabletts_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


class Engine:
    """Thin owner of one ``st_engine`` handle."""

    def __init__(self, noise_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size,
                 gin_channels, operand_dtype="f16", device=0, text_encoder_vocab=None, vocoder=None):
        """text_encoder_vocab: None -> CFM decoder estimator (st_create); n_vocab -> TextEncoder handle
        (st_create_text_encoder; noise_channels is then the encoder's out_channels).
        vocoder: dict(input_channels, dim, intermediate_dim, num_layers, n_fft, hop_length) -> Vocos handle
        (st_create_vocoder; the decoder arguments are ignored)."""
        self.lib = load()
        if operand_dtype not in OPERAND_DTYPES:
            raise ValueError(f"operand_dtype must be one of {sorted(OPERAND_DTYPES)}")
        self.operand_dtype = operand_dtype
        cfg = StConfig(noise_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size,
                       gin_channels, OPERAND_DTYPES[operand_dtype])
        h = ctypes.c_void_p()
        if vocoder is not None:
            vc = StVocosConfig(vocoder["input_channels"], vocoder["dim"], vocoder["intermediate_dim"], vocoder["num_layers"],
                               vocoder["n_fft"], vocoder["hop_length"], OPERAND_DTYPES[operand_dtype])
            rc = self.lib.st_create_vocoder(ctypes.byref(vc), int(device), ctypes.byref(h))
        elif text_encoder_vocab is None:
            rc = self.lib.st_create(ctypes.byref(cfg), int(device), ctypes.byref(h))
        else:
            rc = self.lib.st_create_text_encoder(ctypes.byref(cfg), int(text_encoder_vocab), int(device), ctypes.byref(h))
        if rc != ST_OK:
            raise NativeError(rc, self.lib.st_last_error(None).decode())
        self.handle = h
        self.device = int(device)

    def _check(self, rc):
        if rc != ST_OK:
            raise NativeError(rc, self.lib.st_last_error(self.handle).decode())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.st_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_params(self):
        return self.lib.st_num_params(self.handle)

    def param_info(self):
        """[(reference state_dict name, shape)] the handle expects, in the engine's (name) order."""
        out = []
        for i in range(self.num_params()):
            name = ctypes.c_char_p()
            shape = (ctypes.c_int64 * 4)()
            nd = self.lib.st_param_info(self.handle, i, ctypes.byref(name), shape)
            if nd < 0:
                raise NativeError(nd, "st_param_info")
            out.append((name.value.decode(), tuple(shape[:nd])))
        return out

    def load_state_dict(self, sd):
        """sd: name -> torch.Tensor (fp32, any device), reference ``decoder.estimator.*`` names."""
        for name, t in sd.items():
            t = t.detach().float().contiguous()
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            self._check(self.lib.st_load_param(self.handle, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()))
        self._check(self.lib.st_finalize(self.handle))

    def bind_parameters(self, named_params):
        """named_params: iterable of (reference name, fp32 contiguous tensor ON THE ENGINE'S DEVICE).  The engine reads
        the tensors in place from now on (st_bind_param: no copy; the caller keeps them alive) and packs its 16-bit
        operand copies (st_finalize)."""
        for name, t in named_params:
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            self._check(self.lib.st_bind_param(self.handle, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()))
        self._check(self.lib.st_finalize(self.handle))

    def repack(self, stream):
        """After an in-place update of the bound tensors: re-pack the 16-bit copies as kernels on ``stream`` (no sync)."""
        self._check(self.lib.st_repack(self.handle, ctypes.c_void_p(stream)))

    def estimator_forward(self, t, x, mu, mask, c, out, stream):
        B, _, T = x.shape
        self._check(self.lib.st_estimator_forward(self.handle, t.data_ptr(), int(t.numel()), x.data_ptr(), mu.data_ptr(),
                                                  mask.data_ptr(), c.data_ptr(), out.data_ptr(), B, T,
                                                  ctypes.c_void_p(stream)))

    def cfm_solve(self, mu, mask, z, c, n_steps, solver, use_cfg, cfg_strength, fake_speaker, fake_content, out, stream):
        B, _, T = mu.shape
        self._check(self.lib.st_cfm_solve(self.handle, mu.data_ptr(), mask.data_ptr(), z.data_ptr(), c.data_ptr(),
                                          int(n_steps), int(solver), int(bool(use_cfg)), float(cfg_strength),
                                          fake_speaker.data_ptr() if fake_speaker is not None else None,
                                          fake_content.data_ptr() if fake_content is not None else None,
                                          out.data_ptr(), B, T, ctypes.c_void_p(stream)))

    def text_encoder_forward(self, tokens, lengths, c, x_out, mu_out, mask_out, stream):
        B, T = tokens.shape
        self._check(self.lib.st_text_encoder_forward(self.handle, tokens.data_ptr(), lengths.data_ptr(), c.data_ptr(),
                                                     x_out.data_ptr(), mu_out.data_ptr(), mask_out.data_ptr(), B, T,
                                                     ctypes.c_void_p(stream)))

    def vocos_forward(self, mel, audio, stream):
        B, _, T = mel.shape
        self._check(self.lib.st_vocos_forward(self.handle, mel.data_ptr(), audio.data_ptr(), B, T, ctypes.c_void_p(stream)))

    # ---- training: forward that keeps activations + backward (include/stabletts_hip.h, "training")
    def train_forward(self, t, x, mu, mask, c, out, p_dropout, seed, stream):
        B, _, T = x.shape
        self._check(self.lib.st_train_forward(self.handle, t.data_ptr(), x.data_ptr(), mu.data_ptr(), mask.data_ptr(),
                                              c.data_ptr(), out.data_ptr(), B, T, float(p_dropout), int(seed),
                                              ctypes.c_void_p(stream)))

    def train_serial(self):
        """Serial of the grad-enabled forward whose activations the engine holds (0: none)."""
        return int(self.lib.st_train_serial(self.handle))

    def train_backward(self, serial, grad_out, grad_x, grad_mu, grad_c, stream):
        ptr = lambda v: v.data_ptr() if v is not None else None      # noqa: E731
        B, _, T = grad_out.shape
        self._check(self.lib.st_train_backward(self.handle, int(serial), B, T, grad_out.data_ptr(), ptr(grad_x), ptr(grad_mu),
                                               ptr(grad_c), ctypes.c_void_p(stream)))

    def train_backward_part(self, serial, part, B, T, grad_out, grad_flat, grad_x, grad_mu, grad_c, stream):
        """One of the three parts of a backward (0: final_proj + upper blocks + long-skip convs, 1: lower blocks, 2: in_proj /
        prenet / time MLP + input gradients); grad_flat (part 0) receives every parameter gradient directly (grad_layout())."""
        ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
        self._check(self.lib.st_train_backward_part(self.handle, int(serial), B, T, int(part), ptr(grad_out), ptr(grad_flat),
                                                    grad_flat.numel() if grad_flat is not None else 0, ptr(grad_x), ptr(grad_mu),
                                                    ptr(grad_c), ctypes.c_void_p(stream)))

    def param_part(self, name):
        """Backward part (0, 1, 2) that produces the gradient of parameter `name` (a function of the name only: cached)."""
        cache = self.__dict__.setdefault("_param_part", {})
        r = cache.get(name)
        if r is None:
            r = self.lib.st_train_param_part(self.handle, name.encode())
            if r < 0:
                raise NativeError(r, f"st_train_param_part({name})")
            cache[name] = r
        return r

    def param_grad(self, name, dst, stream):
        self._check(self.lib.st_param_grad(self.handle, name.encode(), dst.data_ptr(), dst.numel(), ctypes.c_void_p(stream)))

    def param_grads_flat(self, dst, stream):
        """Every parameter gradient in one copy: dst (fp32, grad_layout()[None] elements) in the flat layout."""
        self._check(self.lib.st_param_grads_flat(self.handle, dst.data_ptr(), dst.numel(), ctypes.c_void_p(stream)))

    def grad_layout(self):
        """{reference name: (offset, numel, shape)} of param_grads_flat's layout (cached)."""
        lay = getattr(self, "_grad_layout", None)
        if lay is None:
            lay = {}
            for name, shape in self.param_info():
                n = 1
                for d in shape:
                    n *= d
                off = self.lib.st_train_grad_offset(self.handle, name.encode())
                if off < 0:
                    raise NativeError(int(off), f"st_train_grad_offset({name})")
                lay[name] = (int(off), n, shape)       # slices start on 64-byte boundaries
            lay[None] = int(self.lib.st_train_grad_numel(self.handle))
            self._grad_layout = lay
        return lay

    def output_nonfinite(self, stream):
        """True when a call completed since the last query wrote NaN / Inf to its output (synchronises ``stream``)."""
        flag = ctypes.c_int(0)
        self._check(self.lib.st_output_status(self.handle, ctypes.c_void_p(stream), ctypes.byref(flag)))
        return bool(flag.value)

    def last_solve_stats(self):
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.st_last_solve_stats(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(nfe=a.value, steps=b.value, rejects=c.value)

    # ---- test / measurement hooks
    def debug_capture(self, on):
        self._check(self.lib.st_debug_capture(self.handle, int(on)))

    def debug_fetch(self, name):
        import numpy as np
        n = self.lib.st_debug_fetch(self.handle, name.encode(), None, 0)
        if n < 0:
            raise NativeError(int(n), self.lib.st_last_error(self.handle).decode())
        buf = np.empty(int(n), dtype=np.float32)
        r = self.lib.st_debug_fetch(self.handle, name.encode(), buf.ctypes.data_as(ctypes.c_void_p), int(n))
        if r < 0:
            raise NativeError(int(r), self.lib.st_last_error(self.handle).decode())
        return buf

    def profile_enable(self, on, classes=None, stride=1):
        """classes: optional iterable of class names to restrict event recording to; stride: record every
        stride-th launch of a class only (event pairs cost ~10 us of stream time each)."""
        mask = (1 << 64) - 1
        if classes is not None:
            names = [self.lib.st_profile_class_name(i).decode() for i in range(self.lib.st_profile_num_classes())]
            mask = 0
            for c in classes:
                mask |= 1 << names.index(c)
        self._check(self.lib.st_profile_select(self.handle, mask))
        self._check(self.lib.st_profile_stride(self.handle, int(stride)))
        self._check(self.lib.st_profile_enable(self.handle, int(on)))

    def profile_read(self):
        """-> {class_name: dict(launches, total_ms, flops_per_launch)} since the last read."""
        out = {}
        for i in range(self.lib.st_profile_num_classes()):
            n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            self._check(self.lib.st_profile_read(self.handle, i, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
            out[self.lib.st_profile_class_name(i).decode()] = dict(
                launches=n.value, total_ms=ms.value, flops_per_launch=fl.value)
        return out

    def device_bytes(self):
        return self.lib.st_device_bytes(self.handle)

    def set_option(self, name, value):
        """st_set_option: 'attention_precision' 0 (16-bit q / k operands) / 1 (split hi + lo operands)."""
        self._check(self.lib.st_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_int()
        self._check(self.lib.st_get_option(self.handle, name.encode(), ctypes.byref(v)))
        return v.value

    def attention_stats(self, stream):
        """Largest log-sum-exp (natural units) of any valid attention row since the last query (synchronises the stream; -inf: none)."""
        v = ctypes.c_float()
        self._check(self.lib.st_attention_stats(self.handle, ctypes.c_void_p(stream), ctypes.byref(v)))
        return float(v.value)
