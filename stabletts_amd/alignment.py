"""Duration -> alignment -> ``mu_y`` of ``StableTTS.synthesise`` (models/model.py:82-96) on native gfx950 kernels
(SURVEY.md section 8f-3): the step between the TextEncoder and the CFM decoder.

``generate_path(duration, mask)`` is a drop-in for the reference's module-level function (models/model.py:17-27);
``length_regulate(logw, x_mask, mu_x, length_scale)`` performs lines 85-95 -- ``w_ceil``, ``y_lengths``, ``y_mask``,
the alignment and ``mu_y = attn^T mu_x`` -- with the matmul replaced by the gather it is (the alignment has exactly
one 1 per mel frame).  Integer / index results are bit-exact with the reference for exactly summable durations.
There is no CPU fallback.
"""
import ctypes

import torch

from . import _lib


def _check(rc):
    if rc != _lib.ST_OK:
        raise _lib.NativeError(rc, _lib.load().st_last_error(None).decode())


def _dev(t, name):
    if t.device.type != "cuda":
        raise RuntimeError(f"stabletts_amd: {name} must be on a HIP device (there is no CPU fallback)")
    return t.device


def generate_path(duration, mask):
    """models/model.py:17-27.  duration (B, Tx) fp32, mask (B, Tx, Ty) -> path (B, Tx, Ty) of mask.dtype."""
    lib = _lib.load()
    dev = _dev(duration, "duration")
    b, t_x, t_y = mask.shape
    d = duration.detach().to(dtype=torch.float32).contiguous()
    m = mask.detach().to(device=dev, dtype=torch.float32).contiguous()
    cum = torch.empty(b, t_x, device=dev, dtype=torch.float32)
    path = torch.empty(b, t_x, t_y, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _check(lib.st_generate_path(d.data_ptr(), m.data_ptr(), b, t_x, t_y, cum.data_ptr(), path.data_ptr(),
                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return path.to(mask.dtype)


@torch.no_grad()
def length_regulate(logw, x_mask, mu_x, length_scale=1.0, return_attn=True):
    """models/model.py:85-95.  logw, x_mask (B, 1, Tx); mu_x (B, M, Tx) ->
    dict(w_ceil (B,1,Tx), y_lengths (B,) long, y_mask (B,1,Ty), attn (B,1,Tx,Ty) or None, mu_y (B,M,Ty))."""
    lib = _lib.load()
    dev = _dev(logw, "logw")
    B, _, Tx = logw.shape
    M = mu_x.shape[1]
    lw = logw.detach().to(dtype=torch.float32).contiguous()
    xm = x_mask.detach().to(device=dev, dtype=torch.float32).contiguous()
    mx = mu_x.detach().to(device=dev, dtype=torch.float32).contiguous()
    w_ceil = torch.empty(B, 1, Tx, device=dev, dtype=torch.float32)
    cum = torch.empty(B, Tx, device=dev, dtype=torch.float32)
    y_lengths = torch.empty(B, device=dev, dtype=torch.long)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _check(lib.st_durations(lw.data_ptr(), xm.data_ptr(), float(length_scale), B, Tx, w_ceil.data_ptr(), cum.data_ptr(),
                                y_lengths.data_ptr(), stream))
        Ty = int(y_lengths.max())                   # the reference synchronises here too (model.py:88)
        mu_y = torch.empty(B, M, Ty, device=dev, dtype=torch.float32)
        y_mask = torch.empty(B, 1, Ty, device=dev, dtype=torch.float32)
        attn = torch.empty(B, 1, Tx, Ty, device=dev, dtype=torch.float32) if return_attn else None
        _check(lib.st_align(cum.data_ptr(), xm.data_ptr(), y_lengths.data_ptr(), mx.data_ptr(), B, M, Tx, Ty,
                            attn.data_ptr() if attn is not None else None, mu_y.data_ptr(), y_mask.data_ptr(), stream))
    return dict(w_ceil=w_ceil, y_lengths=y_lengths, y_mask=y_mask, attn=attn, mu_y=mu_y)
