"""Training path of the native estimator: a ``torch.autograd.Function`` around ``st_train_forward`` (native forward
that keeps the activations in the engine) and ``st_train_backward`` (native backward: dgrad / wgrad through the
implicit-GEMM kernel, flash-attention backward, LayerNorm / FiLM / adaLN / SiLU / dropout backward).

The Function takes every estimator parameter as an input, so autograd routes their gradients like those of any
module: ``loss.backward()`` fills ``p.grad``, DDP's reducer hooks fire, optimizers need no changes
(reference: train.py:49-51,78-81).  Dropout (reference p_dropout, train mode only) is counter-based: a fresh 63-bit
seed is drawn from torch's CPU generator per forward, so ``torch.manual_seed`` makes a run reproducible.
"""
import torch


# Set to a list to record the order of host-side events of the training path: ("backward_part", k) when part k of a native
# backward is enqueued (tests assert that DDP launches its first bucket before the last part is enqueued).
TRACE = None


def _trace(*ev):
    if TRACE is not None:
        TRACE.append(ev)


class _Shared:
    """State shared by the three chained autograd nodes of ONE estimator call."""
    __slots__ = ("decoder", "engine", "serial", "shapes", "out", "flat", "lay", "stream_dev")


def _check_live(sh):
    decoder, eng = sh.decoder, sh.engine
    if eng is not decoder._engine or eng.handle is None or eng.train_serial() != sh.serial:
        raise RuntimeError(
            "stabletts_amd: this backward's activations are gone -- the engine keeps the activations of ONE "
            "grad-enabled estimator forward, and another grad-enabled forward, an optimizer step / parameter update "
            "or a device move happened since.  Call backward() before the next grad-enabled forward (for "
            "loss_a + loss_b or gradient accumulation: backward each loss separately, gradients accumulate in .grad).")


def _param_grads(sh, names, params, need):
    """Views of the flat gradient buffer the native backward wrote into (no copy): one storage for all parameters of this
    backward, each slice 64-byte aligned; a parameter's .grad keeps that storage alive until it is replaced."""
    out = []
    for name, p, nd in zip(names, params, need):
        if not nd:
            out.append(None)
            continue
        off, n, _ = sh.lay[name]
        out.append(sh.flat[off:off + n].view(p.shape))
    return out


class _BottomFn(torch.autograd.Function):
    """First node of the chain (in_proj, cond prenet, time MLP + the inputs): its forward runs the WHOLE native forward
    (st_train_forward keeps every activation in the engine) and hands a token to the next node; its backward is part 2."""

    @staticmethod
    def forward(ctx, sh, names, t, x, mask, mu, c, *params):
        decoder = sh.decoder
        eng = decoder.engine()
        dev = decoder.device()
        B, M, T = x.shape
        prep = decoder._prep
        t32 = prep(t.reshape(-1).to(dev), dev, "t")
        if t32.numel() == 1:
            t32 = t32.expand(B).contiguous()
        if t32.numel() != B:
            raise ValueError("t must be a scalar or have one entry per batch item")
        x32, mu32, c32, m32 = prep(x, dev, "x"), prep(mu, dev, "mu"), prep(c, dev, "c"), prep(mask, dev, "mask")
        if mu32.shape != x32.shape or m32.shape != (B, 1, T) or c32.shape != (B, decoder.gin_channels):
            raise ValueError("shape mismatch: x/mu (B,M,T), mask (B,1,T), c (B,gin)")
        out = torch.empty_like(x32)
        p_drop = float(decoder.p_dropout) if decoder.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0.0 else 0
        with torch.cuda.device(dev):
            eng.train_forward(t32, x32, mu32, m32, c32, out, p_drop, seed, torch.cuda.current_stream(dev).cuda_stream)
        sh.engine, sh.serial = eng, eng.train_serial()     # the engine keeps the activations of ONE forward: its serial
        sh.shapes = (x32.shape, c32.shape)
        sh.out, sh.flat, sh.lay, sh.stream_dev = out, None, eng.grad_layout(), dev
        ctx.sh, ctx.names, ctx.params = sh, names, params
        return torch.zeros((), device=dev, dtype=torch.float32)

    @staticmethod
    def backward(ctx, _grad_token):
        sh = ctx.sh
        _check_live(sh)
        dev = sh.stream_dev
        need = ctx.needs_input_grad          # (sh, names, t, x, mask, mu, c, *params)
        f32 = dict(device=dev, dtype=torch.float32)      # the kernels write fp32 whatever torch's default dtype is
        gx = torch.empty(sh.shapes[0], **f32) if need[3] else None
        gmu = torch.empty(sh.shapes[0], **f32) if need[5] else None
        gc = torch.empty(sh.shapes[1], **f32) if need[6] else None
        B, _, T = sh.shapes[0]
        with torch.cuda.device(dev):
            _trace("backward_part", 2)
            sh.engine.train_backward_part(sh.serial, 2, B, T, None, None, gx, gmu, gc, torch.cuda.current_stream(dev).cuda_stream)
        pg = _param_grads(sh, ctx.names, ctx.params, need[7:])
        sh.flat = None
        return (None, None, None, gx, None, gmu, gc, *pg)


class _MidFn(torch.autograd.Function):
    """Blocks 0 .. L/2-1: forward passes the token on, backward is part 1."""

    @staticmethod
    def forward(ctx, sh, names, token, *params):
        ctx.sh, ctx.names, ctx.params = sh, names, params
        return token.clone()

    @staticmethod
    def backward(ctx, _grad_token):
        sh = ctx.sh
        _check_live(sh)
        dev = sh.stream_dev
        B, _, T = sh.shapes[0]
        with torch.cuda.device(dev):
            _trace("backward_part", 1)
            sh.engine.train_backward_part(sh.serial, 1, B, T, None, None, None, None, None, torch.cuda.current_stream(dev).cuda_stream)
        return (None, None, torch.zeros((), device=dev), *_param_grads(sh, ctx.names, ctx.params, ctx.needs_input_grad[3:]))


class _TopFn(torch.autograd.Function):
    """final_proj, blocks L/2 .. L-1, the long-skip convs: forward returns the estimator output, backward is part 0 (it also
    allocates the flat buffer every parameter gradient of this backward is written into)."""

    @staticmethod
    def forward(ctx, sh, names, token, *params):
        ctx.sh, ctx.names, ctx.params = sh, names, params
        out, sh.out = sh.out, None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        sh = ctx.sh
        _check_live(sh)
        dev = sh.stream_dev
        g = grad_out.detach().to(dtype=torch.float32).contiguous()
        if tuple(g.shape) != tuple(sh.shapes[0]):
            raise RuntimeError(f"grad_out has shape {tuple(g.shape)}, the forward produced {tuple(sh.shapes[0])}")
        B, _, T = sh.shapes[0]
        with torch.cuda.device(dev):
            # (zeros, not empty: the 64-byte alignment gaps between the slices are never written)
            sh.flat = torch.zeros(sh.lay[None], device=dev, dtype=torch.float32)
            _trace("backward_part", 0)
            sh.engine.train_backward_part(sh.serial, 0, B, T, g, sh.flat, None, None, None, torch.cuda.current_stream(dev).cuda_stream)
        return (None, None, torch.zeros((), device=dev), *_param_grads(sh, ctx.names, ctx.params, ctx.needs_input_grad[3:]))


def estimator_apply(decoder, t, x, mask, mu, c):
    """Decoder.forward under autograd (called by stabletts_amd.estimator.Decoder.forward when gradients are needed).

    Three chained autograd nodes around ONE native forward: their backwards are the three parts of the native backward
    (st_train_backward_part), so the parameter gradients reach autograd -- and DDP's reducer hooks -- in three waves while the
    later parts are still being enqueued / computed, and every gradient is a view of one flat buffer the kernels wrote
    directly (no staging copy)."""
    if not torch.is_tensor(t):
        t = torch.tensor([float(t)])
    eng = decoder.engine()
    named = list(decoder.named_parameters())
    groups = {0: ([], []), 1: ([], []), 2: ([], [])}
    for n, p in named:
        k = eng.param_part(n)
        groups[k][0].append(n); groups[k][1].append(p)
    sh = _Shared()
    sh.decoder = decoder
    tok = _BottomFn.apply(sh, tuple(groups[2][0]), t, x, mask, mu, c, *groups[2][1])
    tok = _MidFn.apply(sh, tuple(groups[1][0]), tok, *groups[1][1])
    return _TopFn.apply(sh, tuple(groups[0][0]), tok, *groups[0][1])


# ---------------------------------------------------------------- compute_loss's own arithmetic (models/flow_matching.py:86-100)
def _check(rc):
    from . import _lib
    if rc != _lib.ST_OK:
        raise _lib.NativeError(rc, _lib.load().st_last_error(None).decode())


def cfm_loss_prep(x1, z, t_rand, sigma_min):
    """t = 1 - cos(t_rand pi / 2), y = (1 - (1 - sigma) t) z + t x1, u = x1 - (1 - sigma) z in ONE native kernel
    (st_cfm_loss_prep).  x1, z: (B, M, T) fp32 on the HIP device, t_rand: B values.  Returns t (B), y, u."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    dev = x1.device
    B, M, T = x1.shape
    x1c, zc = x1.detach().float().contiguous(), z.detach().float().contiguous()
    tr = t_rand.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    if tr.numel() != B or zc.shape != x1c.shape:
        raise ValueError("shape mismatch: x1 / z (B, M, T), t_rand B values")
    t = torch.empty(B, device=dev, dtype=torch.float32)
    y, u = torch.empty_like(x1c), torch.empty_like(x1c)
    with torch.cuda.device(dev):
        _check(lib.st_cfm_loss_prep(x1c.data_ptr(), zc.data_ptr(), tr.data_ptr(), float(sigma_min), B, M, T, t.data_ptr(), y.data_ptr(),
                                    u.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return t, y, u


class _CfmLossFn(torch.autograd.Function):
    """sum((pred - u)^2) / (sum(mask) * n_feats) (:100) natively, forward and backward (u is data: no gradient)."""

    @staticmethod
    def forward(ctx, pred, u, mask):
        import ctypes
        from . import _lib
        lib = _lib.load()
        dev = pred.device
        B, M, T = pred.shape
        p32, u32 = pred.detach().float().contiguous(), u.detach().float().contiguous()
        m32 = mask.detach().to(device=dev, dtype=torch.float32).contiguous()
        if u32.shape != p32.shape or m32.numel() != B * T:
            raise ValueError("shape mismatch: pred / u (B, M, T), mask (B, 1, T)")
        scratch = torch.empty(lib.st_cfm_loss_scratch_floats(), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _check(lib.st_cfm_loss(p32.data_ptr(), u32.data_ptr(), m32.data_ptr(), B, M, T, scratch.data_ptr(), loss.data_ptr(),
                                   ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        ctx.save_for_backward(p32, u32, scratch)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        import ctypes
        from . import _lib
        lib = _lib.load()
        p32, u32, scratch = ctx.saved_tensors
        dev = p32.device
        B, M, T = p32.shape
        g = grad_loss.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        gp = torch.empty_like(p32)
        with torch.cuda.device(dev):
            _check(lib.st_cfm_loss_backward(p32.data_ptr(), u32.data_ptr(), scratch.data_ptr(), g.data_ptr(), B, M, T, gp.data_ptr(),
                                            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return gp, None, None


def cfm_loss(pred, u, mask):
    return _CfmLossFn.apply(pred, u, mask)
