"""Training path of the native estimator: a ``torch.autograd.Function`` around ``st_train_forward`` (native forward
that keeps the activations in the engine) and ``st_train_backward`` (native backward: dgrad / wgrad through the
implicit-GEMM kernel, flash-attention backward, LayerNorm / FiLM / adaLN / SiLU / dropout backward).

The Function takes every estimator parameter as an input, so autograd routes their gradients like those of any
module: ``loss.backward()`` fills ``p.grad``, DDP's reducer hooks fire, optimizers need no changes
(reference: train.py:49-51,78-81).  Dropout (reference p_dropout, train mode only) is counter-based: a fresh 63-bit
seed is drawn from torch's CPU generator per forward, so ``torch.manual_seed`` makes a run reproducible.
"""
import torch


class _EstimatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, decoder, names, t, x, mask, mu, c, *params):
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
        ctx.decoder, ctx.names = decoder, names
        ctx.shapes = (x32.shape, c32.shape)
        ctx.engine, ctx.serial = eng, eng.train_serial()     # the engine keeps the activations of ONE forward: its serial
        return out

    @staticmethod
    def backward(ctx, grad_out):
        decoder = ctx.decoder
        eng = ctx.engine
        if eng is not decoder._engine or eng.handle is None or eng.train_serial() != ctx.serial:
            raise RuntimeError(
                "stabletts_amd: this backward's activations are gone -- the engine keeps the activations of ONE "
                "grad-enabled estimator forward, and another grad-enabled forward, an optimizer step / parameter update "
                "or a device move happened since.  Call backward() before the next grad-enabled forward (for "
                "loss_a + loss_b or gradient accumulation: backward each loss separately, gradients accumulate in .grad).")
        dev = decoder.device()
        g = grad_out.detach().to(dtype=torch.float32).contiguous()
        if tuple(g.shape) != tuple(ctx.shapes[0]):
            raise RuntimeError(f"grad_out has shape {tuple(g.shape)}, the forward produced {tuple(ctx.shapes[0])}")
        need = ctx.needs_input_grad          # (decoder, names, t, x, mask, mu, c, *params)
        f32 = dict(device=dev, dtype=torch.float32)      # the kernels write fp32 whatever torch's default dtype is
        gx = torch.empty(ctx.shapes[0], **f32) if need[3] else None
        gmu = torch.empty(ctx.shapes[0], **f32) if need[5] else None
        gc = torch.empty(ctx.shapes[1], **f32) if need[6] else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            eng.train_backward(ctx.serial, g, gx, gmu, gc, stream)     # the C ABI re-checks serial, B, T (ST_ERR_STATE)
            # all 116 parameter gradients in one device copy; each parameter's gradient is a view into it
            lay = eng.grad_layout()
            flat = torch.empty(lay[None], **f32)
            eng.param_grads_flat(flat, stream)
            pgrads = []
            for name, p, nd in zip(ctx.names, decoder.parameters(), need[7:]):
                if not nd:
                    pgrads.append(None)
                    continue
                off, n, _ = lay[name]
                pgrads.append(flat[off:off + n].view(p.shape))
        return (None, None, None, gx, None, gmu, gc, *pgrads)


def estimator_apply(decoder, t, x, mask, mu, c):
    """Decoder.forward under autograd (called by stabletts_amd.estimator.Decoder.forward when gradients are needed)."""
    if not torch.is_tensor(t):
        t = torch.tensor([float(t)])
    names = tuple(n for n, _ in decoder.named_parameters())
    return _EstimatorFn.apply(decoder, names, t, x, mask, mu, c, *decoder.parameters())
