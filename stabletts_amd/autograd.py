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
        ctx.engine_key = decoder._engine_key
        return out

    @staticmethod
    def backward(ctx, grad_out):
        decoder = ctx.decoder
        eng = decoder._engine
        if eng is None or decoder._engine_key != ctx.engine_key:
            raise RuntimeError("the estimator's parameters changed between forward and backward (the activations live in "
                               "the engine: one backward per forward, before the next optimizer step)")
        dev = decoder.device()
        g = grad_out.detach().to(dtype=torch.float32).contiguous()
        need = ctx.needs_input_grad          # (decoder, names, t, x, mask, mu, c, *params)
        gx = torch.empty(ctx.shapes[0], device=dev) if need[3] else None
        gmu = torch.empty(ctx.shapes[0], device=dev) if need[5] else None
        gc = torch.empty(ctx.shapes[1], device=dev) if need[6] else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            eng.train_backward(g, gx, gmu, gc, stream)
            pgrads = []
            for name, p, nd in zip(ctx.names, decoder.parameters(), need[7:]):
                if not nd:
                    pgrads.append(None)
                    continue
                gp = torch.empty(p.shape, device=dev, dtype=torch.float32)
                eng.param_grad(name, gp, stream)
                pgrads.append(gp)
        return (None, None, None, gx, None, gmu, gc, *pgrads)


def estimator_apply(decoder, t, x, mask, mu, c):
    """Decoder.forward under autograd (called by stabletts_amd.estimator.Decoder.forward when gradients are needed)."""
    if not torch.is_tensor(t):
        t = torch.tensor([float(t)])
    names = tuple(n for n, _ in decoder.named_parameters())
    return _EstimatorFn.apply(decoder, names, t, x, mask, mu, c, *decoder.parameters())
