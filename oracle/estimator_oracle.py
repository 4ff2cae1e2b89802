"""fp32 PyTorch-CPU restatement of the StableTTS CFM decoder (test infrastructure).

Every function cites the reference lines (relative to /root/reference) it follows.
Pinned against the real reference modules by oracle/make_golden.py -> tests/golden.
NOT part of the product: stabletts_amd/ never imports this.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- time embedding
def sinusoidal_pos_emb(t, dim=256, scale=1000.0):
    """models/estimator.py:41-49 (SinusoidalPosEmb.forward). t: () or (B,) -> (1|B, dim)."""
    if t.ndim < 1:
        t = t.unsqueeze(0)
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb)
    emb = scale * t.unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def time_mlp(sd, emb):
    """models/estimator.py:55-62 (Linear, SiLU, Linear)."""
    h = F.linear(emb, sd["time_mlp.layer.0.weight"], sd["time_mlp.layer.0.bias"])
    h = F.silu(h)
    return F.linear(h, sd["time_mlp.layer.2.weight"], sd["time_mlp.layer.2.bias"])


# ----------------------------------------------------------------------------- conditioning prenet
def cond_proj(sd, mu, k=3):
    """models/estimator.py:83-89,118: conv3, SiLU, conv3, SiLU, conv3 on the UNMASKED mu."""
    p = k // 2
    h = F.silu(F.conv1d(mu, sd["cond_proj.0.weight"], sd["cond_proj.0.bias"], padding=p))
    h = F.silu(F.conv1d(h, sd["cond_proj.2.weight"], sd["cond_proj.2.bias"], padding=p))
    return F.conv1d(h, sd["cond_proj.4.weight"], sd["cond_proj.4.bias"], padding=p)


# ----------------------------------------------------------------------------- RoPE / attention / FFN
def rope(x, d=32, base=10000):
    """models/diffusion_transformer.py:145-198 (RotaryPositionalEmbeddings), x: (B,H,T,Dh).

    Partial rotary on the first d features; pairs (j, j+d/2); theta_j = base^(-2j/d);
    position = frame index.
    """
    T = x.shape[2]
    theta = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    idx_theta = torch.einsum("n,d->nd", torch.arange(T).float(), theta)
    idx_theta2 = torch.cat([idx_theta, idx_theta], dim=1)          # (T, d)
    cos = idx_theta2.cos()[None, None]
    sin = idx_theta2.sin()[None, None]
    x_rope, x_pass = x[..., :d], x[..., d:]
    d2 = d // 2
    neg_half = torch.cat([-x_rope[..., d2:], x_rope[..., :d2]], dim=-1)
    x_rope = x_rope * cos + neg_half * sin
    return torch.cat((x_rope, x_pass), dim=-1)


def attention(q, k, v, mask, n_heads=4, drop=None, subst=None):
    """models/diffusion_transformer.py:67-79 + mask construction :107-108.

    q,k,v: (B, C, T) conv outputs; mask (B,1,T) float 0/1. Returns (B, C, T) and the
    post-RoPE per-head tensors. Explicit softmax(QK^T/sqrt(Dh) + M)V; a fully masked
    query row gives a uniform softmax (no NaN), exactly like SDPA with the additive
    -finfo.max mask.
    """
    B, C, T = q.shape
    dh = C // n_heads
    qh = q.view(B, n_heads, dh, T).transpose(2, 3)
    kh = k.view(B, n_heads, dh, T).transpose(2, 3)
    vh = v.view(B, n_heads, dh, T).transpose(2, 3)
    qh = rope(qh, int(dh * 0.5))
    kh = rope(kh, int(dh * 0.5))
    if callable(subst):      # (tools/qk_rounding_sensitivity.py: a function of THIS forward's q, k, v, e.g. their f16 rounding)
        subst = subst(qh, kh, vh)
    if subst is not None:
        # test hook (tests/test_gpu_training.py): evaluate the attention AT the given post-RoPE values (the native
        # forward's own 16-bit q, k, v) while gradients keep flowing through this graph -- a straight-through
        # substitution.  d q, d k are ill-conditioned in v at random init; matching the forward operands removes
        # that amplification from an end-to-end gradient comparison.
        qh = qh + (subst["q"] - qh).detach()
        kh = kh + (subst["k"] - kh).detach()
        vh = vh + (subst["v"] - vh).detach()
    am = mask.unsqueeze(1) * mask.unsqueeze(-1)                    # (B,1,T,T)
    am = torch.zeros_like(am).masked_fill(am == 0, -torch.finfo(q.dtype).max)
    s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(dh) + am
    p = torch.softmax(s, dim=-1)
    if drop is not None:        # train-mode SDPA dropout_p (:77) with an explicit keep/(1-p) factor tensor (B,H,T,T)
        p = p * drop
    o = torch.matmul(p, vh)
    out = o.transpose(2, 3).contiguous().view(B, C, T)
    return out, (qh, kh, vh)


def mha(sd, prefix, x, mask, n_heads=4, taps=None, drop=None, subst=None):
    """models/diffusion_transformer.py:58-65 (MultiHeadAttention.forward)."""
    q = F.conv1d(x, sd[prefix + "conv_q.weight"], sd[prefix + "conv_q.bias"])
    k = F.conv1d(x, sd[prefix + "conv_k.weight"], sd[prefix + "conv_k.bias"])
    v = F.conv1d(x, sd[prefix + "conv_v.weight"], sd[prefix + "conv_v.bias"])
    a, (qh, kh, vh) = attention(q, k, v, mask, n_heads, drop, subst)
    if taps is not None:
        taps["q"], taps["k"], taps["v"], taps["attn"] = qh, kh, vh, a
    return F.conv1d(a, sd[prefix + "conv_o.weight"], sd[prefix + "conv_o.bias"])


def ffn(sd, prefix, x, mask, k=3, taps=None, drop=None):
    """models/diffusion_transformer.py:25-30 (FFN.forward); dropout is identity in eval (drop: explicit train-mode
    keep/(1-p) factor tensor (B,F,T), :28)."""
    p = k // 2
    h = F.conv1d(x * mask, sd[prefix + "conv_1.weight"], sd[prefix + "conv_1.bias"], padding=p)
    h = F.silu(h)
    if drop is not None:
        h = h * drop
    if taps is not None:
        taps["u"] = h * mask
    h = F.conv1d(h * mask, sd[prefix + "conv_2.weight"], sd[prefix + "conv_2.bias"], padding=p)
    return h * mask


def layer_norm_c(x):
    """nn.LayerNorm(C, elementwise_affine=False) on the transposed view
    (models/diffusion_transformer.py:88,90,111-112). x: (B,C,T)."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), eps=1e-5).transpose(1, 2)


def dit_block(sd, i, x, c, tau, mask, n_heads=4, k=3, taps=None, drop=None, qkv_subst=None):
    """DitWrapper.forward (models/estimator.py:15-18) + DiTConVBlock.forward
    (models/diffusion_transformer.py:98-117); arithmetic order of SURVEY.md 3.3."""
    p = f"blocks.{i}."
    film = F.conv1d(tau.unsqueeze(2), sd[p + "time_fusion.film.weight"], sd[p + "time_fusion.film.bias"])
    gamma, beta = torch.chunk(film, 2, dim=1)
    x = (gamma * x + beta) * mask
    x = x * mask
    hc = c
    if (p + "block.adaLN_modulation.0.weight") in sd:
        hc = F.linear(hc, sd[p + "block.adaLN_modulation.0.weight"], sd[p + "block.adaLN_modulation.0.bias"])
    ada = F.linear(F.silu(hc), sd[p + "block.adaLN_modulation.2.weight"], sd[p + "block.adaLN_modulation.2.bias"])
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = ada.unsqueeze(2).chunk(6, dim=1)
    if taps is not None:
        taps["x1"] = x
    h = layer_norm_c(x) * (1 + sc_a) + sh_a
    if taps is not None:
        taps["h1"] = h
    x = x + g_a * mha(sd, p + "block.attn.", h, mask, n_heads, taps, None if drop is None else drop["attn"][i],
                      None if qkv_subst is None else qkv_subst[i]) * mask
    if taps is not None:
        taps["x2"] = x
    h = layer_norm_c(x) * (1 + sc_m) + sh_m
    if taps is not None:
        taps["h2"] = h * mask
    x = x + g_m * ffn(sd, p + "block.mlp.", h, mask, k, taps, None if drop is None else drop["ffn"][i])
    if taps is not None:
        taps["x3"] = x
    return x


def dit_conv_block(sd, p, x, c, mask, n_heads=4, k=3, taps=None):
    """DiTConVBlock.forward (models/diffusion_transformer.py:98-117) with parameters under prefix ``p``
    (the text encoder's blocks, models/text_encoder.py:25,40-41: no FiLM wrapper)."""
    x = x * mask
    hc = c
    if (p + "adaLN_modulation.0.weight") in sd:
        hc = F.linear(hc, sd[p + "adaLN_modulation.0.weight"], sd[p + "adaLN_modulation.0.bias"])
    ada = F.linear(F.silu(hc), sd[p + "adaLN_modulation.2.weight"], sd[p + "adaLN_modulation.2.bias"])
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = ada.unsqueeze(2).chunk(6, dim=1)
    if taps is not None:
        taps["x1"] = x
    h = layer_norm_c(x) * (1 + sc_a) + sh_a
    if taps is not None:
        taps["h1"] = h
    x = x + g_a * mha(sd, p + "attn.", h, mask, n_heads, taps) * mask
    if taps is not None:
        taps["x2"] = x
    h = layer_norm_c(x) * (1 + sc_m) + sh_m
    x = x + g_m * ffn(sd, p + "mlp.", h, mask, k, taps)
    if taps is not None:
        taps["x3"] = x
    return x


def text_encoder_forward(sd, tokens, c, lengths, n_heads=4, k=3, taps=None):
    """models/text_encoder.py:34-44 (TextEncoder.forward): tokens (B,T) long, c (B,gin), lengths (B,) ->
    x (B,C,T), mu_x (B,out,T), x_mask (B,1,T).  sequence_mask = utils/mask.py (arange < length)."""
    C = sd["emb.weight"].shape[1]
    x = F.embedding(tokens, sd["emb.weight"]) * (C ** 0.5)
    x = x.transpose(1, -1)
    T = x.size(2)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).unsqueeze(1).to(x.dtype)
    i = 0
    while f"encoder.{i}.attn.conv_q.weight" in sd:
        bt = {} if taps is not None else None
        x = dit_conv_block(sd, f"encoder.{i}.", x, c, mask, n_heads, k, bt)
        if taps is not None:
            for kk, vv in bt.items():
                taps[f"b{i}.{kk}"] = vv
        i += 1
    mu_x = F.conv1d(x, sd["proj.weight"], sd["proj.bias"]) * mask
    return x, mu_x, mask


# ----------------------------------------------------------------------------- estimator
def decoder_forward(sd, t, x, mask, mu, c, n_heads=4, k=3, taps=None, drop=None, qkv_subst=None):
    """models/estimator.py:103-138 (Decoder.forward): one vector-field evaluation.

    t: () or (B,), x/mu: (B,M,T), mask: (B,1,T), c: (B,gin).  taps: optional dict that
    receives named intermediates (used by the GPU per-stage parity tests).
    """
    n_layers = 0
    while f"blocks.{n_layers}.time_fusion.film.weight" in sd:
        n_layers += 1
    n_lsc = n_layers // 2
    C = sd["in_proj.weight"].shape[0]
    tau = time_mlp(sd, sinusoidal_pos_emb(t, C))
    cond = cond_proj(sd, mu, k)
    x = F.conv1d(torch.cat((x, cond), dim=1), sd["in_proj.weight"], sd["in_proj.bias"])
    if taps is not None:
        taps["tau"], taps["cond"], taps["h0"] = tau, cond, x
    skips = []
    for i in range(n_layers):
        if i < n_lsc:
            skips.append(x)
        else:
            x = torch.cat((x, skips.pop()), dim=1)
            x = F.conv1d(x, sd[f"lsc_layers.{i - n_lsc}.weight"], sd[f"lsc_layers.{i - n_lsc}.bias"],
                         padding=k // 2)
            if taps is not None:
                taps[f"lsc{i - n_lsc}"] = x
        bt = {} if taps is not None else None
        x = dit_block(sd, i, x, c, tau, mask, n_heads, k, bt, drop, qkv_subst)
        if taps is not None:
            for kk, vv in bt.items():
                taps[f"b{i}.{kk}"] = vv
    out = F.conv1d(x * mask, sd["final_proj.weight"], sd["final_proj.bias"])
    return out * mask


# ----------------------------------------------------------------------------- CFM wrapper
def cfg_wrapper(sd, t, x, mask, mu, c, fake_speaker, fake_content, cfg_strength, **kw):
    """models/flow_matching.py:58-67: two estimator calls, u + s*(c - u)."""
    fs = fake_speaker.repeat(x.size(0), 1)
    fc = fake_content.repeat(x.size(0), 1, x.size(-1))
    cond_out = decoder_forward(sd, t, x, mask, mu, c, **kw)
    uncond_out = decoder_forward(sd, t, x, mask, fc, fs, **kw)
    return uncond_out + cfg_strength * (cond_out - uncond_out)


def linspace_f32(n_timesteps):
    """t_span of models/flow_matching.py:46."""
    return torch.linspace(0, 1, n_timesteps + 1)


def odeint_fixed(f, y0, t_span, method="euler"):
    """torchdiffeq 0.2.x fixed-grid solvers restated (call site models/flow_matching.py:54).

    With options=None the integration grid IS t_span.  euler / midpoint / rk4 (torchdiffeq's
    rk4 is the 3/8-rule "rk4_alt_step_func").  f receives t as a 0-dim tensor.  Returns the
    final state (the reference takes trajectory[-1], :55).  PARITY UNPINNED against
    torchdiffeq itself (absent offline); see oracle/__init__.py.
    """
    y = y0
    for i in range(len(t_span) - 1):
        t0, t1 = t_span[i], t_span[i + 1]
        dt = t1 - t0
        if method == "euler":
            y = y + dt * f(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            f0 = f(t0, y)
            y = y + dt * f(t0 + half, y + f0 * half)
        elif method == "rk4":
            k1 = f(t0, y)
            k2 = f(t0 + dt / 3, y + dt * k1 / 3)
            k3 = f(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
            k4 = f(t1, y + dt * (k1 - k2 + k3))
            y = y + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        else:
            raise ValueError(f"oracle: unsupported fixed-grid solver {method!r}")
    return y


def adams_coefficients(order):
    """Exact Adams-Bashforth / Adams-Moulton weights for a uniform grid, newest sample first:
    y1 - y0 = dt * sum_i bashforth[i] * f(t0 - i dt)            (order samples, explicit)
    y1 - y0 = dt * (moulton[0] * f(t1) + sum_i moulton[i + 1] * f(t0 - i dt))   (order samples, implicit)
    from the defining integrals of the Lagrange basis over [0, 1] (unit step), in rational arithmetic.  torchdiffeq's
    fixed_adams.py stores the same numbers as integer tables over a divisor (e.g. [55, -59, 37, -9] / 24 and
    [9, 19, -5, 1] / 24 for four samples)."""
    from fractions import Fraction

    def weights(nodes):
        out = []
        for j, xj in enumerate(nodes):
            poly = [Fraction(1)]                    # coefficients of prod_{i != j} (u - x_i), lowest degree first
            den = Fraction(1)
            for i, xi in enumerate(nodes):
                if i == j:
                    continue
                poly = [Fraction(0)] + poly
                for k in range(len(poly) - 1):
                    poly[k] -= xi * poly[k + 1]
                den *= (xj - xi)
            out.append(sum(c / (k + 1) for k, c in enumerate(poly)) / den)      # integral over [0, 1]
        return out

    bashforth = weights([Fraction(-i) for i in range(order)])          # samples at u = 0, -1, -2, ...
    moulton = weights([Fraction(1)] + [Fraction(-i) for i in range(order - 1)])
    return [float(x) for x in bashforth], [float(x) for x in moulton]


def odeint_implicit_adams(f, y0, t_span, rtol=1e-5, atol=1e-5, max_order=12, max_iters=4, stats=None):
    """torchdiffeq's 'implicit_adams' (fixed_adams.py: AdamsBashforthMoulton on the fixed grid t_span, as offered by the
    reference's webui.py:110 and called at models/flow_matching.py:54 with rtol = atol = 1e-5), restated from memory of the
    published source -- torchdiffeq is absent offline: PARITY UNPINNED.  Per grid step:
      f0 = f(t0, y0) joins the history (newest first, at most max_order - 1 entries); with fewer than 3 entries the step
      is the 3/8-rule Runge-Kutta step reusing f0 (rk4_alt_step_func); otherwise an Adams-Bashforth predictor of the
      history's length followed by functional iteration of the Adams-Moulton corrector with one more sample,
      dy <- dt * m0 * f(t1, y0 + dy) + delta, at most max_iters times, stopping when
      max |dy_old - dy| / (atol + rtol * max(|dy_old|, |dy|)) < 1; if the iteration does not converge the OLDEST history
      entry is dropped (and torchdiffeq warns).  y1 = y0 + dy."""
    import collections
    prev_f = collections.deque(maxlen=max_order - 1)
    y = y0
    nfe = 0
    for i in range(len(t_span) - 1):
        t0, t1 = t_span[i], t_span[i + 1]
        dt = t1 - t0
        f0 = f(t0, y)
        nfe += 1
        prev_f.appendleft(f0)
        order = min(len(prev_f), max_order - 1)
        if order < 3:
            k1 = f0
            k2 = f(t0 + dt / 3, y + dt * k1 / 3)
            k3 = f(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
            k4 = f(t1, y + dt * (k1 - k2 + k3))
            nfe += 3
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        else:
            bash, _ = adams_coefficients(order)
            _, moul = adams_coefficients(order + 1)
            dy = sum((dt * b) * fj for b, fj in zip(bash, prev_f))
            delta = dt * sum(m * fj for m, fj in zip(moul[1:], prev_f))
            converged = False
            for _ in range(max_iters):
                dy_old = dy
                fn = f(t1, y + dy)
                nfe += 1
                dy = (dt * moul[0]) * fn + delta
                tol = atol + rtol * torch.max(dy_old.abs(), dy.abs())
                converged = bool(((dy_old - dy).abs() / tol).max() < 1)
                if converged:
                    break
            if not converged:
                prev_f.pop()
        y = y + dy
    if stats is not None:
        stats.update(nfe=nfe)
    return y


# Dormand-Prince 5(4) tableau and controller constants of torchdiffeq 0.2.x (rk_common.py, dopri5.py,
# misc.py), restated from the published algorithm -- torchdiffeq is not installable offline: PARITY UNPINNED.
_DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_DP_C_SOL = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0]
_DP_C_ERR = [35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
             -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0]
_DP_C_MID = [6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
             187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _rms(x):
    return float(x.abs().pow(2).mean().sqrt())


# torchdiffeq's explicit adaptive Runge-Kutta solvers (rk_common.py + dopri5.py / bosh3.py / fehlberg2.py /
# adaptive_heun.py of torchdiffeq 0.2.x), restated from the published tableaus: (alpha, beta, c_sol, c_error,
# c_mid, order).  PARITY UNPINNED against torchdiffeq itself (absent offline).
ADAPTIVE_TABLEAUS = {
    "dopri5": (_DP_ALPHA, _DP_BETA, _DP_C_SOL, _DP_C_ERR, _DP_C_MID, 5),
    "bosh3": ([1 / 2, 3 / 4, 1.0],
              [[1 / 2], [0.0, 3 / 4], [2 / 9, 1 / 3, 4 / 9]],
              [2 / 9, 1 / 3, 4 / 9, 0.0],
              [2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8],
              [0.0, 0.5, 0.0, 0.0], 3),
    "fehlberg2": ([1 / 2, 1.0],
                  [[1 / 2], [1 / 256, 255 / 256]],
                  [1 / 512, 255 / 256, 1 / 512],
                  [-1 / 512, 0.0, 1 / 512],
                  [0.0, 0.5, 0.0], 2),
    "adaptive_heun": ([1.0], [[1.0]], [0.5, 0.5], [0.5, -0.5], [0.5, 0.0], 2),
}


def odeint_adaptive(f, y0, method="dopri5", t_end=1.0, rtol=1e-5, atol=1e-5, stats=None):
    """torchdiffeq's RKAdaptiveStepsizeODESolver as called at models/flow_matching.py:54 (rtol=atol=1e-5),
    integrating from 0 to t_end and returning the state at t_end.  Intermediate output times only add
    interpolation, never change the steps, so they are not modelled.  Steps are NOT clipped to t_end: the last
    step may overshoot and the result is the 4th-order dense-output interpolant at t_end (_interp_fit with
    f0 = k[0], f1 = k[-1]).  As in torchdiffeq, the derivative carried into the next step is k[-1] whether or not
    the tableau is FSAL.  Time is carried in float64 on the host; f receives it as an fp32 0-dim tensor."""
    alphas, betas, c_sol, c_err, c_mid, order = ADAPTIVE_TABLEAUS[method]
    fsal = c_sol[-1] == 0.0 and list(c_sol[:-1]) == list(betas[-1])
    tt = lambda t: torch.tensor(t, dtype=torch.float32)
    t0 = 0.0
    f0 = f(tt(t0), y0)
    nfe = 1
    # _select_initial_step(order - 1)
    scale = atol + y0.abs() * rtol
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = f(tt(t0 + h0), y0 + h0 * f0)
    nfe += 1
    d2 = _rms((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / order)
    dt = min(100 * h0, h1)
    y, fcur, t = y0, f0, t0
    steps = rejects = 0
    while True:
        t1 = t + dt
        k = [fcur]
        for alpha, beta in zip(alphas, betas):
            ti = t1 if alpha == 1.0 else t + alpha * dt
            yi = y + sum((b * dt) * kj for b, kj in zip(beta, k) if b != 0.0)
            k.append(f(tt(ti), yi))
            nfe += 1
        # c_sol[:-1] == beta[-1] and c_sol[-1] == 0 (FSAL, e.g. Dormand-Prince): the last stage input IS y1
        y1 = yi if fsal else y + sum((c * dt) * kj for c, kj in zip(c_sol, k) if c != 0.0)
        err = sum((c * dt) * kj for c, kj in zip(c_err, k) if c != 0.0)
        tol = atol + rtol * torch.max(y.abs(), y1.abs())
        ratio = _rms(err / tol)
        accept = ratio <= 1.0
        # _optimal_step_size(safety 0.9, ifactor 10, dfactor 0.2, order)
        if ratio == 0.0:
            dt_next = dt * 10.0
        else:
            dfactor = 1.0 if ratio < 1.0 else 0.2
            dt_next = dt * min(10.0, max(0.9 / ratio ** (1.0 / order), dfactor))
        steps += 1
        if accept:
            if t1 >= t_end:
                # dense output (_interp_fit / _interp_evaluate) at t_end inside [t, t1]
                y_mid = y + sum((c * dt) * kj for c, kj in zip(c_mid, k) if c != 0.0)
                fa, fb = k[0], k[-1]
                a = 2 * dt * (fb - fa) - 8 * (y1 + y) + 16 * y_mid
                b = dt * (5 * fa - 3 * fb) + 18 * y + 14 * y1 - 32 * y_mid
                c = dt * (fb - 4 * fa) - 11 * y - 5 * y1 + 16 * y_mid
                d = dt * fa
                x = (t_end - t) / (t1 - t)
                out = (((a * x + b) * x + c) * x + d) * x + y
                if stats is not None:
                    stats.update(nfe=nfe, steps=steps, rejects=rejects)
                return out
            y, fcur, t = y1, k[-1], t1
        else:
            rejects += 1
        dt = dt_next
        if not dt > 0.0 or dt < 1e-12:
            raise RuntimeError(f"{method}: step size underflow")


def odeint_dopri5(f, y0, t_end=1.0, rtol=1e-5, atol=1e-5, stats=None):
    """torchdiffeq's dopri5 (the reference default, models/flow_matching.py:54 with solver=None)."""
    return odeint_adaptive(f, y0, "dopri5", t_end, rtol, atol, stats)


@torch.inference_mode()
def cfm_forward(sd, mu, mask, n_timesteps, z, c, solver="euler", cfg_kwargs=None, **kw):
    """models/flow_matching.py:25-55 (CFMDecoder.forward) with the noise z passed explicitly
    (z must already include the temperature factor of :45)."""
    t_span = linspace_f32(n_timesteps)
    if cfg_kwargs is None:
        f = lambda t, x: decoder_forward(sd, t, x, mask, mu, c, **kw)
    else:
        f = lambda t, x: cfg_wrapper(sd, t, x, mask, mu, c, cfg_kwargs["fake_speaker"],
                                     cfg_kwargs["fake_content"], cfg_kwargs["cfg_strength"], **kw)
    if solver in (None, "dopri5"):
        return odeint_dopri5(f, z, float(t_span[-1]))
    if solver in ADAPTIVE_TABLEAUS:
        return odeint_adaptive(f, z, solver, float(t_span[-1]))
    if solver == "implicit_adams":
        return odeint_implicit_adams(f, z, t_span)
    return odeint_fixed(f, z, t_span, solver)


def compute_loss(sd, x1, mask, mu, c, t_rand, z, sigma_min=1e-4, **kw):
    """models/flow_matching.py:69-100 with the random draws (t_rand = torch.rand([b,1,1]),
    z = randn_like(x1)) passed explicitly.  NB the estimator output is masked but u is not
    (reference quirk kept)."""
    t = 1 - torch.cos(t_rand * 0.5 * torch.pi)
    y = (1 - (1 - sigma_min) * t) * z + t * x1
    u = x1 - (1 - sigma_min) * z
    pred = decoder_forward(sd, t.squeeze(), y, mask, mu, c, **kw)
    loss = F.mse_loss(pred, u, reduction="sum") / (torch.sum(mask) * u.size(1))
    return loss, y
