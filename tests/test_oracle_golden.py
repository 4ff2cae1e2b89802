"""The oracle (oracle/estimator_oracle.py) against fixtures produced by the REAL reference
modules (oracle/make_golden.py).  CPU only.  Tolerance: fp32 restatement vs fp32 reference,
different-but-equivalent op order (explicit softmax vs SDPA) => 2e-5 abs on O(1) values."""
import numpy as np
import pytest
import torch

import oracle
from oracle import estimator_oracle as eo
from oracle.inputs import make_inputs

TOL = 2e-5


def _close(a, b, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max()
    scale = max(1.0, np.abs(b).max())
    assert err <= tol * scale, f"max abs err {err:.3e} (scale {scale:.3f})"


def test_state_dict_names_shapes(sd):
    assert len(sd) == 116
    assert sum(v.numel() for v in sd.values()) == 20347008
    assert sd["in_proj.weight"].shape == (256, 384, 1)
    assert sd["blocks.5.block.adaLN_modulation.2.weight"].shape == (1536, 256)
    assert sd["lsc_layers.2.weight"].shape == (256, 512, 3)


@torch.inference_mode()
def test_nfe_scalar_t(sd, golden):
    inp = make_inputs(2, 70, seed=11, lengths=[70, 51])
    out = oracle.decoder_forward(sd, torch.tensor(0.3), inp["z"], inp["mask"], inp["mu"], inp["c"])
    _close(out, golden["nfe_scalar_t"])
    assert float(out[1, :, 51:].abs().max()) == 0.0


@torch.inference_mode()
def test_nfe_batched_t(sd, golden):
    inp = make_inputs(3, 40, seed=12, lengths=[40, 33, 17])
    tb = torch.tensor([0.05, 0.5, 0.93])
    out = oracle.decoder_forward(sd, tb, inp["z"], inp["mask"], inp["mu"], inp["c"])
    _close(out, golden["nfe_batched_t"])


@torch.inference_mode()
def test_submodules(sd, golden):
    inp = make_inputs(2, 37, seed=13, lengths=[37, 20])
    g = torch.Generator().manual_seed(5)
    xs = torch.randn(2, 256, 37, generator=g)
    m = inp["mask"]
    _close(eo.ffn(sd, "blocks.2.block.mlp.", xs, m), golden["sub_ffn"])
    _close(eo.mha(sd, "blocks.2.block.attn.", xs, m), golden["sub_mha"])
    # DiTConVBlock alone == dit_block with identity FiLM (gamma=1,beta=0): emulate via direct call
    sd2 = dict(sd)
    sd2["blocks.2.time_fusion.film.weight"] = torch.zeros_like(sd["blocks.2.time_fusion.film.weight"])
    b = torch.zeros(512)
    b[:256] = 1.0
    sd2["blocks.2.time_fusion.film.bias"] = b
    _close(eo.dit_block(sd2, 2, xs, inp["c"], torch.zeros(1, 256), m), golden["sub_block"])
    tau = torch.randn(1, 256, generator=g)
    _close(eo.dit_block(sd, 2, xs, inp["c"], tau, m), golden["sub_wrapper"])
    xr = torch.randn(2, 4, 37, 64, generator=g)
    _close(eo.rope(xr, 32), golden["sub_rope"])
    _close(eo.sinusoidal_pos_emb(torch.tensor([0.0, 0.123, 1.0])), golden["sub_temb"])
    _close(eo.cond_proj(sd, inp["mu"]), golden["sub_condproj"])


CASES = [
    ("solve_euler_cfg", 2, 64, [64, 45], 4, "euler", 3.0, 21),
    ("solve_euler_nocfg", 1, 50, [50], 5, "euler", None, 22),
    ("solve_midpoint", 1, 48, [48], 3, "midpoint", None, 23),
    ("solve_rk4_cfg", 2, 33, [33, 30], 2, "rk4", 2.0, 24),
]


@pytest.mark.parametrize("name,B,T,lengths,n,solver,cfg,seed", CASES)
def test_full_solve(sd, cfg_params, golden, name, B, T, lengths, n, solver, cfg, seed):
    inp = make_inputs(B, T, seed=seed, lengths=lengths)
    z = torch.from_numpy(golden[name + "_z"])
    fs, fc = cfg_params
    kw = None if cfg is None else dict(fake_speaker=fs, fake_content=fc, cfg_strength=cfg)
    out = oracle.cfm_forward(sd, inp["mu"], inp["mask"], n, z, inp["c"], solver, kw)
    _close(out, golden[name], 5e-5)


@torch.inference_mode()
def test_compute_loss(sd, golden):
    inp = make_inputs(2, 44, seed=31, lengths=[44, 29])
    x1 = make_inputs(2, 44, seed=32)["z"]
    loss, y = oracle.compute_loss(sd, x1, inp["mask"], inp["mu"], inp["c"],
                                  torch.from_numpy(golden["loss_t_rand"]), torch.from_numpy(golden["loss_z"]))
    _close(y, golden["loss_y"])
    assert abs(float(loss) - float(golden["loss_value"][0])) <= 1e-5 * float(golden["loss_value"][0])


@torch.inference_mode()
def test_fused_cfg_batch_equivalence(sd, cfg_params):
    """cond/uncond as one 2B batch == two sequential calls (the engine's layout)."""
    inp = make_inputs(2, 40, seed=3, lengths=[40, 27])
    fs, fc = cfg_params
    t = torch.tensor(0.4)
    ref = oracle.cfg_wrapper(sd, t, inp["z"], inp["mask"], inp["mu"], inp["c"], fs, fc, 3.0)
    x2 = torch.cat([inp["z"], inp["z"]])
    m2 = torch.cat([inp["mask"], inp["mask"]])
    mu2 = torch.cat([inp["mu"], fc.repeat(2, 1, 40)])
    c2 = torch.cat([inp["c"], fs.repeat(2, 1)])
    o = oracle.decoder_forward(sd, t, x2, m2, mu2, c2)
    fused = o[2:] + 3.0 * (o[:2] - o[2:])
    assert float((fused - ref).abs().max()) <= 1e-6


def test_text_encoder_oracle_matches_reference_fixture():
    """oracle.text_encoder_forward vs the outputs of the real reference TextEncoder
    (tests/golden/text_encoder_outputs.npz, written by oracle/make_golden_text_encoder.py)."""
    import os
    import numpy as np
    import torch
    import oracle
    from oracle.make_golden_text_encoder import CASES, text_inputs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "text_encoder_outputs.npz"))
    sd = oracle.make_text_encoder_state_dict(2468)
    for name, (B, T, lengths, seed) in CASES.items():
        tok, c, lens = text_inputs(B, T, lengths, seed)
        with torch.inference_mode():
            x, mu_x, mask = oracle.text_encoder_forward(sd, tok, c, lens)
        for got, key in ((x, "_x"), (mu_x, "_mu_x"), (mask, "_mask")):
            want = g[name + key]
            assert got.shape == want.shape
            assert np.abs(got.numpy() - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0)


def test_oracle_gradients_match_reference_fixture(sd, golden):
    """Backward arithmetic of the oracle (autograd through oracle.compute_loss) vs gradients of the REAL reference
    modules (tests/golden/loss_grads.npz, oracle/make_golden_grads.py; dropout off): every parameter's gradient
    norm, seven tensors in full, d loss / d mu and d loss / d c.  Pins the checker for the native backward pass
     before that pass exists."""
    import os
    import numpy as np
    import torch
    import oracle
    from oracle.inputs import make_inputs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_grads.npz"))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    inp = make_inputs(2, 44, seed=31, lengths=[44, 29])
    x1 = make_inputs(2, 44, seed=32)["z"]
    mu = inp["mu"].clone().requires_grad_(True)
    c = inp["c"].clone().requires_grad_(True)
    loss, _ = oracle.compute_loss(p, x1, inp["mask"], mu, c, torch.from_numpy(golden["loss_t_rand"]),
                                  torch.from_numpy(golden["loss_z"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss_value"][0])) <= 1e-5 * float(g["loss_value"][0])
    names = [str(n) for n in g["names"]]
    assert names == list(sd.keys()) or set(names) == set(sd.keys())
    scale = float(g["grad_norms"].max())
    for name, want in zip(names, g["grad_norms"]):
        got = float(p[name].grad.double().norm())
        assert abs(got - want) <= 2e-4 * want + 1e-7 * scale, name
    for key in g.files:
        if key.startswith("grad."):
            got, want = p[key[5:]].grad.numpy(), g[key]
            assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-9, key
    for got, want in ((mu.grad.numpy(), g["grad_mu"]), (c.grad.numpy(), g["grad_c"])):
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()


def _order_conditions(b, c, A, order, theta=1.0):
    """Residuals of the Runge-Kutta order conditions up to ``order`` (<= 4) for weights b (at abscissa theta),
    nodes c and stage matrix A: a method y(t + theta h) = y + h sum b_i k_i of that order satisfies all of them."""
    import numpy as np
    b, c, A = (np.asarray(v, dtype=np.float64) for v in (b, c, A))
    res = [b.sum() - theta]
    if order >= 2:
        res.append(b @ c - theta ** 2 / 2)
    if order >= 3:
        res += [b @ c ** 2 - theta ** 3 / 3, b @ (A @ c) - theta ** 3 / 6]
    if order >= 4:
        res += [b @ c ** 3 - theta ** 4 / 4, (b * c) @ (A @ c) - theta ** 4 / 8, b @ (A @ c ** 2) - theta ** 4 / 12,
                b @ (A @ (A @ c)) - theta ** 4 / 24]
    return np.abs(np.array(res)).max()


def test_adaptive_tableaus_pinned_against_scipy_and_order_conditions():
    """torchdiffeq is absent offline, so its adaptive tableaus (oracle.ADAPTIVE_TABLEAUS, restated from the published
    algorithm) are pinned against an independent implementation and against the algebra that defines them:
      * nodes, stage matrix and propagating weights of dopri5 / bosh3 == scipy.integrate RK45 / RK23 (Dormand-Prince
        5(4), Bogacki-Shampine 3(2)); bosh3's error weights == -RK23.E;
      * the embedded lower-order weights (c_sol - c_error) of every pair satisfy the order conditions of order
        (order - 1); dopri5's are the Shampine (ode45) variant torchdiffeq uses, which differs from scipy's
        Dormand-Prince b* by design, so they are pinned by the order conditions instead;
      * the dense-output midpoint weights c_mid (torchdiffeq DPS_C_MID) satisfy the order conditions at theta = 1/2
        (order 4 for dopri5; the half-step Euler/stage value for the low-order pairs);
      * FSAL structure: dopri5 / bosh3's last stage row equals c_sol."""
    import numpy as np
    from scipy.integrate._ivp import rk
    import oracle
    for name, ref in (("dopri5", rk.RK45), ("bosh3", rk.RK23)):
        alpha, beta, c_sol, c_err, c_mid, order = oracle.ADAPTIVE_TABLEAUS[name]
        n = len(alpha)                        # stages after k0
        A = np.zeros((n + 1, n + 1))
        for i, row in enumerate(beta):
            A[i + 1, :len(row)] = row
        c = np.array([0.0] + list(alpha))
        ns = ref.n_stages
        assert np.abs(c[:ns] - ref.C).max() < 1e-15
        assert np.abs(A[:ns, :ns - 1] - ref.A[:, :ns - 1]).max() < 1e-15
        assert np.abs(np.array(c_sol[:ns]) - ref.B).max() < 1e-15 and c_sol[ns] == 0.0
        assert np.abs(A[ns, :ns] - ref.B).max() < 1e-15               # FSAL: last stage input is y1
        assert ref.order == order
        if name == "bosh3":
            assert np.abs(np.array(c_err) + ref.E).max() < 1e-15
    for name, (alpha, beta, c_sol, c_err, c_mid, order) in oracle.ADAPTIVE_TABLEAUS.items():
        n = len(alpha)
        A = np.zeros((n + 1, n + 1))
        for i, row in enumerate(beta):
            A[i + 1, :len(row)] = row
        c = np.array([0.0] + list(alpha))
        assert _order_conditions(c_sol, c, A, min(order, 4)) < 1e-14, name
        emb = np.array(c_sol) - np.array(c_err)
        assert _order_conditions(emb, c, A, min(order - 1, 4)) < 1e-14, name
        assert abs(sum(c_err)) < 1e-15, name
        mid_order = 4 if name == "dopri5" else 1
        assert _order_conditions(c_mid, c, A, mid_order, theta=0.5) < 1e-12, name
    # dopri5 propagating weights are 5th order: the additional order-5 quadrature condition
    alpha, beta, c_sol, *_ = oracle.ADAPTIVE_TABLEAUS["dopri5"]
    c = np.array([0.0] + list(alpha))
    assert abs(np.array(c_sol) @ c ** 4 - 1 / 5) < 1e-15


@pytest.mark.parametrize("method,scipy_method", [("dopri5", "RK45"), ("bosh3", "RK23")])
def test_adaptive_oracle_solves_the_ode_to_tolerance_vs_scipy(method, scipy_method):
    """The restated adaptive controller (torchdiffeq is absent offline: its step controller stays parity-unpinned) is checked
    against an INDEPENDENT integrator for what it is supposed to deliver: the solution of the ODE to the requested tolerance.  A
    small stiff-ish nonlinear system with a time-dependent field (the shape of flow_matching.py:49-54: state in, t in, vector field
    out), integrated over [0, 1] at rtol = atol = 1e-5 as the reference hard-codes:
      * truth = scipy.integrate.solve_ivp(DOP853, rtol = atol = 1e-12);
      * the oracle's answer must be within 50 x the tolerance scale of the truth (measured ~1e-5 relative);
      * scipy's own RK45 at the same rtol / atol (same tableau, scipy's controller) lands at a comparable distance: dopri5 -- the
        reference's default solver -- is not allowed to be more than 10 x further away than scipy is (measured 1.4 vs 0.6 x tol).
        bosh3 is exempt: like torchdiffeq the oracle does not clip its last step to t = 1 but evaluates the dense-output
        interpolant there, and for bosh3 that interpolant's mid-point is only second-order (28 x tol against scipy's 0.3);
      * and the two controllers take a comparable number of steps (within 2 x): neither crawls nor leaps."""
    import numpy as np
    from scipy.integrate import solve_ivp
    torch.manual_seed(3)
    A = (torch.randn(6, 6, dtype=torch.float64) * 0.9 - 1.2 * torch.eye(6, dtype=torch.float64))
    w = torch.linspace(1.0, 9.0, 6, dtype=torch.float64)

    def field64(t, y):
        return np.tanh(A.numpy() @ y) * 3.0 + np.sin(w.numpy() * t * 2.0) + 0.5 * y * np.cos(3.0 * t)

    def field(t, y):          # the oracle integrates fp32 tensors, t arrives as an fp32 0-dim tensor (as the estimator gets it)
        td = t.double()
        return (torch.tanh(A @ y.double()) * 3.0 + torch.sin(w * td * 2.0) + 0.5 * y.double() * torch.cos(3.0 * td)).float()

    y0 = torch.linspace(-1.0, 1.0, 6)
    truth = solve_ivp(field64, (0.0, 1.0), y0.double().numpy(), method="DOP853", rtol=1e-12, atol=1e-12).y[:, -1]
    sp = solve_ivp(field64, (0.0, 1.0), y0.double().numpy(), method=scipy_method, rtol=1e-5, atol=1e-5)
    stats = {}
    got = oracle.odeint_adaptive(field, y0, method, 1.0, 1e-5, 1e-5, stats).double().numpy()
    scale = 1e-5 * (1.0 + np.abs(truth))
    e_or, e_sp = float(np.max(np.abs(got - truth) / scale)), float(np.max(np.abs(sp.y[:, -1] - truth) / scale))
    sp_steps = len(sp.t) - 1
    print(f"{method}: oracle error {e_or:.2f} x tol ({stats['steps']} steps, {stats['rejects']} rejected), scipy {scipy_method} {e_sp:.2f} x tol ({sp_steps} accepted steps)")
    assert e_or <= 50.0
    assert method != "dopri5" or e_or <= 10.0 * max(e_sp, 1.0)
    assert 0.5 * sp_steps <= stats["steps"] - stats["rejects"] <= 2.0 * sp_steps + 2


def test_fixed_grid_rules_have_their_classical_order():
    """euler / midpoint / rk4 (3/8 rule) of oracle.odeint_fixed, written as Butcher tableaus, satisfy the order
    conditions of order 1 / 2 / 4 -- and the rk4 one is the 3/8 rule torchdiffeq uses (rk4_alt_step_func), checked on
    y' = y where one step must equal the degree-4 Taylor polynomial."""
    import math
    import numpy as np
    import torch
    import oracle
    A38 = np.array([[0, 0, 0, 0], [1 / 3, 0, 0, 0], [-1 / 3, 1, 0, 0], [1, -1, 1, 0]], dtype=np.float64)
    assert _order_conditions([1 / 8, 3 / 8, 3 / 8, 1 / 8], [0, 1 / 3, 2 / 3, 1], A38, 4) < 1e-15
    assert _order_conditions([0, 1], [0, 1 / 2], np.array([[0, 0], [1 / 2, 0]]), 2) < 1e-15
    h = 0.25
    t_span = torch.tensor([0.0, h], dtype=torch.float64)
    y0 = torch.ones(3, dtype=torch.float64)
    for method, deg in (("euler", 1), ("midpoint", 2), ("rk4", 4)):
        y = oracle.odeint_fixed(lambda t, y: y, y0, t_span, method)
        taylor = sum(h ** k / math.factorial(k) for k in range(deg + 1))
        assert abs(float(y[0]) - taylor) < 1e-14, method


def test_implicit_adams_oracle_coefficients_and_behaviour():
    """oracle.odeint_implicit_adams (torchdiffeq's 'implicit_adams', restated: PARITY UNPINNED).  What CAN be pinned offline:
    the Adams-Bashforth / Adams-Moulton weights equal the published integer tables over their divisors, integrate
    polynomials of degree < order exactly, the start-up steps are the 3/8-rule Runge-Kutta steps of oracle.odeint_fixed, and
    the scheme converges at high order on a smooth linear problem with the expected number of evaluations."""
    import math
    import numpy as np
    import torch
    import oracle
    from oracle.estimator_oracle import adams_coefficients, odeint_implicit_adams
    published = {     # (bashforth numerators, moulton numerators, divisor): the tables of any numerical-analysis text / fixed_adams.py
        1: ([1], [1], 1), 2: ([3, -1], [1, 1], 2), 3: ([23, -16, 5], [5, 8, -1], 12),
        4: ([55, -59, 37, -9], [9, 19, -5, 1], 24), 5: ([1901, -2774, 2616, -1274, 251], [251, 646, -264, 106, -19], 720),
        6: ([4277, -7923, 9982, -7298, 2877, -475], [475, 1427, -798, 482, -173, 27], 1440)}
    for k, (bn, mn, div) in published.items():
        b, m = adams_coefficients(k)
        assert np.allclose(np.array(b) * div, bn, atol=1e-9) and np.allclose(np.array(m) * div, mn, atol=1e-9), k
    for k in range(1, 13):
        b, m = adams_coefficients(k)
        for deg in range(k):          # integral over [0, 1] of u^deg from its samples at 0, -1, ... (explicit) / 1, 0, -1, ... (implicit)
            exact = 1.0 / (deg + 1)
            eb = sum(w * (float(-i) ** deg if (i or deg) else 1.0) for i, w in enumerate(b))
            nodes = [1.0] + [float(-i) for i in range(k - 1)]
            em = sum(w * (x ** deg if (x or deg) else 1.0) for w, x in zip(m, nodes))
            scale = float(k) ** deg                   # the sums cancel: tolerance relative to the terms' magnitude
            assert abs(eb - exact) < 1e-13 * scale * sum(abs(w) for w in b) + 1e-14, (k, deg)
            assert abs(em - exact) < 1e-13 * scale * sum(abs(w) for w in m) + 1e-14, (k, deg)
    f = lambda t, y: -2 * y + torch.sin(3 * t)
    y0 = torch.tensor([1.0], dtype=torch.float64)
    exact = (1 + 3 / 13) * math.exp(-2.0) + (2 * math.sin(3.0) - 3 * math.cos(3.0)) / 13
    ts = torch.linspace(0, 1, 3, dtype=torch.float64)            # two steps: both are Runge-Kutta start-up steps
    assert torch.equal(odeint_implicit_adams(f, y0, ts), oracle.odeint_fixed(f, y0, ts, "rk4"))
    errs = {}
    for n in (10, 20, 40):
        st = {}
        y = odeint_implicit_adams(f, y0, torch.linspace(0, 1, n + 1, dtype=torch.float64), stats=st)
        errs[n] = abs(float(y) - exact)
        assert 8 + (n - 2) * 2 <= st["nfe"] <= 8 + (n - 2) * 5            # 2 x 4 start-up evaluations, then 1 + (1..4) per step
    assert errs[10] < 1e-5 and errs[40] < errs[10] * 1e-3


def test_alignment_oracle_matches_reference_fixture():
    """oracle.align_oracle (numpy restatement of models/model.py:17-27,85-95) vs outputs of the REAL generate_path /
    sequence_mask (tests/golden/align_outputs.npz, oracle/make_golden_align.py): integer results bit-exact."""
    import os
    import numpy as np
    from oracle.align_oracle import length_regulate
    from oracle.make_golden_align import CASES, align_inputs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "align_outputs.npz"))
    for name, (B, Tx, xl, M, ls, seed) in CASES.items():
        logw, mu_x, x_mask, _ = align_inputs(B, Tx, xl, M, seed)
        r = length_regulate(logw.numpy(), x_mask.numpy(), mu_x.numpy(), ls)
        assert np.array_equal(r["w_ceil"], g[name + "_w_ceil"]), name
        assert np.array_equal(r["y_lengths"], g[name + "_y_lengths"]), name
        assert np.array_equal(r["y_mask"], g[name + "_y_mask"]), name
        assert np.array_equal(r["attn"].astype(np.uint8), g[name + "_attn"]), name
        assert np.abs(r["mu_y"] - g[name + "_mu_y"]).max() <= 1e-6, name
        assert (r["attn"].sum(1) <= 1).all()                       # at most one text position per mel frame


def test_vocos_oracle_matches_reference_fixture():
    """oracle.vocos_oracle (numpy restatement of vocoders/vocos/models/{backbone,module,head}.py) vs outputs of the REAL
    Vocos module in fp32 (tests/golden/vocos_outputs.npz, oracle/make_golden_vocos.py): backbone output and waveform."""
    import os
    import numpy as np
    from oracle import vocos_oracle as vo
    from oracle.make_golden_vocos import CASES, SD_SEED
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocos_outputs.npz"))
    sd = vo.make_vocos_state_dict(SD_SEED)
    for name, (B, T, seed) in CASES.items():
        mel = vo.make_mel(B, T, seed)
        hid = vo.backbone_forward(sd, mel)
        audio = vo.head_forward(sd, hid)
        assert audio.shape == (B, T * 512)
        assert np.abs(hid - g[name + ".hidden"]).max() <= 2e-6 * np.abs(g[name + ".hidden"]).max(), name
        assert np.abs(audio - g[name + ".audio"]).max() <= 5e-6 * np.abs(g[name + ".audio"]).max(), name


def test_vocos_oracle_istft_is_inverse_of_stft():
    """Size-independent property of the ISTFT restatement (head.py:39-72): with "same" padding it inverts an STFT with the
    same window / hop on the interior of the signal (window-sum normalisation), and it is linear in the spectrum."""
    import numpy as np
    from oracle import vocos_oracle as vo
    rng = np.random.default_rng(0)
    n_fft, hop, T = 2048, 512, 12
    win = vo.make_vocos_state_dict(1)["head.istft.window"].astype(np.float64)
    x = rng.standard_normal(T * hop)
    pad = (n_fft - hop) // 2
    xp = np.pad(x, (pad, pad))
    S = np.stack([np.fft.rfft(xp[t * hop:t * hop + n_fft] * win) for t in range(T)], axis=1)[None]
    y = vo.istft_same(S, win, n_fft, hop)[0]
    assert np.abs(y - x)[pad:-pad].max() < 1e-9                                     # interior: all 4 frames present
    S2 = rng.standard_normal(S.shape) + 1j * rng.standard_normal(S.shape)
    lhs = vo.istft_same(2.0 * S + 3.0 * S2, win, n_fft, hop)
    assert np.abs(lhs - (2.0 * y[None] + 3.0 * vo.istft_same(S2, win, n_fft, hop))).max() < 1e-9
