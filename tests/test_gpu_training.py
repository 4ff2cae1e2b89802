"""Native training path (st_train_forward / st_train_backward behind stabletts_amd.autograd) on a real MI355X:
gradients of CFMDecoder.compute_loss against the REAL reference's gradients (tests/golden/loss_grads.npz, dropout
off), the attention backward kernels at matched inputs, counter-based dropout against the oracle run with the same
masks, and a 2-process DDP run.  Run with ``-m gpu``.

Gates (max |native - ref| / max |ref| per tensor): f16 operands 3e-3, bf16 operands 2e-2 -- except the q / k
projections of the attention: d loss / d q and d loss / d k subtract two nearly equal terms (dP - D) and, at these
random-init weights where the softmax is near-uniform, dS ~ dO.(v_j - o_i): the key-independent bulk of v cancels, a 16-bit
error of v does not.  With v and the projection's input as single 16-bit operands (rounds 1-5, ST_TRAIN_VLO=0) those
gradients were 2 % off the fp32 reference at T = 44 and 25 % at T = 1000; since round 6 v is computed from h1 as a hi + lo
pair and kept as a hi + lo pair through the attention (2e-3 at T = 44, 3-4 % at T = 1000, cosine 0.9996): gated at 1e-2 /
8e-2.  The chain itself is gated at matched operands in test_gradients_at_config5_size: the
attention backward kernels vs fp64 on the native q, k, v, d attn; RoPE^T + pack + weight-gradient GEMM vs fp64 on the native
dq, dk, h1; and END TO END against the oracle's autograd evaluated at the native forward's own q, k, v
(oracle.attention(subst=...)): 5e-3 (f16) at B=4 x T=1000.  (That comparison is what exposed, in round 3, that f16 rounded
d q / d k to subnormals at the pass-wide gradient scale; they now carry their own power-of-two scales.)
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle.inputs import make_inputs

pytestmark = [pytest.mark.gpu, pytest.mark.grad]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = {"f16": 3e-3, "bf16": 2e-2}          # measured 5.3e-4 / 4.2e-3
TOL_QK = {"f16": 1e-2, "bf16": 1e-1}       # end to end, measured 2.1e-3 / 2.3e-2 since round 6 (v from hi + lo operands: 2.3e-2 / 3.4e-1 before -- the
                                           # conditioning of dq, dk in v at random init, test_v_as_a_hi_lo_operand_pair_in_the_training_forward)


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _is_qk(name):
    return ".attn.conv_q." in name or ".attn.conv_k." in name


def _native_v(eng, i, B, H, T, Tp, pos):
    """The forward's v operand of block i as the attention kernels see it, [B][H][T][64]: the 16-bit plane plus (round 6: hi + lo v
    operands in the training forward, ST_TRAIN_VLO) its rounding residuals."""
    v = eng.debug_fetch(f"t{i}.vt").reshape(B, H, 64, Tp).astype(np.float64)
    try:
        v = v + eng.debug_fetch(f"t{i}.vtlo").reshape(B, H, 64, Tp).astype(np.float64)
    except Exception:      # noqa: BLE001  (ST_TRAIN_VLO=0: there is no residual plane)
        pass
    return v[..., pos][..., :T].transpose(0, 1, 3, 2)


def _decoder(sd, dt, train=False):
    from stabletts_amd.flow_matching import CFMDecoder
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dt)
    d.estimator.load_state_dict(sd)
    d = d.cuda()
    return d.train() if train else d.eval()


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_gradients_match_reference_fixture(sd, golden, dt):
    """Every parameter gradient, d loss / d mu and d loss / d c of one compute_loss step vs the gradients of the REAL
    reference modules (oracle/make_golden_grads.py)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "loss_grads.npz"))
    dec = _decoder(sd, dt)
    inp = make_inputs(2, 44, seed=31, lengths=[44, 29])
    x1 = make_inputs(2, 44, seed=32)["z"]
    mu = inp["mu"].cuda().requires_grad_(True)
    c = inp["c"].cuda().requires_grad_(True)
    loss, y = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), mu, c, t_rand=torch.from_numpy(golden["loss_t_rand"]).cuda(),
                               z=torch.from_numpy(golden["loss_z"]).cuda())
    assert loss.requires_grad
    loss.backward()
    want = float(g["loss_value"][0])
    assert abs(float(loss.detach()) - want) <= {"f16": 5e-4, "bf16": 3e-3}[dt] * want
    names = [str(n) for n in g["names"]]
    params = dict(dec.estimator.named_parameters())
    assert set(names) == set(params)
    worst = {}
    for name, ref_norm in zip(names, g["grad_norms"]):
        gr = params[name].grad
        assert gr is not None and torch.isfinite(gr).all(), name
        worst[name] = abs(float(gr.double().norm()) - ref_norm) / ref_norm
    for key in g.files:
        if key.startswith("grad."):
            name = key[5:]
            worst[name] = max(worst[name], _rel(params[name].grad.cpu().numpy(), g[key]))
    bad = {k: v for k, v in worst.items() if v > (TOL_QK if _is_qk(k) else TOL)[dt]}
    print(f"[{dt}] worst non-q/k: {max(v for k, v in worst.items() if not _is_qk(k)):.2e}; worst q/k: {max(v for k, v in worst.items() if _is_qk(k)):.2e}")
    assert not bad, bad
    assert _rel(mu.grad.cpu().numpy(), g["grad_mu"]) <= TOL[dt]
    assert _rel(c.grad.cpu().numpy(), g["grad_c"]) <= TOL[dt]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_attention_backward_at_matched_inputs(sd, dt):
    """dq, dk, dv of the flash-attention backward kernels against an fp64 evaluation of the same formulas on the
    NATIVE forward's own 16-bit q, k, v and on the native d attn (debug capture): kernel correctness to operand
    precision, independent of the forward's deviation from the fp32 reference.  Ragged batch, T across two key tiles."""
    dec = _decoder(sd, dt)
    B, T, lengths = 2, 100, [100, 71]
    inp = make_inputs(B, T, seed=41, lengths=lengths)
    x1 = make_inputs(B, T, seed=42)["z"]
    eng = dec.estimator.engine()
    eng.debug_capture(True)
    try:
        loss, _ = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
        loss.backward()
        torch.cuda.synchronize()
        Tp = (T + 63) // 64 * 64
        tt = np.arange(Tp)
        pos = (tt & ~12) | ((tt & 4) << 1) | ((tt & 8) >> 1)
        m = inp["mask"][:, 0].double().numpy()
        bias = (1 - m)[:, None, None, :] * (-1e30)
        for i in (5, 2):
            scale = float(eng.debug_fetch(f"g.scale_a{i}")[0])     # the pass-wide power-of-two scale, re-centred before the attention part
            q = eng.debug_fetch(f"t{i}.q").reshape(B, 4, T, 64).astype(np.float64)        # q_s = q_r * log2(e) / 8
            k = eng.debug_fetch(f"t{i}.k").reshape(B, 4, T, 64).astype(np.float64)
            v = _native_v(eng, i, B, 4, T, Tp, pos)
            dO = (eng.debug_fetch(f"g.dattn_{i}").reshape(B, T, 4, 64).transpose(0, 2, 1, 3) / scale).astype(np.float64)
            S2 = q @ k.transpose(0, 1, 3, 2) + bias
            S2 -= S2.max(-1, keepdims=True)
            P = np.exp2(S2); P /= P.sum(-1, keepdims=True)
            dP = dO @ v.transpose(0, 1, 3, 2)
            D = (P * dP).sum(-1, keepdims=True)
            dS = P * (dP - D)
            want = {"dq": dS @ k, "dk": dS.transpose(0, 1, 3, 2) @ q, "dv": P.transpose(0, 1, 3, 2) @ dO}
            for nm, ref in want.items():
                got = eng.debug_fetch(f"g.{nm}_{i}").reshape(B, 4, T, 64) / scale
                r = _rel(got, ref)
                print(f"[{dt}] block {i} {nm}: {r:.2e}")
                assert r <= {"f16": 1e-3, "bf16": 8e-3}[dt], (i, nm, r)       # measured <= 4e-4 / 3.8e-3
    finally:
        eng.debug_capture(False)


# ---------------------------------------------------------------- BASELINE config 5 shapes (train.py:78-81 at T = 1000)
SIZE_B, SIZE_T, SIZE_LENS = 4, 1000, [1000, 873, 655, 512]
# q / k projections at T = 1000 (measured on MI355X, f16 / bf16):
#   end to end vs the fp32 oracle       2.9e-2 / 1.2e-1, cosine 0.99957 / 0.9886  (round 6: v = W_v (h_hi + h_lo) kept as a hi + lo operand pair;
#                                        2.5e-1 / 1.6, cosine 0.991 / 0.82 with v and its input as single 16-bit operands, ST_TRAIN_VLO=0)
#   vs the oracle AT the native q, k, v  5.4e-3 / 1.5e-2, cosine 0.999994 / 0.9999  -- the native backward chain itself
TOL_QK_SIZE = {"f16": 8e-2, "bf16": 3e-1}
COS_QK_SIZE = {"f16": 0.999, "bf16": 0.97}      # (bf16 is not a training dtype with parity: INTEGRATION.md, train with the default f16 operands)
TOL_QK_MATCHED = {"f16": 1e-2, "bf16": 6e-2}


@pytest.fixture(scope="module")
def size_case(sd):
    """One compute_loss step at config-5 frame counts (B=4 x T=1000, ragged) through the ORACLE's autograd (fp32 CPU,
    ~3 s): loss, all 116 parameter gradients, d loss/d mu, d loss/d c."""
    inp = make_inputs(SIZE_B, SIZE_T, seed=81, lengths=SIZE_LENS)
    x1 = make_inputs(SIZE_B, SIZE_T, seed=82)["z"]
    g0 = torch.Generator().manual_seed(19)
    t_rand = torch.rand(SIZE_B, 1, 1, generator=g0)
    z = torch.randn(SIZE_B, 128, SIZE_T, generator=g0)
    with torch.enable_grad():
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        mu = inp["mu"].clone().requires_grad_(True)
        c = inp["c"].clone().requires_grad_(True)
        loss, _ = oracle.compute_loss(pr, x1, inp["mask"], mu, c, t_rand, z)
        loss.backward()
    return dict(inp=inp, x1=x1, t_rand=t_rand, z=z, loss=float(loss.detach()), gmu=mu.grad.numpy(), gc=c.grad.numpy(),
                grads={k: v.grad.numpy() for k, v in pr.items()})


def _cos(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def _rope_T(d, T):
    """Transpose of the partial rotary embedding (pairs (j, j+16), j < 16, theta_j = 10000^(-2j/32)) applied to a
    gradient d[..., T, 64] (models/diffusion_transformer.py:180-198): dx1 = dy1 c + dy2 s, dx2 = dy2 c - dy1 s."""
    j = np.arange(16)
    ang = np.arange(T)[:, None].astype(np.float32) * (1.0 / np.power(np.float32(10000.0), (2 * j) / np.float32(32.0))).astype(np.float32)
    c, s = np.cos(ang.astype(np.float32)).astype(np.float64), np.sin(ang.astype(np.float32)).astype(np.float64)
    out = d.copy()
    out[..., 0:16] = d[..., 0:16] * c + d[..., 16:32] * s
    out[..., 16:32] = d[..., 16:32] * c - d[..., 0:16] * s
    return out


@pytest.mark.parametrize("dt,tiles", [("f16", "policy"), ("f16", "big"), ("bf16", "big")])
def test_gradients_at_config5_size(sd, size_case, monkeypatch, dt, tiles):
    """The production backward configurations against the oracle's autograd on the same inputs: T = 1000 ragged rows
    (16 key tiles in the attention backward, weight gradients split over up to 64 row chunks, dgrad through the
    k = 3 kernels), once with the tile policy as shipped for this batch size and once with ST_BIG_MIN_BLOCKS=1, which
    makes every conv -- forward, dgrad AND wgrad -- take the 256 x 256 / phased tiles a B = 64 batch runs on.
    Gates: every tensor 3e-3 (f16) / 2e-2 (bf16) of max |ref|, as at the small size -- except conv_q / conv_k, whose
    end-to-end error carries the conditioning of dq, dk in v at random init (module docstring: 8e-2 / cosine 0.999 with f16 operands);
    their chain is gated by the MATCHED-INPUT checks below (attention backward vs fp64 on the native q, k, v, d attn; RoPE^T + pack +
    weight-gradient GEMM vs fp64 on the native dq, dk, h1)."""
    sc = size_case
    if tiles == "big":
        monkeypatch.setenv("ST_BIG_MIN_BLOCKS", "1")
    dec = _decoder(sd, dt)
    eng = dec.estimator.engine()
    monkeypatch.delenv("ST_BIG_MIN_BLOCKS", raising=False)
    inp = sc["inp"]
    mu = inp["mu"].cuda().requires_grad_(True)
    c = inp["c"].cuda().requires_grad_(True)
    eng.debug_capture(True)
    try:
        loss, _ = dec.compute_loss(sc["x1"].cuda(), inp["mask"].cuda(), mu, c, t_rand=sc["t_rand"].cuda(), z=sc["z"].cuda())
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.detach()) - sc["loss"]) <= {"f16": 5e-4, "bf16": 3e-3}[dt] * sc["loss"]
        params = dict(dec.estimator.named_parameters())
        worst, cosq = {}, {}
        for name, ref in sc["grads"].items():
            g = params[name].grad
            assert g is not None and torch.isfinite(g).all(), name
            worst[name] = _rel(g.cpu().numpy(), ref)
            if _is_qk(name):
                cosq[name] = _cos(g.cpu().numpy(), ref)
        wq = max(v for k, v in worst.items() if _is_qk(k))
        wo = max(v for k, v in worst.items() if not _is_qk(k))
        print(f"[{dt}/{tiles}] B={SIZE_B} T={SIZE_T}: worst non-q/k {wo:.2e}; q/k end-to-end {wq:.2e}, min cosine {min(cosq.values()):.6f}; "
              f"d mu {_rel(mu.grad.cpu().numpy(), sc['gmu']):.2e}, d c {_rel(c.grad.cpu().numpy(), sc['gc']):.2e}")
        bad = {k: v for k, v in worst.items() if not _is_qk(k) and v > TOL[dt]}
        assert not bad, bad
        assert _rel(mu.grad.cpu().numpy(), sc["gmu"]) <= TOL[dt]
        assert _rel(c.grad.cpu().numpy(), sc["gc"]) <= TOL[dt]
        # ---- matched inputs at size: attention backward kernels, then RoPE^T + pack + wgrad of conv_q / conv_k / conv_v
        B, T, H = SIZE_B, SIZE_T, 4
        Tp = (T + 63) // 64 * 64
        tt = np.arange(Tp)
        pos = (tt & ~12) | ((tt & 4) << 1) | ((tt & 8) >> 1)
        m = inp["mask"][:, 0].double().numpy()
        bias = (1 - m)[:, None, None, :] * (-1e30)
        for i in (5, 0):
            scale = float(eng.debug_fetch(f"g.scale_a{i}")[0])     # the pass-wide power-of-two scale, re-centred before the attention part
            q = eng.debug_fetch(f"t{i}.q").reshape(B, H, T, 64).astype(np.float64)
            k = eng.debug_fetch(f"t{i}.k").reshape(B, H, T, 64).astype(np.float64)
            v = _native_v(eng, i, B, H, T, Tp, pos)
            dO = (eng.debug_fetch(f"g.dattn_{i}").reshape(B, T, H, 64).transpose(0, 2, 1, 3) / scale).astype(np.float64)
            S2 = q @ k.transpose(0, 1, 3, 2) + bias
            S2 -= S2.max(-1, keepdims=True)
            P = np.exp2(S2); P /= P.sum(-1, keepdims=True)
            dP = dO @ v.transpose(0, 1, 3, 2)
            D = (P * dP).sum(-1, keepdims=True)
            dS = P * (dP - D)
            want = {"dq": dS @ k, "dk": dS.transpose(0, 1, 3, 2) @ q, "dv": P.transpose(0, 1, 3, 2) @ dO}
            got = {}
            for nm, ref in want.items():
                got[nm] = eng.debug_fetch(f"g.{nm}_{i}").reshape(B, H, T, 64).astype(np.float64) / scale
                r = _rel(got[nm], ref)
                print(f"[{dt}/{tiles}] block {i} {nm} at matched inputs: {r:.2e}")
                assert r <= {"f16": 1e-3, "bf16": 8e-3}[dt], (i, nm, r)
            # d q_proj = R^T(dq / 8), d k_proj = R^T(dk ln 2), d v_proj = dv; dW = sum_{n,t} d proj[n,t,co] h1[n,t,ci]
            h1 = eng.debug_fetch(f"t{i}.h1").reshape(B * T, 256).astype(np.float64)
            for nm, dproj in (("q", _rope_T(got["dq"] / 8.0, T)), ("k", _rope_T(got["dk"] * math.log(2.0), T)), ("v", got["dv"])):
                dp = dproj.transpose(0, 2, 1, 3).reshape(B * T, 256)
                name = f"blocks.{i}.block.attn.conv_{nm}"
                r = _rel(params[name + ".weight"].grad[:, :, 0].cpu().numpy(), dp.T @ h1)
                rb = _rel(params[name + ".bias"].grad.cpu().numpy(), dp.sum(0))
                print(f"[{dt}/{tiles}] block {i} conv_{nm} weight / bias gradient from the native d{nm}, h1 (fp64): {r:.2e} / {rb:.2e}")
                assert max(r, rb) <= {"f16": 2e-3, "bf16": 1.2e-2}[dt], (name, r, rb)
        # ---- end to end with MATCHED ATTENTION OPERANDS: the oracle's autograd evaluated at the native forward's own
        # 16-bit q, k, v of every block (straight-through substitution, oracle.attention(subst=...)).  This removes the
        # amplification of the forward's operand rounding by the conditioning of d q, d k and leaves the whole native
        # backward chain -- attention backward, RoPE^T + pack, wgrad, and everything upstream of it -- compared end to end.
        subst = []
        for i in range(6):
            qn = eng.debug_fetch(f"t{i}.q").reshape(B, H, T, 64) * (8.0 / math.log2(math.e))
            kn = eng.debug_fetch(f"t{i}.k").reshape(B, H, T, 64)
            vn = _native_v(eng, i, B, H, T, Tp, pos).astype(np.float32)
            subst.append({"q": torch.from_numpy(np.ascontiguousarray(qn)), "k": torch.from_numpy(np.ascontiguousarray(kn)),
                          "v": torch.from_numpy(np.ascontiguousarray(vn))})
        with torch.enable_grad():
            pr = {k_: v_.clone().requires_grad_(True) for k_, v_ in sd.items()}
            loss2, _ = oracle.compute_loss(pr, sc["x1"], inp["mask"], inp["mu"], inp["c"], sc["t_rand"], sc["z"], qkv_subst=subst)
            loss2.backward()
        wm = {n: _rel(params[n].grad.cpu().numpy(), pr[n].grad.numpy()) for n in params if _is_qk(n)}
        cm = {n: _cos(params[n].grad.cpu().numpy(), pr[n].grad.numpy()) for n in params if _is_qk(n)}
        print(f"[{dt}/{tiles}] q/k gradients vs the oracle evaluated at the native q, k, v: worst {max(wm.values()):.2e}, min cosine {min(cm.values()):.6f}")
        assert max(wm.values()) <= TOL_QK_MATCHED[dt], wm
        # ---- and the end-to-end q / k numbers (module docstring)
        badq = {k: v for k, v in worst.items() if _is_qk(k) and TOL_QK_SIZE[dt] is not None and v > TOL_QK_SIZE[dt]}
        assert not badq, badq
        assert min(cosq.values()) >= COS_QK_SIZE[dt], cosq
    finally:
        eng.debug_capture(False)


def test_gradients_at_the_benchmarked_batch(sd):
    """BASELINE config 5 at the shape bench.py times (`train_step` / `train_ddp`: B = 64 utterances x T = 1000, ragged lengths of
    make_inputs(64, 1000, seed=0, ragged=True), the tile policy as shipped -- no ST_* override): loss, all 116 parameter gradients,
    d mu and d c against ONE pass of the oracle's autograd on the host (fp32 PyTorch-CPU, ~1-2 min on the GPU box's cores; eval mode:
    the dropout masks of a 64 x 4 x 1000 x 1000 attention site are not reproduced in numpy here -- the dropout test above covers the
    masks).  Gates as at B = 4: every non-q/k tensor 3e-3 of max |ref|, d mu / d c 3e-3, loss 5e-4; conv_q / conv_k end to end (module docstring): cosine 0.999
    (measured 3.8e-2, 0.99968)."""
    B, T = 64, 1000
    raw = make_inputs(B, T, seed=0, ragged=True)
    x1 = make_inputs(B, T, seed=1)["z"]
    g0 = torch.Generator().manual_seed(23)
    t_rand = torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)
    dec = _decoder(sd, "f16")
    mu = raw["mu"].cuda().requires_grad_(True)
    c = raw["c"].cuda().requires_grad_(True)
    loss, _ = dec.compute_loss(x1.cuda(), raw["mask"].cuda(), mu, c, t_rand=t_rand.cuda(), z=z.cuda())
    loss.backward()
    torch.cuda.synchronize()
    got = {n: q.grad.detach().cpu().numpy() for n, q in dec.estimator.named_parameters()}
    gmu, gc, lv = mu.grad.cpu().numpy(), c.grad.cpu().numpy(), float(loss.detach())
    del dec
    torch.cuda.empty_cache()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.enable_grad():
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        mu0 = raw["mu"].clone().requires_grad_(True); c0 = raw["c"].clone().requires_grad_(True)
        ref_loss, _ = oracle.compute_loss(pr, x1, raw["mask"], mu0, c0, t_rand, z)
        ref_loss.backward()
    assert abs(lv - float(ref_loss.detach())) <= 5e-4 * float(ref_loss.detach())
    worst = {n: _rel(got[n], pr[n].grad.numpy()) for n in got}
    cosq = {n: _cos(got[n], pr[n].grad.numpy()) for n in got if _is_qk(n)}
    wo = max(v for k, v in worst.items() if not _is_qk(k)); wq = max(v for k, v in worst.items() if _is_qk(k))
    rmu, rc = _rel(gmu, mu0.grad.numpy()), _rel(gc, c0.grad.numpy())
    print(f"[f16] B={B} T={T} ragged ({int(raw['lengths'].sum())} valid frames), shipped tile policy: worst non-q/k {wo:.2e}; q/k end-to-end {wq:.2e}, "
          f"min cosine {min(cosq.values()):.6f}; d mu {rmu:.2e}, d c {rc:.2e}; loss {lv:.6f} vs {float(ref_loss.detach()):.6f}")
    bad = {k: v for k, v in worst.items() if not _is_qk(k) and v > TOL["f16"]}
    assert not bad, bad
    assert rmu <= TOL["f16"] and rc <= TOL["f16"]
    assert min(cosq.values()) >= COS_QK_SIZE["f16"], cosq
    assert all(np.isfinite(v).all() for v in got.values())


def test_v_as_a_hi_lo_operand_pair_in_the_training_forward(sd, size_case, monkeypatch):
    """Round 6: the conv_q / conv_k weight gradients are ill-conditioned in v at random init (near-uniform softmax: dS ~ dO.(v_j - o_i), the
    key-independent bulk of v cancels, its 16-bit error does not; tools/train_qk_split_estimate.py: rounding v moves them by 18 %, rounding the
    projection's input h1 by 11 %, rounding q and k by 0.1 %).  ST_TRAIN_VLO=0: v and h1 as single 16-bit operands (rounds 1-5); 1: v kept as a hi +
    lo pair (attention output = P v_hi + P v_lo); 2 (the default): v = W_v h_hi + W_v h_lo, kept as a pair.  End to end vs the fp32 oracle at B = 4 x
    T = 1000: 2.5e-1 -> 1.7e-1 -> 2.9e-2 (cosine 0.991 -> 0.9965 -> 0.99957); every other gate holds in every mode; deterministic on re-used
    buffers and under dropout."""
    sc = size_case
    inp = sc["inp"]

    def run(vlo):
        monkeypatch.setenv("ST_TRAIN_VLO", str(int(vlo)))
        dec = _decoder(sd, "f16")
        out = None
        for _ in range(2):
            dec.zero_grad()
            mu = inp["mu"].cuda().requires_grad_(True)
            c = inp["c"].cuda().requires_grad_(True)
            loss, _ = dec.compute_loss(sc["x1"].cuda(), inp["mask"].cuda(), mu, c, t_rand=sc["t_rand"].cuda(), z=sc["z"].cuda())
            loss.backward()
            torch.cuda.synchronize()
            got = {n: q.grad.detach().cpu().numpy() for n, q in dec.estimator.named_parameters()}
            if out is not None:
                assert all(np.array_equal(got[n], out[0][n]) for n in got)      # deterministic across steps on re-used buffers
            out = (got, float(loss.detach()), mu.grad.cpu().numpy(), c.grad.cpu().numpy())
        monkeypatch.delenv("ST_TRAIN_VLO", raising=False)
        return out

    res = {}
    for vlo in (0, 1, 2):
        got, lv, gmu, gc = run(vlo)
        assert abs(lv - sc["loss"]) <= 5e-4 * sc["loss"]
        worst = {n: _rel(got[n], sc["grads"][n]) for n in got}
        bad = {k: v for k, v in worst.items() if not _is_qk(k) and v > TOL["f16"]}
        assert not bad, bad
        assert _rel(gmu, sc["gmu"]) <= TOL["f16"] and _rel(gc, sc["gc"]) <= TOL["f16"]
        res[vlo] = (max(v for k, v in worst.items() if _is_qk(k)), min(_cos(got[n], sc["grads"][n]) for n in got if _is_qk(n)),
                    max(v for k, v in worst.items() if not _is_qk(k)))
        print(f"[f16, {('v one operand', 'v hi + lo', 'v hi + lo from h1 hi + lo')[vlo]}] B={SIZE_B} T={SIZE_T}: q/k end-to-end {res[vlo][0]:.2e}, min cosine {res[vlo][1]:.6f}; "
              f"worst non-q/k {res[vlo][2]:.2e}")
    assert res[1][0] < 0.85 * res[0][0] and res[1][1] > res[0][1] and res[1][1] >= 0.995
    assert res[2][0] < 0.25 * res[0][0] and res[2][0] <= TOL_QK_SIZE["f16"] and res[2][1] >= COS_QK_SIZE["f16"]
    # train mode (dropout masks) under the mode: finite, deterministic for a fixed torch seed
    dec = _decoder(sd, "f16", train=True)
    vals = []
    for _ in range(2):
        torch.manual_seed(7)
        dec.zero_grad()
        loss, _ = dec.compute_loss(sc["x1"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
        loss.backward()
        torch.cuda.synchronize()
        g = {n: q.grad.detach().cpu().numpy().copy() for n, q in dec.estimator.named_parameters()}
        assert all(np.isfinite(v).all() for v in g.values())
        vals.append((float(loss.detach()), g))
    assert vals[0][0] == vals[1][0] and all(np.array_equal(vals[0][1][n], vals[1][1][n]) for n in vals[0][1])


@pytest.mark.parametrize("B,T,lengths", [(96, 64, None), (2, 2500, [2500, 1733]), (5, 333, [333, 332, 97, 32, 1])])
def test_gradients_at_shapes_off_the_benchmarks(sd, B, T, lengths):
    """The training path away from the BASELINE shapes: many short utterances (96 items: per-item tables, reduction chunks and the K split
    of the weight-gradient GEMMs at an item count no benchmark has), a paragraph-length pair (T = 2500: 40 key tiles per attention row, 79
    reduction chunks per item), and odd lengths down to ONE frame.  Eval mode, f16 operands, against one pass of the oracle's autograd;
    gates as at the benchmark sizes (non-q/k tensors, d mu, d c: 3e-3 of max |ref|; loss 5e-4; conv_q / conv_k by direction)."""
    raw = make_inputs(B, T, seed=33, lengths=lengths, ragged=lengths is None)
    x1 = make_inputs(B, T, seed=34)["z"]
    g0 = torch.Generator().manual_seed(29)
    t_rand = torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)
    dec = _decoder(sd, "f16")
    mu = raw["mu"].cuda().requires_grad_(True)
    c = raw["c"].cuda().requires_grad_(True)
    loss, _ = dec.compute_loss(x1.cuda(), raw["mask"].cuda(), mu, c, t_rand=t_rand.cuda(), z=z.cuda())
    loss.backward()
    torch.cuda.synchronize()
    got = {n: q.grad.detach().cpu().numpy() for n, q in dec.estimator.named_parameters()}
    gmu, gc, lv = mu.grad.cpu().numpy(), c.grad.cpu().numpy(), float(loss.detach())
    with torch.enable_grad():
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        mu0 = raw["mu"].clone().requires_grad_(True); c0 = raw["c"].clone().requires_grad_(True)
        ref_loss, _ = oracle.compute_loss(pr, x1, raw["mask"], mu0, c0, t_rand, z)
        ref_loss.backward()
    assert abs(lv - float(ref_loss.detach())) <= 5e-4 * float(ref_loss.detach())
    worst = {n: _rel(got[n], pr[n].grad.numpy()) for n in got}
    cosq = {n: _cos(got[n], pr[n].grad.numpy()) for n in got if _is_qk(n)}
    wo = max(v for k, v in worst.items() if not _is_qk(k)); wq = max(v for k, v in worst.items() if _is_qk(k))
    rmu, rc = _rel(gmu, mu0.grad.numpy()), _rel(gc, c0.grad.numpy())
    print(f"[f16] B={B} T={T} ({int(raw['lengths'].sum())} valid frames): worst non-q/k {wo:.2e}; q/k end-to-end {wq:.2e}, min cosine {min(cosq.values()):.6f}; "
          f"d mu {rmu:.2e}, d c {rc:.2e}")
    bad = {k: v for k, v in worst.items() if not _is_qk(k) and v > TOL["f16"]}
    assert not bad, bad
    assert rmu <= TOL["f16"] and rc <= TOL["f16"]
    assert min(cosq.values()) >= 0.998 and wq <= TOL_QK_SIZE["f16"], (wq, cosq)      # (2 x 2500: 3.9e-2, cosine 0.99936)
    assert all(np.isfinite(v).all() for v in got.values())


# ---------------------------------------------------------------- K optimizer steps vs the fp32 oracle
TRAJ_B, TRAJ_T, TRAJ_LENS, TRAJ_K = 4, 250, [250, 231, 188, 120], 12


def _traj_draws(k):
    g0 = torch.Generator().manual_seed(7000 + k)
    x1 = torch.randn(TRAJ_B, 128, TRAJ_T, generator=g0)
    t_rand = torch.rand(TRAJ_B, 1, 1, generator=g0)
    z = torch.randn(TRAJ_B, 128, TRAJ_T, generator=g0)
    return x1, t_rand, z


@pytest.fixture(scope="module")
def oracle_trajectories(sd):
    """K AdamW steps of oracle.compute_loss (fp32 PyTorch-CPU autograd: models/flow_matching.py:69-100 + train.py:78-82's
    zero_grad / backward / step) from the seeded weights, dropout off, fixed per-step draws; for the reference's learning rate
    (config.py:37: 1e-4) and for 10x it (ten times the parameter drift)."""
    inp = make_inputs(TRAJ_B, TRAJ_T, seed=101, lengths=TRAJ_LENS)
    out = {}
    for lr in (1e-4, 1e-3):
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        opt = torch.optim.AdamW(list(pr.values()), lr=lr)
        losses = []
        for k in range(TRAJ_K):
            x1, t_rand, z = _traj_draws(k)
            opt.zero_grad()
            with torch.enable_grad():
                loss, _ = oracle.compute_loss(pr, x1, inp["mask"], inp["mu"], inp["c"], t_rand, z)
                loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        out[lr] = dict(losses=losses, params={k: v.detach().numpy().copy() for k, v in pr.items()})
    return dict(inp=inp, runs=out)


@pytest.mark.parametrize("lr", [1e-4, 1e-3])
def test_training_trajectory_matches_the_fp32_oracle(sd, oracle_trajectories, lr):
    """Config 5's end-to-end statement: K = 12 AdamW steps through the NATIVE forward / backward (f16 operands, the shipping
    type; in-place weight updates re-packed on the stream) against the same K steps through the fp32 oracle on the same
    draws.  What is asserted: per step the loss agrees to 1e-3; after K steps the accumulated UPDATE theta_K - theta_0 of EVERY
    parameter tensor -- conv_q / conv_k, whose single-step gradient is ill-conditioned at random init, included -- points the
    oracle's way (cosine >= 0.995) and no single element is more than 3 lr off the oracle's value.  (Not asserted, printed: the
    max-norm difference of the parameters and of the updates -- Adam normalises each element's step to ~lr, so an element whose
    tiny gradient flips sign moves by 2 lr per step whatever the backward's accuracy; for small-valued tensors that is up to 9e-3
    of max |theta|)"""
    inp = oracle_trajectories["inp"]
    ref = oracle_trajectories["runs"][lr]
    dec = _decoder(sd, "f16")                      # eval mode: dropout off (the masks are covered by their own test)
    opt = torch.optim.AdamW(dec.parameters(), lr=lr)
    mask, mu, c = inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()
    losses = []
    for k in range(TRAJ_K):
        x1, t_rand, z = _traj_draws(k)
        opt.zero_grad()
        loss, _ = dec.compute_loss(x1.cuda(), mask, mu, c, t_rand=t_rand.cuda(), z=z.cuda())
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    rel_loss = [abs(a - b) / b for a, b in zip(losses, ref["losses"])]
    print(f"[lr={lr:g}] loss oracle {ref['losses'][0]:.5f} -> {ref['losses'][-1]:.5f}, native {losses[0]:.5f} -> {losses[-1]:.5f}; "
          f"worst per-step rel diff {max(rel_loss):.2e}")
    assert max(rel_loss) <= 1e-3, rel_loss
    worst, cosw, upd = {}, {}, {}
    for name, p in dec.estimator.named_parameters():
        a, b, p0 = p.detach().cpu().numpy(), ref["params"][name], sd[name].numpy()
        worst[name] = _rel(a, b)
        cosw[name] = _cos(a - p0, b - p0)
        upd[name] = _rel(a - p0, b - p0)
    qk = [n for n in worst if _is_qk(n)]
    print(f"[lr={lr:g}] after {TRAJ_K} steps: worst parameter diff {max(worst.values()):.2e} (q/k {max(worst[n] for n in qk):.2e}); update "
          f"cosine min {min(cosw.values()):.4f} (q/k {min(cosw[n] for n in qk):.4f}); update max-norm diff worst {max(upd.values()):.2e} "
          f"(q/k {max(upd[n] for n in qk):.2e})")
    steps_off = {n: float(np.abs(p.detach().cpu().numpy() - ref["params"][n]).max()) / lr for n, p in dec.estimator.named_parameters()}
    print(f"[lr={lr:g}] worst element deviation {max(steps_off.values()):.2f} lr (q/k {max(steps_off[n] for n in qk):.2f} lr)")
    badc = {k: v for k, v in cosw.items() if v < 0.995}
    assert not badc, badc
    bads = {k: v for k, v in steps_off.items() if v > 3.0}
    assert not bads, bads


def test_training_trajectory_at_config5_frame_count(sd):
    """The same statement at T = 1000 (the round-5 review: the T = 250 trajectory sees attention 4x less peaky, and the single-step
    conv_q / conv_k gradients were 22 % off end to end at T = 1000 then, 3 % since round 6): K = 8 AdamW steps at the reference's lr = 1e-4, B = 4 x T = 1000
    ragged, native f16 against the fp32 oracle on the same draws.  Asserted as at T = 250: per-step loss within 1e-3, the accumulated
    update of EVERY tensor -- conv_q / conv_k included -- within cosine 0.99 of the oracle's, no element more than 3 lr off."""
    B, T, K, lr = 4, 1000, 8, 1e-4
    inp = make_inputs(B, T, seed=111, lengths=[1000, 873, 655, 512])

    def draws(k):
        g0 = torch.Generator().manual_seed(9000 + k)
        return torch.randn(B, 128, T, generator=g0), torch.rand(B, 1, 1, generator=g0), torch.randn(B, 128, T, generator=g0)

    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(pr.values()), lr=lr)
    ref_losses = []
    for k in range(K):
        x1, t_rand, z = draws(k)
        opt.zero_grad()
        with torch.enable_grad():
            loss, _ = oracle.compute_loss(pr, x1, inp["mask"], inp["mu"], inp["c"], t_rand, z)
            loss.backward()
        opt.step()
        ref_losses.append(float(loss.detach()))
    dec = _decoder(sd, "f16")
    opt2 = torch.optim.AdamW(dec.parameters(), lr=lr)
    mask, mu, c = inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()
    losses = []
    for k in range(K):
        x1, t_rand, z = draws(k)
        opt2.zero_grad()
        loss, _ = dec.compute_loss(x1.cuda(), mask, mu, c, t_rand=t_rand.cuda(), z=z.cuda())
        loss.backward()
        opt2.step()
        losses.append(float(loss.detach()))
    rel_loss = [abs(a - b) / b for a, b in zip(losses, ref_losses)]
    cosw, steps_off = {}, {}
    for name, p in dec.estimator.named_parameters():
        a, b, p0 = p.detach().cpu().numpy(), pr[name].detach().numpy(), sd[name].numpy()
        cosw[name] = _cos(a - p0, b - p0)
        steps_off[name] = float(np.abs(a - b).max()) / lr
    qk = [n for n in cosw if _is_qk(n)]
    print(f"[T=1000, lr={lr:g}, {K} steps] worst per-step loss diff {max(rel_loss):.2e}; update cosine min {min(cosw.values()):.4f} "
          f"(q/k {min(cosw[n] for n in qk):.4f}); worst element deviation {max(steps_off.values()):.2f} lr (q/k {max(steps_off[n] for n in qk):.2f} lr)")
    assert max(rel_loss) <= 1e-3, rel_loss
    badc = {k: v for k, v in cosw.items() if v < 0.99}
    assert not badc, badc
    bads = {k: v for k, v in steps_off.items() if v > 3.0}
    assert not bads, bads


def test_gradients_with_trained_like_weights(size_case):
    """Weights that look TRAINED rather than initialised: adaLN gates of O(1) (ada_std 0.15) and q / k projections scaled 6x
    (peaky attention), B = 4 x T = 1000 ragged, shipping dtype.  Two statements:
      1. The native BACKWARD is right: every gradient -- all 116 tensors, conv_q / conv_k included -- agrees with the oracle's
         autograd evaluated at the native forward's own q, k, v to 1.5e-2 (measured 6e-3).  (Round 3's backward overflowed f16
         inside a block here -- the gradient grows ~350x between two LayerNorm backwards -- and returned NaN; the scale is now
         re-centred after each of them.)
      2. Why the comparison is made at matched attention operands: this high-gain random network is CHAOTIC.  Rounding the fp32
         oracle's own q, k, v to f16 -- nothing native involved -- moves its own gradients by tens of percent (asserted > 5 %,
         measured O(1)), so an end-to-end number against the un-perturbed fp32 oracle measures the network's conditioning, not the
         kernels.  (At the seeded init the same end-to-end comparison holds to 1e-3 for every tensor but q / k, and the K-step
         trajectory test above holds for all of them.)"""
    sc = size_case
    sd2 = oracle.make_state_dict(1234, ada_std=0.15)
    for i in range(6):
        for nm in ("q", "k"):
            sd2[f"blocks.{i}.block.attn.conv_{nm}.weight"] = sd2[f"blocks.{i}.block.attn.conv_{nm}.weight"] * 6.0
    inp = sc["inp"]
    B, T, H = SIZE_B, SIZE_T, 4
    Tp = (T + 63) // 64 * 64
    tt = np.arange(Tp)
    pos = (tt & ~12) | ((tt & 4) << 1) | ((tt & 8) >> 1)

    def oracle_grads(subst=None):
        with torch.enable_grad():
            pr = {k: v.clone().requires_grad_(True) for k, v in sd2.items()}
            loss, _ = oracle.compute_loss(pr, sc["x1"], inp["mask"], inp["mu"], inp["c"], sc["t_rand"], sc["z"],
                                          **({"qkv_subst": subst} if subst is not None else {}))
            loss.backward()
        return float(loss.detach()), {k: v.grad.numpy() for k, v in pr.items()}

    loss_ref, g_ref = oracle_grads()
    # (2) the oracle's own q, k, v rounded to f16 (taps of its fp32 forward), straight-through
    taps = {}
    with torch.no_grad():
        t = 1 - torch.cos(sc["t_rand"] * 0.5 * torch.pi)
        y = (1 - (1 - 1e-4) * t) * sc["z"] + t * sc["x1"]
        oracle.decoder_forward(sd2, t.squeeze(), y, inp["mask"], inp["mu"], inp["c"], taps=taps)
    sub16 = [{k: taps[f"b{i}.{k}"].half().float() for k in ("q", "k", "v")} for i in range(6)]
    _, g_16 = oracle_grads(sub16)
    cond = {n: _rel(g_16[n], g_ref[n]) for n in g_ref}
    print(f"trained-like weights: fp32 oracle vs the SAME oracle with its q, k, v rounded to f16: gradients move by up to "
          f"{max(cond.values()):.2e} (median over tensors {float(np.median(list(cond.values()))):.2e})")
    assert max(cond.values()) > 5e-2
    # (1) native backward vs the oracle at the native q, k, v
    dec = _decoder(sd2, "f16")
    eng = dec.estimator.engine()
    eng.debug_capture(True)
    try:
        loss, _ = dec.compute_loss(sc["x1"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda(), t_rand=sc["t_rand"].cuda(), z=sc["z"].cuda())
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.detach()) - loss_ref) <= 1e-3 * loss_ref
        subst = []
        for i in range(6):
            qn = eng.debug_fetch(f"t{i}.q").reshape(B, H, T, 64) * (8.0 / math.log2(math.e))
            kn = eng.debug_fetch(f"t{i}.k").reshape(B, H, T, 64)
            vn = _native_v(eng, i, B, H, T, Tp, pos).astype(np.float32)
            subst.append({"q": torch.from_numpy(np.ascontiguousarray(qn)), "k": torch.from_numpy(np.ascontiguousarray(kn)),
                          "v": torch.from_numpy(np.ascontiguousarray(vn))})
    finally:
        eng.debug_capture(False)
    _, g_m = oracle_grads(subst)
    with torch.no_grad():       # how far the 16-bit FORWARD is from fp32 in this regime (one evaluation at t = 0.5)
        tq = torch.tensor(0.5)
        fr = oracle.decoder_forward(sd2, tq, sc["z"], inp["mask"], inp["mu"], inp["c"])
        fn = dec.estimator(tq.cuda(), sc["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
    print(f"trained-like weights: one forward evaluation, native f16 vs fp32 oracle: {_rel(fn.numpy(), fr.numpy()):.2e}")
    params = dict(dec.estimator.named_parameters())
    assert all(torch.isfinite(p.grad).all() for p in params.values())
    worst = {n: _rel(params[n].grad.cpu().numpy(), g_m[n]) for n in params}
    e2e = {n: _rel(params[n].grad.cpu().numpy(), g_ref[n]) for n in params}
    print(f"trained-like weights: native vs the oracle at the native q, k, v: worst {max(worst.values()):.2e} (q/k "
          f"{max(v for k, v in worst.items() if _is_qk(k)):.2e}); vs the un-perturbed fp32 oracle {max(e2e.values()):.2e} (conditioning, see (2))")
    bad = {k: v for k, v in worst.items() if v > 1.5e-2}
    assert not bad, bad


@pytest.mark.parametrize("dt", ["f16"])
def test_gin_channels_not_equal_hidden(dt):
    """gin_channels != hidden_channels adds a Linear(gin, hidden) in front of every block's adaLN modulation
    (models/diffusion_transformer.py:92-96): one evaluation and every gradient -- the extra linears' and d loss / d c through
    them included -- against the oracle's autograd."""
    from stabletts_amd.flow_matching import CFMDecoder
    cfg = oracle.DecoderConfig(gin_channels=128)
    sdg = oracle.make_state_dict(4242, cfg)
    assert "blocks.0.block.adaLN_modulation.0.weight" in sdg
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 128, operand_dtype=dt)
    dec.estimator.load_state_dict(sdg)
    dec = dec.cuda().eval()
    B, T, lengths = 2, 52, [52, 37]
    inp = make_inputs(B, T, seed=91, lengths=lengths, gin=128)
    x1 = make_inputs(B, T, seed=92, gin=128)["z"]
    g0 = torch.Generator().manual_seed(23)
    t_rand = torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)
    with torch.no_grad():
        tt = torch.tensor(0.37)
        ref = oracle.decoder_forward(sdg, tt, inp["z"], inp["mask"], inp["mu"], inp["c"])
        out = dec.estimator(tt.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda()).cpu()
    assert _rel(out.numpy(), ref.numpy()) <= 1e-3
    c = inp["c"].cuda().requires_grad_(True)
    loss, _ = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), inp["mu"].cuda(), c, t_rand=t_rand.cuda(), z=z.cuda())
    loss.backward()
    pr = {k: v.clone().requires_grad_(True) for k, v in sdg.items()}
    cr = inp["c"].clone().requires_grad_(True)
    lref, _ = oracle.compute_loss(pr, x1, inp["mask"], inp["mu"], cr, t_rand, z)
    lref.backward()
    assert abs(float(loss.detach()) - float(lref.detach())) <= 5e-4 * float(lref.detach())
    params = dict(dec.estimator.named_parameters())
    assert set(params) == set(pr)
    worst = {n: _rel(params[n].grad.cpu().numpy(), pr[n].grad.numpy()) for n in params}
    # (q / k projections: conditioning-limited end to end at random init, gated by the matched-operand chain of the size test)
    bad = {k: v for k, v in worst.items() if v > (2e-1 if _is_qk(k) else TOL[dt])}
    print(f"[gin=128 {dt}] worst non-q/k {max(v for k, v in worst.items() if not _is_qk(k)):.2e}; adaLN.0 "
          f"{max(v for k, v in worst.items() if 'adaLN_modulation.0' in k):.2e}; d c {_rel(c.grad.cpu().numpy(), cr.grad.numpy()):.2e}")
    assert not bad, bad
    assert _rel(c.grad.cpu().numpy(), cr.grad.numpy()) <= TOL[dt]


def test_backward_of_replaced_activations_fails_loudly(sd):
    """The engine keeps the activations of ONE grad-enabled forward: a backward whose activations were
    replaced by a later forward must raise (autograd Function) / return ST_ERR_STATE (C ABI) before any memory is
    touched -- not silently use the other forward's activations, whose (B, T) may differ."""
    from stabletts_amd._lib import NativeError, ST_ERR_STATE
    dec = _decoder(sd, "f16")
    a = make_inputs(2, 40, seed=61, lengths=[40, 31])
    b = make_inputs(1, 72, seed=62)
    la, _ = dec.compute_loss(make_inputs(2, 40, seed=63)["z"].cuda(), a["mask"].cuda(), a["mu"].cuda(), a["c"].cuda())
    lb, _ = dec.compute_loss(make_inputs(1, 72, seed=64)["z"].cuda(), b["mask"].cuda(), b["mu"].cuda(), b["c"].cuda())
    with pytest.raises(RuntimeError, match="activations of ONE"):
        la.backward()                                # la's activations were replaced by lb's forward
    assert all(p.grad is None for p in dec.estimator.parameters())
    lb2, _ = dec.compute_loss(make_inputs(1, 72, seed=64)["z"].cuda(), b["mask"].cuda(), b["mu"].cuda(), b["c"].cuda())
    lb2.backward()                                   # the newest forward's backward is fine
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in dec.estimator.parameters())
    # C ABI: wrong serial / wrong shape / after a parameter update
    eng = dec.estimator.engine()
    g = torch.zeros(1, 128, 72, device="cuda")
    serial = eng.train_serial()
    assert serial > 0
    for ser, gg in ((serial + 1, g), (serial - 1, g), (serial, torch.zeros(1, 128, 71, device="cuda")), (serial, torch.zeros(2, 128, 72, device="cuda"))):
        with pytest.raises(NativeError) as ei:
            eng.train_backward(ser, gg, None, None, None, torch.cuda.current_stream().cuda_stream)
        assert ei.value.code == ST_ERR_STATE
    eng.train_backward(serial, g, None, None, None, torch.cuda.current_stream().cuda_stream)     # the matching one is accepted
    with torch.no_grad():
        dec.estimator.final_proj.bias.add_(0.5)      # version bump -> re-pack -> the held activations are invalidated
    lc, _ = dec.compute_loss(make_inputs(2, 40, seed=63)["z"].cuda(), a["mask"].cuda(), a["mu"].cuda(), a["c"].cuda())
    with torch.no_grad():
        dec.estimator.final_proj.bias.add_(0.5)
    dec.estimator.engine()                           # ... picked up by any later use of the engine
    with pytest.raises(RuntimeError, match="activations of ONE"):
        lc.backward()


@pytest.mark.parametrize("qkv_ws", [False, True, "big_tiles"])
def test_optimizer_steps_repack_in_place_without_reallocation(sd, monkeypatch, qkv_ws):
    """Every optimizer step bumps the parameters' version counters (train.py:82).  The engine reads the fp32 tensors
    where torch keeps them (st_bind_param) and re-packs its 16-bit copies on the stream (st_repack): device bytes and
    the time-step count stay put, and the result equals a fresh decoder loaded with the updated weights bit for bit.
    qkv_ws: the weight-stationary q/k/v kernel forced at this small size (it reads its own fragment-ordered copy of the
    weight, re-packed by the same job list: PackJob kind 4) against a fresh decoder on the generic tile.
    "big_tiles": both engines on the big-grid kernels at this small size (ST_BIG_MIN_BLOCKS=1) -- the evaluation goes through the
    fused Winograd FFN, whose weight stream (three transformed planes per tap triple) is re-packed by PackJob kind 5."""
    from stabletts_amd.flow_matching import CFMDecoder
    big = qkv_ws == "big_tiles"
    qkv_ws = qkv_ws is True
    if big:
        monkeypatch.setenv("ST_BIG_MIN_BLOCKS", "1")
    if qkv_ws:
        monkeypatch.setenv("ST_QKV_WS_MIN_TILES", "1")
    dec = _decoder(sd, "f16", train=True)
    dec.estimator.engine()
    if qkv_ws:
        monkeypatch.setenv("ST_QKV_WS", "0")          # every engine created from here on: the generic tile
    opt = torch.optim.AdamW(dec.parameters(), lr=1e-3)
    inp = make_inputs(2, 64, seed=71, lengths=[64, 50])
    x1 = make_inputs(2, 64, seed=72)["z"].cuda()
    args = (inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    eng = dec.estimator.engine()
    handle, nbytes = eng.handle.value, None
    for step in range(3):
        torch.manual_seed(step)
        loss, _ = dec.compute_loss(x1, *args)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        assert dec.estimator.engine().handle.value == handle
        if nbytes is None:
            nbytes = eng.device_bytes()
        assert eng.device_bytes() == nbytes          # nothing is freed / re-allocated by a parameter update
    assert dec.estimator._staging is None            # fp32 parameters are read in place: no copies
    dec.eval()
    t = torch.tensor(0.35).cuda()
    with torch.no_grad():
        got = dec.estimator(t, inp["z"].cuda(), *args)
        fresh = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
        fresh.estimator.load_state_dict({k: v.detach().cpu() for k, v in dec.estimator.state_dict().items()})
        want = fresh.cuda().eval().estimator(t, inp["z"].cuda(), *args)
    assert torch.equal(got, want)
    # a module whose parameters are not fp32 (.half()) is served from fp32 staging copies
    with torch.no_grad():
        h = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
        h.estimator.load_state_dict(sd)
        h = h.cuda().half().eval()
        outh = h.estimator(t, inp["z"].cuda(), *args)
        assert h.estimator._staging is not None and outh.dtype == torch.float32
        r = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
        r.estimator.load_state_dict({k: v.half().float() for k, v in sd.items()})
        assert torch.equal(outh, r.cuda().eval().estimator(t, inp["z"].cuda(), *args))


# ---- python port of the counter-based dropout hash (stabletts_amd/csrc/common.h: drop_hash, train_kernels.hip: make_drop)
_M = np.uint64(0xFFFFFFFF)


def _mix32(h):
    h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x85ebca6b)) & _M
    h = h ^ (h >> np.uint64(13)); h = (h * np.uint64(0xc2b2ae35)) & _M
    return h ^ (h >> np.uint64(16))


def _pair32(x):                     # drop_pair (csrc/common.h): one odd multiply + xor-shift of rowh ^ colh
    h = (x * np.uint64(0x9E3779B1)) & _M
    return h ^ (h >> np.uint64(15))


def _seed_words(seed, salt):
    s64 = (seed * 0x100000001B3 + (salt + 1) * 0xD6E8FEB86659FD93) % (1 << 64)        # make_drop (train_kernels.hip)
    return np.uint64(s64 & 0xFFFFFFFF), np.uint64(s64 >> 32)


def _keep16(h, odd, p):
    """Low / high 16 bits of a pair hash against thresh16 = round(p * 2^16): kept -> 1 / (1 - p), dropped -> 0 (DropCfg, launch.h)."""
    t16 = np.uint64(min(max(int(p * 65536.0 + 0.5), 1), 65535))
    bits = np.where(odd, h >> np.uint64(16), h & np.uint64(0xFFFF))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(bits >= t16, scale, np.float32(0.0)).astype(np.float32)


def _drop_ffn(seed, salt, p, idx):
    """FFN site, element index idx: h = mix32(((u32)seed ^ (idx >> 1) * 0x9E3779B1) + (u32)(seed >> 32)), half = idx & 1."""
    lo, hi = _seed_words(seed, salt)
    idx = idx.astype(np.uint64)
    h = _mix32(((lo ^ (((idx >> np.uint64(1)) * np.uint64(0x9E3779B1)) & _M)) + hi) & _M)
    return _keep16(h, (idx & np.uint64(1)) == 1, p)


def _drop_attn(seed, salt, p, row, key):
    """Attention site (row = (item * H + head) * T + query, key): pair(rowh[row], colh[key >> 1]), half = key & 1."""
    lo, hi = _seed_words(seed, salt)
    row = row.astype(np.uint64); key = key.astype(np.uint64)
    rh = _mix32(lo ^ ((row * np.uint64(0x9E3779B1)) & _M))
    ch = _mix32(hi ^ (((key >> np.uint64(1)) * np.uint64(0x85ebca77)) & _M))
    return _keep16(_pair32(rh ^ ch), (key & np.uint64(1)) == 1, p)


@pytest.mark.parametrize("dt,T,lengths,tiles", [("f16", 70, [70, 45], "policy"), ("f16", 250, [250, 181], "big")])
def test_dropout_forward_and_backward_match_oracle_with_same_masks(sd, monkeypatch, dt, T, lengths, tiles):
    """Train mode (p_dropout = 0.1 on the FFN activations and the attention probabilities,
    diffusion_transformer.py:28,77): the native counter-based dropout is reproduced in numpy, the oracle is run under
    autograd with exactly those keep / (1 - p) factors, and loss + gradients must agree as in eval mode.  Also:
    same torch seed -> bitwise identical loss; different seed -> different masks.
    "big": ST_BIG_MIN_BLOCKS=1 puts every conv on the tiles of a B = 64 batch, where the FFN's SiLU + dropout + mask step
    runs inside the phased kernel's epilogue (EPI_SILU, forward and dgrad); that case is additionally compared BITWISE with
    the stand-alone silu_drop / silu_bwd kernels (ST_FUSE_SILU=0)."""
    B, p = 2, 0.1
    if tiles == "big":
        monkeypatch.setenv("ST_BIG_MIN_BLOCKS", "1")
    dec = _decoder(sd, dt, train=True)
    dec.estimator.engine()
    inp = make_inputs(B, T, seed=51, lengths=lengths)
    x1 = make_inputs(B, T, seed=52)["z"]
    g0 = torch.Generator().manual_seed(9)
    t_rand = torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)

    def native(seed):
        torch.manual_seed(seed)
        dec.zero_grad()
        mu = inp["mu"].cuda().requires_grad_(True)
        loss, _ = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), mu, inp["c"].cuda(), t_rand=t_rand.cuda(), z=z.cuda())
        loss.backward()
        return float(loss.detach()), mu.grad.cpu(), {n: q.grad.detach().cpu().clone() for n, q in dec.estimator.named_parameters()}

    l1, gmu1, gp1 = native(123)
    l1b, _, _ = native(123)
    l2, _, _ = native(124)
    assert l1 == l1b and l1 != l2
    if tiles == "big":
        monkeypatch.setenv("ST_FUSE_SILU", "0")
        fused = dec
        dec = _decoder(sd, dt, train=True)
        l3, gmu3, gp3 = native(123)
        dec = fused
        monkeypatch.delenv("ST_FUSE_SILU")
        monkeypatch.delenv("ST_BIG_MIN_BLOCKS")
        assert l3 == l1 and torch.equal(gmu3, gmu1)
        assert all(torch.equal(gp3[n], gp1[n]) for n in gp1)
    torch.manual_seed(123)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    F_, H = 1024, 4
    drop = {"ffn": [], "attn": []}
    n_, t_, c_ = np.meshgrid(np.arange(B), np.arange(T), np.arange(F_), indexing="ij")
    idx = ((n_ * T + t_) * F_ + c_).astype(np.uint64)
    nn, hh, qq, kk = np.meshgrid(np.arange(B), np.arange(H), np.arange(T), np.arange(T), indexing="ij")
    for i in range(6):
        f = _drop_ffn(seed, 2 * i, p, idx)                                                    # [B][T][F] time-major
        drop["ffn"].append(torch.from_numpy(f).permute(0, 2, 1).contiguous())
        a = _drop_attn(seed, 2 * i + 1, p, ((nn * H + hh) * T + qq).astype(np.uint64), kk.astype(np.uint64))
        drop["attn"].append(torch.from_numpy(a))
    assert 0.85 < float((drop["ffn"][0] > 0).float().mean()) < 0.95
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    mu = inp["mu"].clone().requires_grad_(True)
    t = 1 - torch.cos(t_rand * 0.5 * torch.pi)
    y = (1 - (1 - 1e-4) * t) * z + t * x1
    u = x1 - (1 - 1e-4) * z
    pred = oracle.decoder_forward(pr, t.squeeze(), y, inp["mask"], mu, inp["c"], drop=drop)
    loss = torch.nn.functional.mse_loss(pred, u, reduction="sum") / (inp["mask"].sum() * 128)
    loss.backward()
    assert abs(l1 - float(loss.detach())) <= 1e-3 * float(loss.detach())
    assert _rel(gmu1.numpy(), mu.grad.numpy()) <= TOL[dt]
    worst = {n: _rel(gp1[n].numpy(), pr[n].grad.numpy()) for n in gp1}
    bad = {k: v for k, v in worst.items() if v > (TOL_QK if _is_qk(k) else TOL)[dt]}
    print(f"[dropout {dt}] worst non-q/k {max(v for k, v in worst.items() if not _is_qk(k)):.2e}, q/k {max(v for k, v in worst.items() if _is_qk(k)):.2e}")
    assert not bad, bad


@pytest.mark.parametrize("B,T,lengths", [(3, 300, [300, 211, 64]), (8, 1000, None)])
def test_side_streams_are_bitwise_neutral(sd, monkeypatch, B, T, lengths):
    """Round 6: the backward's weight-gradient GEMMs (+ their reductions) run on a side stream and each block's gradient-independent
    attention operand copies on a second one (csrc/engine_train.cpp: wgrad_side / attn_prep); every operand a weight gradient reads has
    its own buffer and the parts join the streams before they return.  The SAME kernels run either way, so loss and every gradient must
    be bit-identical to the single-stream order (ST_TRAIN_SIDE=0) -- a missed dependency would show up here as a difference.  Train
    mode (dropout tables shared by forward and backward), two steps each so that buffers are re-used across backwards."""
    inp = make_inputs(B, T, seed=71, lengths=lengths, ragged=lengths is None)
    x1 = make_inputs(B, T, seed=72)["z"]
    g0 = torch.Generator().manual_seed(11)
    t_rand = torch.rand(B, 1, 1, generator=g0); z = torch.randn(B, 128, T, generator=g0)

    def run(side):
        if side:
            monkeypatch.delenv("ST_TRAIN_SIDE", raising=False)
        else:
            monkeypatch.setenv("ST_TRAIN_SIDE", "0")
        dec = _decoder(sd, "f16", train=True)
        outs = []
        for step in range(2):
            torch.manual_seed(500 + step)
            dec.zero_grad()
            mu = inp["mu"].cuda().requires_grad_(True)
            c = inp["c"].cuda().requires_grad_(True)
            loss, _ = dec.compute_loss(x1.cuda(), inp["mask"].cuda(), mu, c, t_rand=t_rand.cuda(), z=z.cuda())
            loss.backward()
            torch.cuda.synchronize()
            outs.append((float(loss.detach()), mu.grad.cpu(), c.grad.cpu(), {n: q.grad.detach().cpu().clone() for n, q in dec.estimator.named_parameters()}))
        return outs

    a, b = run(True), run(False)
    for (la, gma, gca, gpa), (lb, gmb, gcb, gpb) in zip(a, b):
        assert la == lb and torch.equal(gma, gmb) and torch.equal(gca, gcb)
        diff = [n for n in gpa if not torch.equal(gpa[n], gpb[n])]
        assert not diff, diff
        assert all(torch.isfinite(v).all() for v in gpa.values())


def test_inference_mode_and_frozen_parameters_take_the_inference_path(sd):
    """No autograd graph when nothing requires grad; the training path when a leaf does (mu only, frozen decoder)."""
    dec = _decoder(sd, "f16")
    inp = make_inputs(1, 40, seed=3)
    t = torch.tensor(0.4)
    args = (t, inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    with torch.no_grad():
        a = dec.estimator(*args)
    assert not a.requires_grad
    dec.requires_grad_(False)
    b = dec.estimator(*args)
    assert not b.requires_grad and torch.equal(a, b)
    mu = inp["mu"].cuda().requires_grad_(True)
    c = dec.estimator(t, inp["z"].cuda(), inp["mask"].cuda(), mu, inp["c"].cuda())
    assert c.requires_grad
    c.square().sum().backward()
    assert mu.grad is not None and torch.isfinite(mu.grad).all() and float(mu.grad.abs().max()) > 0
    assert all(p.grad is None for p in dec.parameters())
    assert float((c.detach() - a).abs().max() / a.abs().max()) <= 2e-3     # unfused training forward vs fused inference forward


def test_ddp_two_ranks_match_single_process(tmp_path):
    """BASELINE config 5 in miniature (train.py:49-51,78-81): DistributedDataParallel around the native decoder,
    2 processes, K optimizer steps; per-step loss must match one process training on the concatenated batch (equal
    lengths => the mean of the two rank losses is the full-batch loss, DDP's averaged gradient the full-batch
    gradient).  The box has one GPU, so both ranks share it and the gradient all-reduce runs over gloo (NCCL / RCCL
    refuses two ranks on one device); with one GPU per rank the same script runs with --backend nccl (RCCL)."""
    out2, out1 = tmp_path / "ddp.pt", tmp_path / "one.pt"
    port = 29700 + (os.getpid() % 200)
    env = dict(os.environ, BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "train_ddp.py"), "--out", str(out2), "--backend", "gloo"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_ddp.py"), "--out", str(out1)],
                        env=dict(os.environ), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    a, b = torch.load(out2), torch.load(out1)
    assert a["world"] == 2 and b["world"] == 1 and len(a["losses"]) == len(b["losses"]) == 4
    for la, lb in zip(a["losses"], b["losses"]):
        assert abs(la - lb) <= 2e-4 * abs(lb), (a["losses"], b["losses"])
    assert a["losses"][-1] < a["losses"][0]                       # it trains
    for k in a["params"]:
        assert _rel(a["params"][k].numpy(), b["params"][k].numpy()) <= 2e-3, k


def test_ddp_reduces_buckets_while_the_backward_is_still_running(tmp_path):
    """train.py:49-51,78-81 under DDP: the native backward runs in three parts behind three chained autograd nodes
    (st_train_backward_part), so the gradients of final_proj / the upper blocks / the long-skip convs reach DDP's reducer while
    the lower blocks and the prenet are still to be enqueued.  From the second step on (DDP rebuilds its buckets in the order the
    gradients arrived in the first) the reducer must launch its first bucket BEFORE the last part of the backward is enqueued
    -- with one autograd node for the whole backward every bucket became ready at the same instant, after it.  2 ranks over gloo
    on the shared GPU; losses as in the single-process run."""
    out2, out1 = tmp_path / "ddp.pt", tmp_path / "one.pt"
    port = 29500 + (os.getpid() % 200)
    env = dict(os.environ, BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "train_ddp.py"), "--out", str(out2), "--backend", "gloo", "--trace"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_ddp.py"), "--out", str(out1)],
                        env=dict(os.environ), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    a, b = torch.load(out2), torch.load(out1)
    for la, lb in zip(a["losses"], b["losses"]):
        assert abs(la - lb) <= 2e-4 * abs(lb), (a["losses"], b["losses"])
    tr = a["trace"]
    steps = [i for i, ev in enumerate(tr) if ev[0] == "step"] + [len(tr)]
    for k in range(1, len(steps) - 1):                      # steps 1.. (buckets rebuilt after step 0)
        ev = tr[steps[k]:steps[k + 1]]
        parts = [i for i, x in enumerate(ev) if x[0] == "backward_part"]
        buckets = [i for i, x in enumerate(ev) if x[0] == "bucket"]
        assert [ev[i][1] for i in parts] == [0, 1, 2], ev
        assert buckets and buckets[0] < parts[2], ev        # the reducer is at work before the last part is enqueued
        before_last = sum(ev[i][2] for i in buckets if i < parts[2])
        print(f"step {k}: {len(buckets)} buckets, {sum(1 for i in buckets if i < parts[1])} launched before part 1, "
              f"{sum(1 for i in buckets if i < parts[2])} before part 2 ({before_last * 4 / 1e6:.1f} MB of {sum(ev[i][2] for i in buckets) * 4 / 1e6:.1f} MB)")


def test_ddp_over_rccl_one_rank(tmp_path):
    """The nccl (= RCCL on ROCm) backend around the shim's parameters: a 1-GPU box cannot host two RCCL ranks, but a
    world_size-1 process group still runs RCCL's communicator set-up, DDP's parameter broadcast / bucket construction over
    the 116 native parameters and the bucketed gradient all-reduce on the GPU (train.py:25-31,51).  Losses and parameters
    after 4 AdamW steps must equal the plain single-process run."""
    outd, out1 = tmp_path / "rccl.pt", tmp_path / "one.pt"
    port = 29300 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "train_ddp.py"), "--out", str(outd), "--backend", "nccl",
           "--force-dist"]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_ddp.py"), "--out", str(out1)],
                        env=dict(os.environ), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    a, b = torch.load(outd), torch.load(out1)
    assert a["world"] == 1 and len(a["losses"]) == 4
    for la, lb in zip(a["losses"], b["losses"]):
        assert abs(la - lb) <= 1e-5 * abs(lb), (a["losses"], b["losses"])
    for k in a["params"]:
        assert _rel(a["params"][k].numpy(), b["params"][k].numpy()) <= 1e-4, k
