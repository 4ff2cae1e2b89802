"""Native training path (st_train_forward / st_train_backward behind stabletts_amd.autograd) on a real MI355X:
gradients of CFMDecoder.compute_loss against the REAL reference's gradients (tests/golden/loss_grads.npz, dropout
off), the attention backward kernels at matched inputs, counter-based dropout against the oracle run with the same
masks, and a 2-process DDP run.  Run with ``-m gpu``.

Gates (max |native - ref| / max |ref| per tensor): f16 operands 3e-3, bf16 operands 2e-2 -- except the q / k
projections of the attention: d loss / d q and d loss / d k subtract two nearly equal terms (dP - D) and, at these
random-init weights where V and K are almost uncorrelated over the keys, a 5e-4 relative difference in V (the size of
the 16-bit FORWARD's own deviation from the fp32 reference) moves them by several percent (oracle experiment recorded
in DESIGN.md).  The backward kernels themselves are checked to operand precision at matched inputs below.
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
TOL_QK = {"f16": 6e-2, "bf16": 4e-1}       # measured 2.3e-2 / 1.9e-1 (conditioning of dq, dk at random init, see above)


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _is_qk(name):
    return ".attn.conv_q." in name or ".attn.conv_k." in name


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
        scale = float(eng.debug_fetch("g.scale")[0])
        m = inp["mask"][:, 0].double().numpy()
        bias = (1 - m)[:, None, None, :] * (-1e30)
        for i in (5, 2):
            q = eng.debug_fetch(f"t{i}.q").reshape(B, 4, T, 64).astype(np.float64)        # q_s = q_r * log2(e) / 8
            k = eng.debug_fetch(f"t{i}.k").reshape(B, 4, T, 64).astype(np.float64)
            v = eng.debug_fetch(f"t{i}.vt").reshape(B, 4, 64, Tp)[..., pos][..., :T].transpose(0, 1, 3, 2).astype(np.float64)
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


# ---- python port of the counter-based dropout hash (stabletts_amd/csrc/common.h: drop_hash, train_kernels.hip: make_drop)
_M = np.uint64(0xFFFFFFFF)


def _mix32(h):
    h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x85ebca6b)) & _M
    h = h ^ (h >> np.uint64(13)); h = (h * np.uint64(0xc2b2ae35)) & _M
    return h ^ (h >> np.uint64(16))


def _drop_factor(seed, salt, p, a, b):
    s64 = (seed * 0x100000001B3 + (salt + 1) * 0xD6E8FEB86659FD93) % (1 << 64)
    lo, hi = np.uint64(s64 & 0xFFFFFFFF), np.uint64(s64 >> 32)
    a = a.astype(np.uint64); b = b.astype(np.uint64)
    h = _mix32(lo ^ ((a * np.uint64(0x9E3779B1)) & _M))
    h = _mix32(h ^ hi ^ ((b * np.uint64(0x85ebca77)) & _M))
    thresh = min(int(p * 4294967296.0), 4294967295)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(h >= np.uint64(thresh), scale, np.float32(0.0)).astype(np.float32)


@pytest.mark.parametrize("dt", ["f16"])
def test_dropout_forward_and_backward_match_oracle_with_same_masks(sd, dt):
    """Train mode (p_dropout = 0.1 on the FFN activations and the attention probabilities,
    diffusion_transformer.py:28,77): the native counter-based dropout is reproduced in numpy, the oracle is run under
    autograd with exactly those keep / (1 - p) factors, and loss + gradients must agree as in eval mode.  Also:
    same torch seed -> bitwise identical loss; different seed -> different masks."""
    B, T, lengths, p = 2, 70, [70, 45], 0.1
    dec = _decoder(sd, dt, train=True)
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
    torch.manual_seed(123)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    F_, H = 1024, 4
    drop = {"ffn": [], "attn": []}
    n_, t_, c_ = np.meshgrid(np.arange(B), np.arange(T), np.arange(F_), indexing="ij")
    idx = ((n_ * T + t_) * F_ + c_).astype(np.uint64)
    nn, hh, qq, kk = np.meshgrid(np.arange(B), np.arange(H), np.arange(T), np.arange(T), indexing="ij")
    for i in range(6):
        f = _drop_factor(seed, 2 * i, p, idx & _M, idx >> np.uint64(32))                      # [B][T][F] time-major
        drop["ffn"].append(torch.from_numpy(f).permute(0, 2, 1).contiguous())
        a = _drop_factor(seed, 2 * i + 1, p, ((nn * H + hh) * T + qq).astype(np.uint64), kk.astype(np.uint64))
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
