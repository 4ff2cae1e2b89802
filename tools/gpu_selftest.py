"""Stage-by-stage GPU check of the native estimator against the oracle (developer tool).

Run on the GPU box:  python tools/gpu_selftest.py [--quick]
Writes gpurun_out/selftest.json.  Uses the oracle only as the checker.
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                            # noqa: E402
from oracle.inputs import make_inputs                    # noqa: E402
from stabletts_amd.flow_matching import CFMDecoder       # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
report = {"stages": {}, "solves": {}, "timing": {}}


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def tm(x):  # (B,C,T) -> (B,T,C)
    return x.transpose(1, 2).contiguous().numpy()


def stage_check(dtype, B, T, lengths, seed, sd, t_val=0.37):
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    inp = make_inputs(B, T, seed=seed, lengths=lengths)
    taps = {}
    with torch.inference_mode():
        ref = oracle.decoder_forward(sd, torch.tensor(t_val), inp["z"], inp["mask"], inp["mu"], inp["c"], taps=taps)
    eng = dec.estimator.engine()
    eng.debug_capture(True)
    out = dec.estimator(torch.tensor(t_val), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    torch.cuda.synchronize()
    res = {}
    valid = inp["mask"][:, 0].bool().numpy()            # (B,T)
    Tp = (T + 63) // 64 * 64
    qs = math.log2(math.e) / 8.0

    def cmp(name, got, want, only_valid=False):
        got = got.reshape(want.shape)
        if only_valid:
            got = got[valid]; want = want[valid]
        res[name] = rel(got, want)

    cmp("cond", eng.debug_fetch("cond"), tm(taps["cond"]))
    cmp("h0", eng.debug_fetch("h0"), tm(taps["h0"]))
    for i in range(6):
        b = f"b{i}."
        if i >= 3:
            try:    # not captured when the long-skip conv carries the fused FiLM+LayerNorm epilogue
                cmp(f"lsc{i-3}", eng.debug_fetch(f"lsc{i-3}"), tm(taps[f"lsc{i-3}"]))
            except Exception:
                pass
        cmp(b + "x1", eng.debug_fetch(b + "x1"), tm(taps[b + "x1"]))
        cmp(b + "h1", eng.debug_fetch(b + "h1"), tm(taps[b + "h1"]))
        cmp(b + "q", eng.debug_fetch(b + "q"), (taps[b + "q"] * qs).numpy())
        cmp(b + "k", eng.debug_fetch(b + "k"), taps[b + "k"].numpy())
        tt = np.arange(Tp)
        pos = (tt & ~12) | ((tt & 4) << 1) | ((tt & 8) >> 1)      # key order inside each group of 16
        vt = eng.debug_fetch(b + "vt").reshape(B, 4, 64, Tp)[..., pos][..., :T]
        res[b + "vt"] = rel(vt, taps[b + "v"].numpy().transpose(0, 1, 3, 2))
        cmp(b + "attn", eng.debug_fetch(b + "attn"), tm(taps[b + "attn"]), only_valid=True)
        cmp(b + "x2", eng.debug_fetch(b + "x2"), tm(taps[b + "x2"]))
        cmp(b + "h2", eng.debug_fetch(b + "h2"), tm(taps[b + "h2"]))
        cmp(b + "u", eng.debug_fetch(b + "u"), tm(taps[b + "u"]))
        cmp(b + "x3", eng.debug_fetch(b + "x3"), tm(taps[b + "x3"]))
    res["out"] = rel(out.cpu().numpy(), ref.numpy())
    eng.debug_capture(False)
    pad = out.cpu()[~inp["mask"].bool().expand_as(out)]
    res["pad_absmax"] = float(pad.abs().max()) if pad.numel() else 0.0
    return res


def solve_check(dtype, B, T, lengths, n, solver, cfg, seed, sd, fs, fc):
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    inp = make_inputs(B, T, seed=seed, lengths=lengths)
    kw = None if cfg is None else dict(fake_speaker=fs, fake_content=fc, cfg_strength=cfg)
    ref = oracle.cfm_forward(sd, inp["mu"], inp["mask"], n, inp["z"], inp["c"], solver, kw)
    kwg = None if cfg is None else dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=cfg)
    out = dec(inp["mu"].cuda(), inp["mask"].cuda(), n, 1.0, inp["c"].cuda(), solver, kwg, z=inp["z"].cuda()).cpu()
    disp = (ref - inp["z"])
    return dict(rel_mel=rel(out.numpy(), ref.numpy()),
                rel_disp=float((out - ref).abs().max() / disp.abs().max()),
                finite=bool(torch.isfinite(out).all()))


def timing(dtype, B, T, n, sd, fs, fc):
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.cuda()
    inp = make_inputs(B, T, seed=0)
    g = {k: v.cuda() for k, v in inp.items() if k != "lengths"}
    kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
    for _ in range(2):
        dec(g["mu"], g["mask"], n, 1.0, g["c"], "euler", kw, z=g["z"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        dec(g["mu"], g["mask"], n, 1.0, g["c"], "euler", kw, z=g["z"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    eng = dec.estimator.engine()
    eng.profile_enable(True)
    dec(g["mu"], g["mask"], n, 1.0, g["c"], "euler", kw, z=g["z"])
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    for k, v in prof.items():
        if v["launches"]:
            v["avg_us"] = 1e3 * v["total_ms"] / v["launches"]
            if v["flops_per_launch"]:
                v["tflops"] = v["flops_per_launch"] / (v["avg_us"] * 1e-6) / 1e12
    return dict(ms_per_solve=dt * 1e3, frames_per_s=B * T / dt, profile=prof, device_mb=eng.device_bytes() / 1e6)


def main():
    quick = "--quick" in sys.argv
    print(torch.cuda.get_device_name(0), torch.version.hip)
    sd = oracle.make_state_dict(1234)
    fs, fc = oracle.make_cfg_params(4321)
    for dtype in ("bf16", "f16"):
        for tag, (B, T, lens, seed) in {"small": (2, 70, [70, 51], 11), "multi_tile": (2, 300, [300, 171], 12)}.items():
            try:
                r = stage_check(dtype, B, T, lens, seed, sd)
            except Exception as ex:  # keep going: we want the whole report from one GPU call
                r = {"error": repr(ex)}
            report["stages"][f"{dtype}/{tag}"] = r
            print(dtype, tag, json.dumps(r, indent=None))
        for name, args in {"euler_cfg": (2, 64, [64, 45], 4, "euler", 3.0, 21),
                           "midpoint": (1, 48, [48], 3, "midpoint", None, 23),
                           "rk4_cfg": (2, 33, [33, 30], 2, "rk4", 2.0, 24),
                           "euler10_cfg_T200": (2, 200, [200, 140], 10, "euler", 3.0, 25)}.items():
            try:
                r = solve_check(dtype, *args, sd, fs, fc)
            except Exception as ex:
                r = {"error": repr(ex)}
            report["solves"][f"{dtype}/{name}"] = r
            print(dtype, name, r)
        if not quick:
            for (B, T) in ((1, 500), (32, 1000)):
                try:
                    r = timing(dtype, B, T, 10, sd, fs, fc)
                except Exception as ex:
                    r = {"error": repr(ex)}
                report["timing"][f"{dtype}/B{B}_T{T}"] = r
                print(dtype, B, T, json.dumps({k: v for k, v in r.items() if k != "profile"}))
                if "profile" in r:
                    for k, v in r["profile"].items():
                        if v["launches"]:
                            print(f"   {k:12s} n={v['launches']:4d} total={v['total_ms']:9.3f} ms avg={v['avg_us']:9.1f} us "
                                  f"{v.get('tflops', 0):8.1f} TF/s")
    with open(os.path.join(OUT, "selftest.json"), "w") as fh:
        json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
