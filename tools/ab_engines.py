#!/usr/bin/env python
"""Interleaved A/B of two engine configurations inside ONE process (developer tool): each configuration = a set of ST_* variables
read at st_create (and visible at solve time: ST_SPLIT); the headline solve (B=32 x T=1000, 10 Euler steps, CFG) alternates between the two engines so that clock /
thermal drift hits both alike.  ONLY for configurations with the same number of solve parts: two engines' part streams share the
process's hardware queues, and a 4-part engine next to a 2-part one measured 30.7 ms instead of its 24.4 -- compare part counts in
separate processes, alternating (tools/r04b_session10.sh).  A configuration may also name its own BUILD of the library
(STABLETTS_HIP_LIB=/path/variant.so, e.g. ST_BUILD_DEFS=-DST_NT_DMA=1 ST_BUILD_OUT=... python -m stabletts_amd.build): both libraries are
loaded into the one process.     usage: python tools/ab_engines.py "" "ST_FUSED_FFN=3" [rounds] [per_round]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder

cfgs = [dict(kv.split("=") for kv in a.split()) if a.strip() else {} for a in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 30
per = int(sys.argv[4]) if len(sys.argv) > 4 else 3
RAGGED = os.environ.get("AB_RAGGED") == "1"
sd = oracle.make_state_dict(1234)
fs, fc = oracle.make_cfg_params(4321)
g = {k: v.cuda() for k, v in make_inputs(32, 1000, seed=0, ragged=RAGGED).items() if k != "lengths"}
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
decs = []
from stabletts_amd import _lib as _stlib
_default_lib = _stlib.LIB_PATH
for c in cfgs:
    for k, v in c.items(): os.environ[k] = v
    # a configuration may name its own build of the library (STABLETTS_HIP_LIB=path): an Engine keeps the CDLL it was created with
    _stlib.LIB_PATH, _stlib._lib = c.get("STABLETTS_HIP_LIB", _default_lib), None
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256).cuda()
    d.estimator.load_state_dict(sd); d.estimator.engine()
    for k in c: del os.environ[k]
    decs.append(d)
def run(d):
    c = cfgs[decs.index(d)]          # the configuration's variables are also visible at solve time (ST_SPLIT is read per call)
    for k, v in c.items(): os.environ[k] = v
    out = d(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"])
    for k in c: del os.environ[k]
    return out
outs = [run(d) for d in decs]
print("outputs bit-identical:", bool(torch.equal(outs[0], outs[1])), " max |diff|", float((outs[0] - outs[1]).abs().max()))
for d in decs:
    for _ in range(3): run(d)
ts = [[], []]
for r in range(rounds):
    for i in ((0, 1) if r % 2 == 0 else (1, 0)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(per): run(decs[i])
        torch.cuda.synchronize(); ts[i].append((time.perf_counter() - t0) / per * 1e3)
import statistics as st
for i in range(2):
    print(f"[{sys.argv[1 + i]}] {st.mean(ts[i]):.3f} ms  +- {st.stdev(ts[i]) / len(ts[i]) ** 0.5:.3f} (sem, {len(ts[i])} rounds x {per} solves)  median {st.median(ts[i]):.3f}")
d = [a - b for a, b in zip(ts[0], ts[1])]
print(f"paired difference A - B: {st.mean(d):+.3f} ms +- {st.stdev(d) / len(d) ** 0.5:.3f}  ({st.mean(d) / st.mean(ts[0]) * 100:+.2f} %)")
