#!/usr/bin/env python
"""Per-class kernel times of the headline solve (one launch sequence, ST_SPLIT=1) for several builds of the library in ONE process
(developer tool: ablation builds, ST_BUILD_DEFS="-DST_DEVTOOLS -DST_OWS_VAR=.." ST_BUILD_OUT=tools/ab/x.so python -m stabletts_amd.build).
    python tools/class_times_libs.py default tools/ab/ows1.so tools/ab/ows2.so ..."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.inputs import make_inputs
from stabletts_amd import _lib as _stlib
from stabletts_amd.flow_matching import CFMDecoder

os.environ["ST_SPLIT"] = "1"
RAGGED = os.environ.get("CLASS_TIMES_RAGGED") == "1"
sd = oracle.make_state_dict(1234)
fs, fc = oracle.make_cfg_params(4321)
g = {k: v.cuda() for k, v in make_inputs(32, 1000, seed=0, ragged=RAGGED).items() if k != "lengths"}
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
default = _stlib.LIB_PATH
decs = []
for name in sys.argv[1:]:
    _stlib.LIB_PATH, _stlib._lib = (default if name == "default" else os.path.abspath(name)), None
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256).cuda()
    d.estimator.load_state_dict(sd); d.estimator.engine()
    decs.append((name, d))
run = lambda d: d(g["mu"], g["mask"], 10, 1.0, g["c"], "euler", kw, z=g["z"])
for rep in range(2):
    for name, d in decs:
        for _ in range(2): run(d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): run(d)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        eng = d.estimator.engine(); eng.profile_enable(True); run(d); torch.cuda.synchronize()
        pr = eng.profile_read(); eng.profile_enable(False)
        print(f"{os.path.basename(name):12s} solve {ms:6.2f} ms  " + "  ".join(f"{k} {pr[k]['total_ms']:.2f}" for k in ("qkv_rope", "attention", "out_proj", "ffn_conv2", "lsc_conv")), flush=True)
