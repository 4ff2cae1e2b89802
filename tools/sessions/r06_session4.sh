#!/bin/bash
# Round 6, session 4: split-precision attention operands + the run-time attention statistic: new parity tests, the tests that guard the
# q/k/v and attention kernels, and the cost of the split mode at the headline shape.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "statistic or split_precision or auto_switches or trained_like or re_reference" 2>&1 | grep -v "^$" | tail -40 > $OUT/r06_s4_pytest_split.log; cat $OUT/r06_s4_pytest_split.log
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -5
python - <<'PY' 2>&1 | tee $OUT/r06_s4_split_cost.txt
import time, torch, oracle
from oracle.inputs import make_inputs
from stabletts_amd.flow_matching import CFMDecoder
sd = oracle.make_state_dict(1234); fs, fc = oracle.make_cfg_params(4321)
inp = {k: v.cuda() for k, v in make_inputs(32, 1000, seed=0).items() if k != "lengths"}
kw = dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=3.0)
decs = {}
for mode in ("16bit", "split"):
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, attention_precision=mode); d.estimator.load_state_dict(sd); decs[mode] = d.cuda()
def run(d, n=5):
    for _ in range(2): d(inp["mu"], inp["mask"], 10, 1.0, inp["c"], "euler", kw, z=inp["z"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): d(inp["mu"], inp["mask"], 10, 1.0, inp["c"], "euler", kw, z=inp["z"])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for r in range(3):
    print("round", r, {m: round(run(d), 3) for m, d in decs.items()}, "ms per solve (B=32 x T=1000, n=10, CFG)")
for m, d in decs.items():
    e = d.estimator.engine(); e.profile_enable(True, ["attention", "qkv_rope"])
    import os; os.environ["ST_SPLIT"] = "1"
    d(inp["mu"], inp["mask"], 10, 1.0, inp["c"], "euler", kw, z=inp["z"]); p = e.profile_read(); e.profile_enable(False); del os.environ["ST_SPLIT"]
    print(m, {k: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 1) for k, v in p.items() if v["launches"]}, "us per launch (single sequence, event-timed)")
PY
