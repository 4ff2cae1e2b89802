#!/bin/bash
# Round 6, session 5: the one structural attempt on the two-part solve's queueing: qkv_ws blocks pulling tiles from
# a shared per-launch queue (ST_QKV_WS_QUEUE=1) -- correctness under the whole engine / parity suites, then paired A/B in one process
# (all-ones, ragged, single sequence).
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
ST_QKV_WS_QUEUE=1 timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/r06_s5_pytest_queue.log
{
echo "== all-ones, default two-part solve"; timeout 600 python tools/ab_engines.py "" "ST_QKV_WS_QUEUE=1" 30 3 2>&1 | grep -v amdgpu.ids
echo "== ragged (32 utterances U{600..1000}), default two-part solve"; AB_RAGGED=1 timeout 600 python tools/ab_engines.py "" "ST_QKV_WS_QUEUE=1" 30 3 2>&1 | grep -v amdgpu.ids
echo "== all-ones, single launch sequence (ST_SPLIT=1)"; timeout 600 python tools/ab_engines.py "ST_SPLIT=1" "ST_SPLIT=1 ST_QKV_WS_QUEUE=1" 20 3 2>&1 | grep -v amdgpu.ids
} | tee $OUT/r06_s5_ab_qkv_queue.txt
