#!/bin/bash
# Round 5, session 4: attention XCD interleave (new default) vs contiguous; block counts of the weight-stationary kernels
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py -x -q -k "weight_stationary or ragged_tile or two_part or c2_ or interior or length_one" 2>&1 | tail -3 | tee $OUT/r05_s4_tests.log
{
for r in 0 1; do
  echo "== ragged=$r ST_SPLIT=1 class times: contiguous | interleaved"
  echo -n "[contig] "; env CLASS_TIMES_RAGGED=$r ST_SPLIT=1 ST_ATTN_XCD_CONTIGUOUS=1 timeout 200 python tools/class_times.py 2>&1 | tail -1
  echo -n "[inter ] "; env CLASS_TIMES_RAGGED=$r ST_SPLIT=1 timeout 200 python tools/class_times.py 2>&1 | tail -1
done
echo "== paired all-ones: attention contiguous | interleaved"
timeout 400 python tools/ab_engines.py "ST_ATTN_XCD_CONTIGUOUS=1" "" 12 3 2>&1 | tail -4
echo "== paired ragged: attention contiguous | interleaved"
AB_RAGGED=1 timeout 400 python tools/ab_engines.py "ST_ATTN_XCD_CONTIGUOUS=1" "" 12 3 2>&1 | tail -4
for ob in 192 256; do
  echo "== paired all-ones: oproj blocks default(224) | $ob"
  timeout 400 python tools/ab_engines.py "" "ST_OPROJ_WS_BLOCKS=$ob" 10 3 2>&1 | tail -3
done
echo "== paired ragged: oproj blocks default(224) | 256"
AB_RAGGED=1 timeout 400 python tools/ab_engines.py "" "ST_OPROJ_WS_BLOCKS=256" 10 3 2>&1 | tail -3
echo "== paired all-ones: qkv blocks default(240) | 192"
timeout 400 python tools/ab_engines.py "" "ST_QKV_WS_BLOCKS=192" 10 3 2>&1 | tail -3
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_attn_xcd_ws_blocks.txt
