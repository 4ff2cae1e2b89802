#!/bin/bash
# Round 5, session 13: the part phase offset (now built in) -- bitwise tests, ragged A/B against the library without it
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py -x -q -k "two_part or graph or c2_ or default_engine" 2>&1 | tail -3
{
echo "== paired all-ones: previous commit's library | with the 100-us part offset"
timeout 300 python tools/ab_engines.py "STABLETTS_HIP_LIB=$ROOT/tools/ab/base.so" "" 10 3 2>&1 | tail -4
echo "== paired ragged"
AB_RAGGED=1 timeout 300 python tools/ab_engines.py "STABLETTS_HIP_LIB=$ROOT/tools/ab/base.so" "" 10 3 2>&1 | tail -3
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_part_phase_builtin.txt
