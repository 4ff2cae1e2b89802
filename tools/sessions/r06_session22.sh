#!/bin/bash
# Round 6, session 22: upper bound of a two-part training step: two independent half-batch chains on two streams vs one full-batch chain.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
{ for i in 1 2 3; do timeout 300 python tools/train_two_chains.py 2>&1 | tail -1; done; ST_TRAIN_SIDE=0 timeout 300 python tools/train_two_chains.py 2>&1 | tail -1; } | tee $OUT/r06_s22_two_chains.txt
