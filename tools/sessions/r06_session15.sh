#!/bin/bash
# Round 6, session 15: the training step of round 5's tree (commit 60cb8fc, its own Python + library under tools/ab/r05_tree) against
# the current tree, alternating in one call: the cumulative effect of the round's training work on one box.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tb() { (cd $2 && timeout 300 python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', {k: round(v,3) for k,v in j.items() if k.startswith('ms_') or k=='frac_of_mfma_peak'})"); }
for i in 1 2 3 4; do tb round5_tree $ROOT/tools/ab/r05_tree; tb round6_tree $ROOT; done 2>&1 | tee $OUT/r06_s15_train_r05_vs_r06.txt
