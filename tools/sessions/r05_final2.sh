#!/bin/bash
# Round-5 end state, second pass (after the s_setprio change): full GPU suite, smoke, bench lines, kernel stats + PMC, training profile
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v Warning > $OUT/r05_pytest_gpu_final.log; tail -3 $OUT/r05_pytest_gpu_final.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep smoke | tee $OUT/r05_smoke_final.log
bash tools/r05_final.sh
