#!/bin/bash
# Round 5, session 10: stream priority of the second solve part; fused-FFN tiles dealt round-robin to the XCDs
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
{
for pr in 1 2; do
  echo "== separate processes, alternating: default | ST_PART_PRIO=$pr (1 = second part on a LOW-priority stream, 2 = HIGH)"
  for i in 1 2 3; do
    echo -n "[default] "; timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
    echo -n "[prio $pr ] "; ST_PART_PRIO=$pr timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-42
  done
done
echo "== paired: default | fused-FFN tiles round-robin over the XCDs"
timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/ffnrr.so" 10 3 2>&1 | tail -4
AB_RAGGED=1 timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/ffnrr.so" 8 3 2>&1 | tail -3
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_prio_ffnrr.txt
