#!/bin/bash
# Round 5, session 5: anatomy of the two weight-stationary kernels inside the solve (ablation builds, one launch sequence)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 900 python tools/class_times_libs.py default tools/ab/ows1.so tools/ab/ows2.so tools/ab/ows4.so tools/ab/ows6.so tools/ab/ows8.so tools/ab/ows16.so tools/ab/ows24.so tools/ab/ows32.so \
   tools/ab/qws2.so tools/ab/qws4.so tools/ab/qws6.so tools/ab/qws8.so tools/ab/qws16.so 2>&1 | grep -v Warning | tee $OUT/r05_ws_anatomy.txt
