#!/bin/bash
# Interleaved A/B of library variants inside one GPU session: per-class kernel times of the headline solve.
# usage: bash tools/ab_classes.sh rounds lib1 lib2 ...   ("" = the in-tree default library)
R=$1; shift
for i in $(seq $R); do
  for lib in "" "$@"; do
    STABLETTS_HIP_LIB=$lib timeout 300 python tools/class_times.py 2>&1 | tail -1
  done
done
