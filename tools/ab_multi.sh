#!/bin/bash
# Interleaved A/B/C... of library variants inside one GPU session.  usage: bash tools/ab_multi.sh rounds lib1 lib2 ...
R=$1; shift
for i in $(seq $R); do
  for lib in "" "$@"; do
    STABLETTS_HIP_LIB=$lib timeout 200 python bench.py --dev-env --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-default}', round(d['value']), round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
  done
done
