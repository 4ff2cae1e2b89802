#!/bin/bash
# A/B of two library builds inside one GPU session: interleaved bench runs (value, ms/step).
# usage: bash tools/ab_bench.sh <variant.so> [rounds]
V=$1; R=${2:-2}
for i in $(seq $R); do
  for lib in "" "$V"; do
    STABLETTS_HIP_LIB=$lib timeout 200 python bench.py --dev-env --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-default}', round(d['value']), round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
  done
done
