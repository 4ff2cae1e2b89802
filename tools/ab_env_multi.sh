#!/bin/bash
# Interleaved A/B of environment settings inside one GPU session: bash tools/ab_env_multi.sh rounds "A=1" "B=2 C=3" ...
R=$1; shift
for i in $(seq $R); do
  for kv in "ST_NOP=0" "$@"; do
    env $kv timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
  done
done
