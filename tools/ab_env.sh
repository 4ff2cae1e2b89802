#!/bin/bash
# A/B of an environment toggle inside one GPU session: interleaved bench runs.
# usage: bash tools/ab_env.sh "VAR=value" [rounds]
KV=$1; R=${2:-2}
for i in $(seq $R); do
  for kv in "ST_NOP=0" "$KV"; do
    env $kv timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv', round(d['value']), round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
  done
done
