#!/bin/bash
# Interleaved A/B of ST_SPLIT (and ST_HIP_GRAPH) settings on the headline solve. usage: bash tools/ab_env_split.sh rounds "split[:graph]" ...
R=$1; shift
for i in $(seq $R); do
  for v in "$@"; do
    sp=${v%%:*}; gr=0; [[ "$v" == *:* ]] && gr=${v##*:}
    ST_SPLIT=$sp ST_HIP_GRAPH=$gr python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ST_SPLIT=$sp graph=$gr', round(d['value']), round(d['ms_per_step'],3))"
  done
done
