#!/bin/bash
# interleaved A/B of bench.py under different environments (developer tool): tools/ab_env.sh "A=1" "B=2 C=3" ...
for rep in 1 2 3; do
  for cfg in "$@"; do
    echo -n "rep $rep [$cfg] "; env $cfg timeout 120 python bench.py --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']))"
  done
done
