#!/bin/bash
set -u
O=gpurun_out/r02c; mkdir -p $O
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>$O/err_$label.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})" | tee -a $O/ab.log
}
for i in 1 2 3; do
  run new ST_NOP=0
  run r1 STABLETTS_HIP_LIB=tools/ab/r1.so
done
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf --timeout 900 > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 300 python tools/parity_report.py > $O/parity.json 2>$O/parity.err; cat $O/parity.json
