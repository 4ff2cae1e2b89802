#!/bin/bash
set -u
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -m gpu -q --no-header -rf --timeout 900 -k "one_nfe_scalar or pad_leak or peaky or parameter_update or every_stage or tile_boundaries or strong_gates" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/parity_report.py > $O/parity.json 2>$O/parity.err; cat $O/parity.json
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
rocm-smi --showclocks --showpower 2>/dev/null | head -30 > $O/smi.txt
