#!/bin/bash
# round-2 GPU session 1: parity suite + bench lines + two-part-solve A/B
set -u
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf --timeout 900 -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
run() {  # label, env..., extra args via BARGS
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras $BARGS 2>$O/err_$label.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['roofline']['kernel'], {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})" | tee -a $O/ab.log
}
BARGS=""
for i in 1 2; do
  run split1 ST_SPLIT=1
  run split2 ST_SPLIT=2
  run split1_graph ST_SPLIT=1 ST_HIP_GRAPH=1
  run split2_graph ST_SPLIT=2 ST_HIP_GRAPH=1
done
BARGS="--dtype f16"; run f16_split1 ST_SPLIT=1; run f16_split2 ST_SPLIT=2
BARGS="--ragged"; run ragged_split1 ST_SPLIT=1; run ragged_split2 ST_SPLIT=2
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
