#!/bin/bash
# Round 5, session 7: non-temporal LDS-DMA for streamed-once activation tiles (new default) vs without; nt residual-row loads; sorted ragged batch
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -k "weight_stationary or fused_ffn_is_bit or phased_k" 2>&1 | tail -2
{
echo "== one launch sequence (ST_SPLIT=1), class times: default = nt LDS-DMA on streamed-once tiles (not q/k/v), nont = none, ntres = + nt residual-row loads"
timeout 900 python tools/class_times_libs.py default tools/ab/nont.so tools/ab/ntres.so 2>&1
for v in nont ntres; do
  echo "== paired default parts: default | $v"
  timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/$v.so" 10 3 2>&1 | tail -4
done
echo "== paired ragged: default | nont"
AB_RAGGED=1 timeout 300 python tools/ab_engines.py "" "STABLETTS_HIP_LIB=$ROOT/tools/ab/nont.so" 8 3 2>&1 | tail -3
echo "== bench --ragged (length-sorted batch): attention groups contiguous per XCD | dealt round-robin (default), twice"
for i in 1 2; do
  ST_ATTN_XCD_CONTIGUOUS=1 timeout 300 python bench.py --dev-env --ragged --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('contiguous', round(d['ms_per_step'],3), round(d['value']), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
  timeout 300 python bench.py --ragged --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round-robin', round(d['ms_per_step'],3), round(d['value']), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all-ones', round(d['ms_per_step'],3), round(d['value']), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
} 2>&1 | grep -v Warning | tee $OUT/r05_ab_nt_dma.txt
