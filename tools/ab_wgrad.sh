#!/bin/bash
# Ablation of the TN weight-gradient kernel's K loop (wgrad_tn.hip): full loop vs compute only (-DST_TN_NO_DMA) vs DMA only
# (-DST_TN_NO_MMA).  Build the variants first (no GPU needed):
#   ST_BUILD_DEFS=-DST_TN_NO_DMA ST_BUILD_OUT=$PWD/tools/ab_tn_nodma.so python -m stabletts_amd.build
#   ST_BUILD_DEFS=-DST_TN_NO_MMA ST_BUILD_OUT=$PWD/tools/ab_tn_nomma.so python -m stabletts_amd.build
for v in "" $PWD/tools/ab_tn_nodma.so $PWD/tools/ab_tn_nomma.so; do
  echo "== ${v:-default library}"
  STABLETTS_HIP_LIB=$v bash tools/profile_train.sh ab 2>&1 | grep -E "wgrad_tn|wgrad_reduce"
done
