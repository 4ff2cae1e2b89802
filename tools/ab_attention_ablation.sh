#!/bin/bash
# Ablation of the inference attention kernel's softmax arithmetic (attention.hip, kAblNoExp / kAblNoSum; the maxima left the common path after this ablation showed their cost):
# class time of the headline solve with the exps replaced by multiplies, without the row sums, and without both.  Results are numerically meaningless; only the attention class time is read.  Build first (no GPU):
#   for v in NOEXP NOSUM; do ST_BUILD_DEFS="-DST_DEVTOOLS -DST_ABL_$v" ST_BUILD_OUT=$PWD/tools/ab_$v.so python -m stabletts_amd.build; done
#   ST_BUILD_DEFS="-DST_DEVTOOLS -DST_ABL_NOEXP -DST_ABL_NOSUM" ST_BUILD_OUT=$PWD/tools/ab_NOALL.so python -m stabletts_amd.build
ST_SPLIT=1 bash tools/ab_classes.sh ${1:-2} $PWD/tools/ab_NOEXP.so $PWD/tools/ab_NOSUM.so $PWD/tools/ab_NOALL.so
