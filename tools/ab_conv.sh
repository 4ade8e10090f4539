#!/bin/bash
# A/B of library builds / environment switches on ONE box, alternating rounds (bench_conv.py per variant).
#   tools/ab_conv.sh BATCH ROUNDS "name1:lib1[:ENV=VAL,...]" "name2:lib2[:ENV=VAL,...]" ...
BATCH=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for V in "$@"; do
    IFS=: read -r NAME LIB ENVS <<< "$V"
    echo "== $NAME B=$BATCH"
    env SUBREG_LIB=$PWD/subspace-reg_amd/subreg_hip/$LIB $(echo $ENVS | tr ',' ' ') python tools/bench_conv.py --batch $BATCH ${ONLY:+--only "$ONLY"} 2>/dev/null | grep -v amdgpu.ids
  done
done
