#!/bin/bash
# Alternating A/B/.../N of several library builds on ONE box (box-to-box spread exceeds most single changes).
# Usage: bash tools/abn.sh <rounds> <lib1.so> <lib2.so> ... ; kbench args via KB_ARGS
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    printf "%-28s " "$(basename $L .so)"
    OFLK_LIB=$L timeout -k 10 180 python3 tools/kbench.py --pairs 32 --reps 10 $KB_ARGS 2>&1 | grep -E "pyramidal without|lk_iter_finest|lk_iter |pyr_down|flow_upsample" | sed 's/== pyramidal without per-kernel events://' | tr -s ' ' | tr '\n' '|'
    echo
  done
done
