#!/bin/bash
# like abn.sh, for both plan kinds: single-scale and pyramidal lines of tools/kbench.py per library build
# Usage: bash tools/abn2.sh <rounds> <lib1.so> ... ; kbench args via KB_ARGS
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    printf "%-10s " "$(basename $L .so)"
    OFLK_LIB=$L timeout -k 10 180 python3 tools/kbench.py --reps 10 $KB_ARGS 2>&1 | grep -E "without|lk_iter_finest|lk_iter |lk_single" | sed 's/ without per-kernel events//' | tr -s ' ' | tr '\n' '|'
    echo
  done
done
