#!/bin/bash
# A/B two builds of liboflk on ONE box, alternating (box-to-box spread is ~5 %, larger than most
# single changes).  Usage: bash tools/ab.sh <old.so> <new.so> [rounds] [kbench args...]
OLD=$1; NEW=$2; R=${3:-3}; shift 3 || true
for r in $(seq $R); do
  for v in old new; do
    L=$NEW; [ $v = old ] && L=$OLD
    echo -n "$v "
    OFLK_LIB=$L timeout -k 10 120 python3 tools/kbench.py --pairs 32 --reps 10 "$@" 2>&1 | grep -E "pyramidal without|lk_iter_finest|pyr_down|flow_upsample" | sed 's/== pyramidal without per-kernel events://' | tr -s ' ' | tr '\n' '|'
    echo
  done
done
