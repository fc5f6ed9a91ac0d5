#!/bin/bash
# Counters of the fp16 single-scale kernel (BASELINE config 5 geometry): HBM bytes and SQ wave states (development tool).
# Usage on the GPU box: bash tools/pmc_fp16.sh <tag> [env assignments...]   -> gpurun_out/pmc_fp16_<tag>/summary.txt
set -e
TAG=${1:-fp16}; shift || true
R=$(pwd)
OUT=$R/gpurun_out/pmc_fp16_$TAG
rm -rf $OUT && mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $R/tools/fp16_bench.py --reps 3 > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $R/tools/fp16_bench.py --reps 3 > $OUT/w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/a -- python3 $R/tools/fp16_bench.py --reps 3 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES \
  --output-format csv -d $OUT/b -- python3 $R/tools/fp16_bench.py --reps 3 > $OUT/b.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT "k_lk16" > $OUT/summary.txt
cat $OUT/summary.txt
