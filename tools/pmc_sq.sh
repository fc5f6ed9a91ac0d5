#!/bin/bash
# SQ issue/wait breakdown + effective clock of the LK kernels (development tool).
# Usage on the GPU box: bash tools/pmc_sq.sh <tag> [env assignments...]
set -e
TAG=${1:-sq}; shift || true
R=$(pwd)
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/a -- python3 $R/tools/kbench.py --pairs 32 --reps 3 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS \
  --output-format csv -d $OUT/b -- python3 $R/tools/kbench.py --pairs 32 --reps 3 > $OUT/b.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT "k_lkw<2" > $OUT/summary.txt
cat $OUT/summary.txt
