#!/bin/bash
# What do SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES mean in cycles?  The same counters over tools/ubench/valu_wall (kernels whose
# vector-ALU occupancy is known from wall-clock) and over the LK kernels.  Usage on the GPU box: bash tools/pmc_calib.sh <tag>
set -e
TAG=${1:-calib}
R=$(pwd)
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/ubench -- $R/tools/ubench/valu_wall > $OUT/ubench.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/lk -- python3 $R/tools/kbench.py --pairs 32 --reps 3 > $OUT/lk.log 2>&1
cd $R
python3 tools/pmc_calib.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
