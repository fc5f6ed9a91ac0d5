#!/bin/bash
# Timing ablations of the fused LK kernel (development tool; the variants compute WRONG results).
# Builds liboflk with one stage's arithmetic removed at a time (OFLK_ABLATE bits, oflk_kernels.hpp)
# and times the finest-level iteration launch and the single-scale launch at 32 x 1080p.
#   bit 0 (1)  no 2x2 solve        bit 1 (2)  no window adds (LDS reads kept)
#   bit 2 (4)  no window LDS reads  bit 3 (8)  no fp64 warp arithmetic (flows become garbage: ITER figure invalid)
#   bit 4 (16) no Sobel arithmetic  31 = all of them (what is left is the data-movement skeleton)
#   bit 5 (32) flow_out = flow_in (identity warp: keeps the gather pattern sane for the ITER figures)
#   bit 6 (64) no fp64 tap sums     bit 7 (128) no fp64 tap coordinates / weights
# Usage on the GPU box: bash tools/ablate.sh > gpurun_out/ablate.txt   (build the variants first, on any host:
#   bash tools/ablate.sh build)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
VARIANTS=${VARIANTS:-"0 1 2 4 16 31"}
if [ "$1" = build ]; then
  for a in $VARIANTS; do
    make -C $R/optical-flow-fpga_amd/csrc -B OUT=$R/tools/liboflk_abl$a.so DEFS=-DOFLK_ABLATE=$a > /dev/null &
  done
  wait
  exit 0
fi
for r in 1 2; do
  for a in $VARIANTS; do
    echo -n "ablate=$a "
    OFLK_LIB=$R/tools/liboflk_abl$a.so timeout -k 10 120 python3 $R/tools/kbench.py --pairs 32 --reps 6 2>&1 | grep -E "lk_iter_finest|lk_single" | tr -s " " | tr "\n" "|"
    echo
  done
done
