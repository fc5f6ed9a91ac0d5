#!/bin/bash
# Regenerate everything under profiles/ for one tag (run on the GPU box from the repo root):
#   bash tools/refresh_profiles.sh r01
# 1. bench.py line (default workload)            -> profiles/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same -> profiles/<tag>_rocprofv3_kernel_stats.csv
#                                                    profiles/<tag>_bench_under_rocprof.json
#                                                    profiles/<tag>_rocprofv3_dominant_kernel.json (finest-level launches only)
# 3. HBM traffic of the dominant kernel (PMC)     -> profiles/<tag>_hbm_traffic.json, <tag>_pmc/
# 4. SQ issue/wait counters of the LK kernels     -> profiles/<tag>_sq_counters.txt
set -e
TAG=${1:-r01}
R=$(pwd)
mkdir -p gpurun_out profiles/${TAG}_pmc
S=$R/gpurun_out/stats_$TAG
rm -rf $S && mkdir -p $S
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $S -- python3 $R/bench.py --no-cpu-baseline --no-one-pair > $S/bench.log 2>&1
cd $R
grep -h '^{' $S/bench.log | tail -1 > profiles/${TAG}_bench_under_rocprof.json
python3 tools/dominant_from_trace.py $S profiles/${TAG}_rocprofv3_dominant_kernel.json
cp "$(find $S -name '*kernel_stats.csv' | head -1)" profiles/${TAG}_rocprofv3_kernel_stats.csv
bash tools/measure_traffic.sh $TAG > gpurun_out/traffic_$TAG.log 2>&1
for k in fetch write; do
  cp "$(find gpurun_out/traffic_$TAG/$k -name '*counter_collection.csv' | head -1)" profiles/${TAG}_pmc/${k}_size_counter_collection.csv
done
# keep only the LK kernel rows of the raw counter files (the full files are several MB)
python3 - "$TAG" <<'PY'
import csv, sys
tag = sys.argv[1]
for k in ("fetch", "write"):
    p = f"profiles/{tag}_pmc/{k}_size_counter_collection.csv"
    rows = list(csv.DictReader(open(p)))
    keep = [r for r in rows if "k_lkw" in r["Kernel_Name"]]
    with open(p, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
PY
bash tools/pmc_sq.sh $TAG > gpurun_out/sq_$TAG.log 2>&1
cp gpurun_out/pmc_$TAG/summary.txt profiles/${TAG}_sq_counters.txt
# the bench line proper, last so that it sees the fresh traffic file
python3 bench.py > gpurun_out/bench_$TAG.log 2>&1
grep -h '^{' gpurun_out/bench_$TAG.log | tail -1 > profiles/${TAG}_bench.json
cat profiles/${TAG}_bench.json
