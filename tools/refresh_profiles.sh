#!/bin/bash
# Regenerate everything under profiles/ for one tag (run on the GPU box from the repo root):
#   bash tools/refresh_profiles.sh r02
# 1. bench.py line (default workload)            -> profiles/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same -> profiles/<tag>_rocprofv3_kernel_stats.csv
#                                                    profiles/<tag>_bench_under_rocprof.json
#                                                    profiles/<tag>_rocprofv3_dominant_kernel.json (finest-level launches only)
# 3. HBM traffic of the dominant kernel (PMC)     -> profiles/<tag>_hbm_traffic.json, <tag>_pmc/
# 4. SQ issue/wait counters of the LK kernels     -> profiles/<tag>_sq_counters.txt
set -e
TAG=${1:-r02}
R=$(pwd)
mkdir -p gpurun_out profiles/${TAG}_pmc
S=$R/gpurun_out/stats_$TAG
rm -rf $S && mkdir -p $S
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $S -- python3 $R/bench.py --no-cpu-baseline --no-one-pair --no-live-traffic > $S/bench.log 2>&1
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
# 5. instruction-side bounds of the dominant kernel -> profiles/<tag>_issue_bounds.json.  The ISA it prices is regenerated here
#    from the sources of the library being measured (a stale oflk_gfx950.s once outlived the .so it described)
make -C optical-flow-fpga_amd/csrc asm > gpurun_out/asm_$TAG.log 2>&1 || true
# (vector-ALU cycles per instruction checked against wall-clock, with a census of where the waves ran)
[ -x tools/ubench/valu_wall ] && timeout -k 10 300 ./tools/ubench/valu_wall > profiles/${TAG}_valu_wall.txt 2>&1 || true
python3 tools/issue_bounds.py gpurun_out/pmc_$TAG profiles/${TAG}_valu_wall.txt profiles/${TAG}_issue_bounds.json > gpurun_out/issue_$TAG.log 2>&1 || cat gpurun_out/issue_$TAG.log
# 5b. what the SQ "VALU busy" counters mean in cycles (same counters over kernels of known occupancy), and which pipe a
#     partner kernel on a second stream takes from the step
bash tools/pmc_calib.sh ${TAG}_calib > gpurun_out/calib_$TAG.log 2>&1 && cp gpurun_out/pmc_${TAG}_calib/summary.txt profiles/${TAG}_sq_counter_calibration.txt || true
[ -f tools/ubench/libspin.so ] && timeout -k 10 400 python3 tools/corun.py --out profiles/${TAG}_corun.json > gpurun_out/corun_$TAG.log 2>&1 || true
# 6. in-kernel timeline of the same launch (diagnostic build)                  -> profiles/<tag>_stamps_timeline.{json,txt}
if [ -f tools/liboflk_stamps.so ]; then
  OFLK_LIB=tools/liboflk_stamps.so timeout -k 10 200 python3 tools/stamps.py profiles/${TAG}_stamps_timeline.json > profiles/${TAG}_stamps_timeline.txt 2>&1 || true
  rm -f profiles/${TAG}_stamps_timeline.raw.npy
fi
# 6b. block lifetimes of the same launch, fp16 mode counters / memory-pattern ceilings / EPE report
if [ -f tools/liboflk_bt.so ]; then
  OFLK_LIB=tools/liboflk_bt.so timeout -k 10 200 python3 tools/block_times.py profiles/${TAG}_block_times.json > gpurun_out/bt_$TAG.log 2>&1 || true
  rm -f profiles/${TAG}_block_times.raw.npy
fi
bash tools/pmc_fp16.sh $TAG > gpurun_out/pmc_fp16_$TAG.log 2>&1 && cp gpurun_out/pmc_fp16_$TAG/summary.txt profiles/${TAG}_pmc_fp16.txt || true
python3 tools/hbm_probe.py > profiles/${TAG}_hbm_probe.txt 2>&1 || true
[ -x tools/ubench/rowwalk ] && timeout -k 10 120 tools/ubench/rowwalk 76 > profiles/${TAG}_rowwalk.txt 2>&1 || true
python3 -m pytest tests/test_gpu_fp16.py -q > gpurun_out/fp16_tests_$TAG.log 2>&1 && cp gpurun_out/fp16_epe.json profiles/${TAG}_fp16_epe.json || true
python3 -m pytest tests/test_gpu_round3.py -q -k contracted > gpurun_out/contracted_tests_$TAG.log 2>&1 && cp gpurun_out/contracted_epe.json profiles/${TAG}_contracted_epe.json || true
python3 tools/host_latency.py > profiles/${TAG}_host_latency.txt 2>&1 && python3 tools/host_latency.py batch >> profiles/${TAG}_host_latency.txt 2>&1 || true
# 7. every BASELINE config that fits one GPU, and BASELINE configs[3] as one job on this GPU
python3 tools/measure_configs.py profiles/${TAG}_configs.json > gpurun_out/cfg_$TAG.log 2>&1 || tail -3 gpurun_out/cfg_$TAG.log
python3 bench.py --config 4k64 --steps 5 --warmup 1 --no-one-pair > gpurun_out/bench_4k64_$TAG.log 2>&1 || tail -3 gpurun_out/bench_4k64_$TAG.log
grep -h '^{' gpurun_out/bench_4k64_$TAG.log | tail -1 > profiles/${TAG}_bench_4k64_1gpu.json
# the bench line proper, last so that it sees the fresh traffic file
python3 bench.py > gpurun_out/bench_$TAG.log 2>&1
grep -h '^{' gpurun_out/bench_$TAG.log | tail -1 > profiles/${TAG}_bench.json
cat profiles/${TAG}_bench.json
python3 tools/profiles_summary.py $TAG || true
# gpurun merges only gpurun_out/ back to the build container: leave a copy of everything there
rm -rf gpurun_out/profiles_$TAG && mkdir -p gpurun_out/profiles_$TAG && cp -r profiles/${TAG}_* gpurun_out/profiles_$TAG/
