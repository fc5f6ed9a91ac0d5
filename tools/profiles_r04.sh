#!/bin/bash
# Round-4 profiles (run on the GPU box from the repo root; ~10 minutes):  bash tools/profiles_r04.sh
#   r04_bench.json                         the contract line (exact value + tolerance_mode + single-scale / gather extras)
#   r04_rocprofv3_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the bench command
#   r04_rocprofv3_dominant_kernel.json     finest-level launches of the exact iteration kernel k_lkw<2, ITER>
#   r04_rocprofv3_tolerant_kernel.json     ... of the streaming kernel k_lks<ITER> (tolerance_mode's dominant kernel)
#   r04_hbm_traffic.json, r04_hbm_traffic_tolerant.json, r04_pmc/   FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
#   r04_sq_counters.txt, r04_issue_bounds.json                      SQ counters; instruction-side bounds incl. the single-scale kernels
#   r04_configs.json, r04_bench_4k64_1gpu.json, r04_tolerant_epe.json, r04_host_latency.txt
set -e
TAG=r04
R=$(pwd)
mkdir -p gpurun_out profiles/${TAG}_pmc
make -C optical-flow-fpga_amd/csrc asm > gpurun_out/asm_$TAG.log 2>&1 || true     # the ISA of the shipped library, for issue_bounds
S=$R/gpurun_out/stats_$TAG
rm -rf $S && mkdir -p $S
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $S -- python3 $R/bench.py --no-cpu-baseline --no-one-pair --no-live-traffic > $S/bench.log 2>&1
cd $R
grep -h '^{' $S/bench.log | tail -1 > profiles/${TAG}_bench_under_rocprof.json
python3 tools/dominant_from_trace.py $S profiles/${TAG}_rocprofv3_dominant_kernel.json
python3 tools/dominant_from_trace.py $S profiles/${TAG}_rocprofv3_tolerant_kernel.json "k_lks<"
cp "$(find $S -name '*kernel_stats.csv' | head -1)" profiles/${TAG}_rocprofv3_kernel_stats.csv
echo "== kernel stats done"
bash tools/measure_traffic.sh $TAG > gpurun_out/traffic_$TAG.log 2>&1 || tail -5 gpurun_out/traffic_$TAG.log
for k in fetch write; do
  cp "$(find gpurun_out/traffic_$TAG/$k -name '*counter_collection.csv' | head -1)" profiles/${TAG}_pmc/${k}_size_counter_collection.csv
done
python3 - "$TAG" <<'PY'
import csv, json, sys
tag = sys.argv[1]
def per_launch(k, counter, match):
    rows = [(int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f"profiles/{tag}_pmc/{k}_size_counter_collection.csv"))
            if match in r["Kernel_Name"] and r["Counter_Name"] == counter]
    big = max(g for g, _ in rows)
    v = [x for g, x in rows if g == big]
    return sum(v) / len(v), len(v), big
try:
    f, nf, grid = per_launch("fetch", "FETCH_SIZE", "k_lks<1")
    w, nw, _ = per_launch("write", "WRITE_SIZE", "k_lks<1")
    pairs = json.load(open(f"profiles/{tag}_hbm_traffic.json"))["pairs"]
    res = {"kernel": "k_lks<MODE_ITER> finest level (tolerance_mode)", "pairs": pairs, "shape": [1080, 1920], "grid_size": grid, "launches_averaged": [nf, nw],
           "fetch_bytes_corrected_x2": f * 1024 * 2, "write_bytes": w * 1024, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
           "algorithmic_bytes_per_launch": 24 * 1080 * 1920 * pairs,
           "method": "as profiles/%s_hbm_traffic.json: separate --pmc passes over the bench command (its tolerance_mode leg), FETCH_SIZE doubled" % tag}
    json.dump(res, open(f"profiles/{tag}_hbm_traffic_tolerant.json", "w"), indent=1)
    print(json.dumps(res))
except Exception as e:
    print("tolerant traffic:", e)
for k in ("fetch", "write"):     # keep only the LK kernel rows of the raw counter files
    p = f"profiles/{tag}_pmc/{k}_size_counter_collection.csv"
    rows = list(csv.DictReader(open(p)))
    keep = [r for r in rows if "k_lkw" in r["Kernel_Name"] or "k_lks" in r["Kernel_Name"]]
    with open(p, "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=rows[0].keys())
        wr.writeheader()
        wr.writerows(keep)
PY
echo "== traffic done"
bash tools/pmc_sq.sh $TAG > gpurun_out/sq_$TAG.log 2>&1 || tail -5 gpurun_out/sq_$TAG.log
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG "k_lk" > profiles/${TAG}_sq_counters.txt
# the single-scale kernels on their own: tile kernel 5x5 and 7x7, streaming kernel (32 pairs of 1080p)
OUT=$R/gpurun_out/pmc_${TAG}_single
rm -rf $OUT && mkdir -p $OUT
cd /tmp
for v in "tile5:--only single --kernels 1 --window 5" "tile7:--only single --kernels 1 --window 7" "stream5:--only single --kernels 0 --window 5" "tol:--only pyramidal --arith 2"; do
  n=${v%%:*}; a=${v#*:}
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/$n/a -- python3 $R/tools/kbench.py --pairs 32 --reps 3 $a > $OUT/$n.a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR \
    --output-format csv -d $OUT/$n/b -- python3 $R/tools/kbench.py --pairs 32 --reps 3 $a > $OUT/$n.b.log 2>&1
done
cd $R
python3 tools/issue_bounds.py gpurun_out/pmc_$TAG profiles/r03_valu_wall.txt profiles/${TAG}_issue_bounds.json gpurun_out/pmc_${TAG}_single > gpurun_out/issue_$TAG.log 2>&1 || tail -5 gpurun_out/issue_$TAG.log
echo "== counters done"
python3 -m pytest tests/test_gpu_round4.py -q -k "within_tolerance" > gpurun_out/tol_tests_$TAG.log 2>&1 && cp gpurun_out/tolerant_epe.json profiles/${TAG}_tolerant_epe.json || true
python3 tools/host_latency.py > profiles/${TAG}_host_latency.txt 2>&1 && python3 tools/host_latency.py batch >> profiles/${TAG}_host_latency.txt 2>&1 || true
python3 tools/measure_configs.py profiles/${TAG}_configs.json > gpurun_out/cfg_$TAG.log 2>&1 || tail -3 gpurun_out/cfg_$TAG.log
python3 bench.py --config 4k64 --steps 5 --warmup 1 --no-one-pair > gpurun_out/bench_4k64_$TAG.log 2>&1 || tail -3 gpurun_out/bench_4k64_$TAG.log
grep -h '^{' gpurun_out/bench_4k64_$TAG.log | tail -1 > profiles/${TAG}_bench_4k64_1gpu.json
echo "== configs done"
python3 bench.py > gpurun_out/bench_$TAG.log 2>&1
grep -h '^{' gpurun_out/bench_$TAG.log | tail -1 > profiles/${TAG}_bench.json
python3 -c "import json; b=json.load(open('profiles/${TAG}_bench.json')); print(b['value'], b['tolerance_mode']['value'], b['roofline']['frac'], b['tolerance_mode']['roofline'])"
rm -rf gpurun_out/profiles_$TAG && mkdir -p gpurun_out/profiles_$TAG && cp -r profiles/${TAG}_* gpurun_out/profiles_$TAG/
