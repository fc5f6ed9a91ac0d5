#!/bin/bash
# HBM traffic of the dominant kernel (finest-level fused LK iteration) from rocprofv3 PMC
# counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes,
# both in KiB-like units of 1024 B; on gfx950 FETCH_SIZE reads half the bytes of a wide
# coalesced stream, so it is doubled.  Writes profiles/<tag>_hbm_traffic.json.
# Usage (on the GPU box, from the repo root):  bash tools/measure_traffic.sh r01
set -e
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/traffic_$TAG
rm -rf $OUT && mkdir -p $OUT/fetch $OUT/write
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-pair --no-live-traffic > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-pair --no-live-traffic > $OUT/write.log 2>&1
cd $R
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
def per_launch(sub, counter):
    rows = []
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_lkw<2, 1" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                rows.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    big = max(g for g, _ in rows)                       # finest level = largest grid
    vals = [v for g, v in rows if g == big]
    return sum(vals) / len(vals), len(vals), big
fetch, nf, grid = per_launch("fetch", "FETCH_SIZE")
write, nw, _ = per_launch("write", "WRITE_SIZE")
bench = [json.loads(l) for l in open(f"{out}/fetch.log") if l.startswith("{")][-1]   # the bench line of the profiled run
pairs = bench["config"]["pairs_per_gpu_per_step"]
res = {"kernel": "k_lkw<2, MODE_ITER> finest level", "pairs": pairs, "shape": [1080, 1920], "grid_size": grid, "launches_averaged": [nf, nw],
       "FETCH_SIZE_raw_units_1024B": fetch, "WRITE_SIZE_raw_units_1024B": write,
       "fetch_bytes_corrected_x2": fetch * 1024 * 2, "write_bytes": write * 1024,
       "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section)"}
json.dump(res, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
