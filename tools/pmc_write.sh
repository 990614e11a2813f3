#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the headline bench's render kernel under RT2_OPTIONS variants (one PMC pass each):
#   bash tools/pmc_write.sh "batch_tile_major=0" "batch_tile_major=1" ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -m ray_tracer_2_amd.build > /dev/null 2>&1   # no compile under the profiler
cd /tmp && export TMPDIR=/tmp
for opt in "$@"; do
  for ctr in WRITE_SIZE FETCH_SIZE; do
    out=$REPO/gpurun_out/pmcw_$(echo "$opt$ctr" | tr -c 'a-zA-Z0-9' '_')
    rm -rf $out
    RT2_OPTIONS="$opt" rocprofv3 --pmc $ctr --output-format csv -d $out -- python3 $REPO/bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-extras > $out.log 2>&1 || { echo "pass failed: $opt $ctr"; tail -3 $out.log; }
    python3 - "$out" "$opt" $ctr <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "rt_render" in r["Kernel_Name"]]
k = 2048 if sys.argv[3] == "FETCH_SIZE" else 1024
print(f"{sys.argv[2]:40s} {sys.argv[3]}: MB per frame of the render kernel (32 frames per launch):", [round(x * k / 1e6 / 32, 1) for x in v])
PY
  done
done
