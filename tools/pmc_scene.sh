#!/bin/bash
# WRITE_SIZE / FETCH_SIZE (and optionally more counters) of the render kernels of tools/bench_scene.py under option
# variants, one PMC pass per counter (never together with a trace):
#   PS_ARGS="0 8" PS_ENV="BS_MESHES=340 BS_DETAIL=8 BS_BATCH=8" bash tools/pmc_scene.sh <tag> "" "pixel_cache=0" ...
# each argument after the tag is one BS_OPTS variant.  PS_CTRS="WRITE_SIZE FETCH_SIZE" by default; PS_LIB=<.so> profiles
# another build (tools/build_variant.sh).  Prints MB per frame per kernel (FETCH_SIZE doubled, as the guide prescribes).
TAG=$1
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CTRS=${PS_CTRS:-"WRITE_SIZE FETCH_SIZE"}
ARGS=${PS_ARGS:-"0 8"}
OUT=$REPO/gpurun_out/pmcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in $PS_ENV; do export "$e"; done
[ -n "$PS_LIB" ] && export RT2_LIB=$PS_LIB
export BS_COUNTERS=0
i=0
for opt in "$@"; do
  i=$((i + 1))
  for ctr in $CTRS; do
    d=$OUT/v${i}_$ctr
    rm -rf $d
    BS_OPTS="$opt" BS_JSON=$d.json rocprofv3 --pmc $ctr --output-format csv -d $d -- python3 $REPO/tools/bench_scene.py $ARGS > $d.log 2>&1 || { echo "pass failed: $opt $ctr"; tail -3 $d.log; }
    python3 - "$d" "$opt" $ctr <<'PY'
import collections, csv, glob, json, sys
d, opt, ctr = sys.argv[1:4]
per = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
try:
    j = json.load(open(d + ".json"))
    fpl, ms, ft = j.get("frames_per_launch", 1), j.get("ms_per_frame", 0.0), j.get("frames_total", 0)
except Exception:
    fpl, ms, ft = 1, 0.0, 0
k = 2048 if ctr == "FETCH_SIZE" else 1024
print(f"[{opt or 'default'}] {ctr}  ({fpl} frames per launch, {ms:.3f} ms/frame under the profiler)")
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) * k < 1e6:
        continue
    print(f"    {name:90s} launches {len(v):4d}  total {sum(v) * k / 1e6:10.1f} MB  mean per launch {sum(v) / len(v) * k / 1e6:9.1f} MB"
          + (f"  = {sum(v) * k / 1e6 / ft:8.1f} MB per frame ({ft} frames)" if ft else ""))
PY
  done
done
