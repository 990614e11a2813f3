#!/bin/bash
# VALU instruction count + kernel time of the headline bench under RT2_OPTIONS (A/B of knobs)
set -e
python3 -m ray_tracer_2_amd.build > /dev/null 2>&1   # no compile under the profiler
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$REPO/gpurun_out/prof_valu_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
for f in glob.glob(out+"/trace/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "persistent" in r["Name"]: print("avg ns",r["AverageNs"])
for f in glob.glob(out+"/sq/**/*counter_collection.csv",recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "persistent" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, "%.4g"%(sum(v)/len(v)))
PY
