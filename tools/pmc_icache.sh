#!/bin/bash
set -e
python3 -m ray_tracer_2_amd.build > /dev/null 2>&1   # no compile under the profiler
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_ic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_IFETCH --output-format csv -d $OUT/ic -- $BENCH > $OUT/ic.log 2>&1 || { echo "pass ic failed"; tail -3 $OUT/ic.log; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1 || { echo "pass sq failed"; tail -3 $OUT/sq.log; }
find $OUT -name '*counter_collection.csv'
