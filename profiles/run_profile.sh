#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):
#   bash profiles/run_profile.sh <tag> [command ...]
# default command: the headline bench (python3 bench.py --steps 64 --warmup 32 --batch 32 --no-cpu-baseline --no-extras --no-per-frame-leg --no-live-traffic --repeats 1).
# Writes rocprofv3 outputs under gpurun_out/prof_<tag>/ ; profiles/summarize.py then distils them into
# profiles/<tag>_summary.txt.  Counters are collected in their own passes (never together with a trace);
# FETCH_SIZE and WRITE_SIZE each in a pass of their own, as MI355X_MICROARCH.md prescribes.
TAG=${1:-r02}
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
# the product library must exist before the profiler starts (no compile under rocprofv3)
python3 -m ray_tracer_2_amd.build > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
if [ $# -gt 0 ]; then CMD="$*"; else CMD="python3 $REPO/bench.py --steps 64 --warmup 32 --batch 32 --no-cpu-baseline --no-extras --no-per-frame-leg --no-live-traffic --repeats 1"; fi
echo "$CMD" > $OUT/command.txt
cd /tmp && export TMPDIR=/tmp
fail=0
pass() {   # pass <name> <required 0|1> <rocprofv3 args...>
    local name=$1 required=$2
    shift 2
    rocprofv3 "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1
    local rc=$?
    echo "pass $name: exit $rc" | tee -a $OUT/status.txt
    if [ $rc -ne 0 ]; then
        tail -3 $OUT/$name.log
        if [ $required -eq 1 ]; then fail=1; fi
    fi
}
: > $OUT/status.txt
pass trace 1 --kernel-trace --stats
[ $fail -eq 0 ] && pass pmc_fetch 1 --pmc FETCH_SIZE
[ $fail -eq 0 ] && pass pmc_write 1 --pmc WRITE_SIZE
[ $fail -eq 0 ] && pass pmc_sq 1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
# optional counters (a failed pass is reported above and in status.txt, not hidden)
[ $fail -eq 0 ] && pass pmc_sq2 0 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM
[ $fail -eq 0 ] && pass pmc_l2 0 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
cd $REPO
exit $fail
