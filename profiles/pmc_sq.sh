#!/bin/bash
# SQ counters of every kernel of one bench run (own pass, kernel-trace only): wave cycles, waits, VALU issue and lane occupancy.
#   profiles/pmc_sq.sh <tag> [env assignments for bench.py ...]   -> gpurun_out/prof_sq_<tag>/ + a per-kernel summary on stdout
# lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES;
# VALU issue share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (quad-cycle units on both sides, MI355X_MICROARCH.md "cycle constants")
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/prof_sq_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD \
    --output-format csv -d $OUT -- python3 $R/bench.py --steps ${SQ_STEPS:-8} --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS > $OUT/run.log 2>&1 || { echo "pmc sq failed"; tail -5 $OUT/run.log; exit 1; }
python3 $R/profiles/summarize_sq.py $OUT
