#!/bin/bash
# Any set of counters for every kernel of one bench run (own pass, kernel-trace only); prints per-kernel sums per launch.
#   profiles/pmc_any.sh <tag> "<COUNTER ...>" [env assignments for bench.py ...]   -> gpurun_out/prof_any_<tag>/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; CTRS=$2; shift 2
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/prof_any_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $R/bench.py --steps ${SQ_STEPS:-8} --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS > $OUT/run.log 2>&1 || { echo "pmc pass failed"; tail -5 $OUT/run.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void tg::", "").replace("tg::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in sorted(tot):
    if k.startswith("k_"):
        n = max(1, len(calls[k]))
        print(f"{k:36s} n={n:4d} " + " ".join(f"{c}={v/n:.4g}" for c, v in sorted(tot[k].items())))
PY
