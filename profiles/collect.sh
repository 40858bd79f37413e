#!/bin/bash
# Collects the round's evidence on an MI355X box (run through gpurun from the repo root):
#   1. bench.py JSON line                         -> gpurun_out/prof/bench.json
#   2. rocprofv3 --kernel-trace --stats           -> gpurun_out/prof/stats/
#   3. PMC passes (own runs, kernel-trace only)   -> gpurun_out/prof/pmc_{fetch,write}/
# Afterwards `python profiles/summarize.py rNN` copies the summaries into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py > $OUT/bench.log 2>&1 && tail -1 $OUT/bench.log > $OUT/bench.json || { echo "bench failed"; tail -5 $OUT/bench.log; exit 1; }
BENCH="python3 $R/bench.py --steps 16 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || { echo "rocprof stats failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --no-roofline > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $BENCH --no-roofline > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
echo collected; cat $OUT/bench.json | cut -c1-300
