#!/bin/bash
# rocprof --stats of an 8-rank shard render for the tail variants
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp
for v in "TGPU_TAIL_COOP=0" "TGPU_TAIL_COOP=1"; do
  export $v
  rm -rf /tmp/tp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python3 $R/profiles/shard_time.py 8 > /tmp/tp.log 2>&1
  echo "== $v: $(grep N= /tmp/tp.log)"
  python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/tp/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_tail' in r['Name']: print('   ', r['Name'][:60], 'calls', r['Calls'], 'avg ms', float(r['AverageNs'])/1e6, 'total ms', float(r['TotalDurationNs'])/1e6)
PY
  unset TGPU_TAIL_COOP TGPU_TAIL_MAX
done
