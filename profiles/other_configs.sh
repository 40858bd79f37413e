#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python assets/instances/make_c4.py /tmp/c4 > /dev/null
timeout -k 10 300 python bench.py --scene assets/cornell-box/scene_sobol.json --workload "configs[3]: cornell-box 1000x563 sobol" --steps 64 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c3.json
timeout -k 10 300 python bench.py --scene /tmp/c4/c4.json --workload "procedural 10M-triangle instanced mesh 1280x720 (BASELINE configs[3]), 1xMI355X" --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c4.json
timeout -k 10 300 python bench.py --scene assets/materialtest/c5.json --workload "C5: materialtest 1280x720, rough_dielectric T-Ball + homogeneous medium inside" --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c5.json
for c in c3 c4 c5; do python - $c <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/{sys.argv[1]}.json')); r=d['roofline']
print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['kernel_ms_per_step'])
PY
done
