#!/bin/bash
# bench.py --config c5 in three fresh processes: how much the step time moves between allocations of the same tables
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 600 python bench.py --config c5 --steps 400 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('run $i: %.1f us/step, k_triple_score %.1f us, update %.1f us, %.3f G triples/s' % (d['ms_per_step']*1e3, r['avg_launch_us'], r['update_kernel']['avg_launch_us'], d['value']/1e9))"
done
