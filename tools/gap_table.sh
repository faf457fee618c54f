#!/bin/bash
# usage (GPU box): tools/gap_table.sh <name> [bench args...]   — rocprofv3 kernel trace of the driver-style line
# `python bench.py --steps 20 --warmup 5` and the per-dispatch gap table of its timed region (tools/rocpd_summary.py --gaps)
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $root/gpurun_out/$name -o p -- python $root/bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline "$@" > $root/gpurun_out/$name.log 2>&1
cd $root
db=$(find gpurun_out/$name -name "*.db" | head -1)
echo "command: rocprofv3 --kernel-trace -- python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline $*"
echo
python tools/rocpd_summary.py $db --gaps 3 20
echo
echo "result line of the same invocation (profiler attached):"
grep '^{' gpurun_out/$name.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ("value","ms_per_step","steps","warmup")}))'
rm -rf gpurun_out/$name
