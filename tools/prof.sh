#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <outdir-name> <top-n> <python script and args...>
# rocprofv3 kernel trace of the command, summarised per (kernel, grid) by tools/rocpd_summary.py.  The command's own last
# JSON line (bench.py's result line, measured in THIS invocation) is printed under the table, so a summary's kernel times and
# the step time they are compared with come from one run; the headline kernels' per-dispatch rows are kept as
# gpurun_out/<name>_dispatches.csv (the raw database is deleted: tens of MB).
name=$1; top=$2; shift 2
script=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$name -o p -- python $root/$script "$@" > $root/gpurun_out/$name.log 2>&1
cd $root
db=$(find gpurun_out/$name -name "*.db" | head -1)
echo "command: rocprofv3 --kernel-trace --stats -- python $script $*"
echo
python tools/rocpd_summary.py $db $top g
python tools/rocpd_summary.py $db --dump "k_triple_score" gpurun_out/${name}_dispatches.csv > /dev/null
python tools/rocpd_summary.py $db --dump "k_rows_update_multi" gpurun_out/${name}_dispatches_update.csv > /dev/null
echo
echo "result line of the same invocation (profiler attached: its HIP-event figures — roofline.avg_launch_us, the step breakdown —"
echo "carry the profiler's per-dispatch overhead; the kernel times to read are the table's averages):"
echo
grep '^{' gpurun_out/$name.log | tail -1 | python -c '
import json, sys
try:
    d = json.loads(sys.stdin.read())
except Exception:
    sys.exit(0)
if not isinstance(d, dict) or "value" not in d:
    print("```json"); print(json.dumps(d, indent=1)); print("```"); sys.exit(0)
r = d.get("roofline") or {}
keep = {k: d[k] for k in ("value", "ms_per_step", "steps", "warmup") if k in d}
keep["workload"] = d.get("config", {}).get("workload")
keep["roofline"] = {k: r.get(k) for k in ("avg_launch_us", "achieved", "frac", "frac_basis", "achieved_counter", "frac_counter", "traffic", "step_breakdown_us") if k in r}
if "update_kernel" in r:
    keep["update_kernel_avg_launch_us"] = r["update_kernel"]["avg_launch_us"]
print("```json"); print(json.dumps(keep, indent=1)); print("```")
'
rm -rf gpurun_out/$name   # the raw database is tens of MB; gpurun merges back at most 64 MiB
