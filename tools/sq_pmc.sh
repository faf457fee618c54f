#!/bin/bash
# usage (GPU box): tools/sq_pmc.sh <name> <script args...> — one rocprofv3 --pmc pass of eight SQ counters (kernel trace only) over a
# command, per-kernel averages as a markdown table: where a kernel's wave cycles go (issuing / parked at a wait / stalled on issue)
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES \
  -d $root/gpurun_out/$name -o p -- python $root/"$@" > $root/gpurun_out/$name.log 2>&1
cd $root
db=$(find gpurun_out/$name -name "*.db" | head -1)
echo "command: rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES -- python $*"
echo
python tools/rocpd_summary.py $db --pmc mke
rm -rf gpurun_out/$name
