#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <outdir-name> <top-n> <python script and args...>
# rocprofv3 kernel trace of the command, summarised per (kernel, grid) by tools/rocpd_summary.py
name=$1; top=$2; shift 2
script=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$name -o p -- python $root/$script "$@" > $root/gpurun_out/$name.log 2>&1
cd $root
python tools/rocpd_summary.py $(find gpurun_out/$name -name "*.db" | head -1) $top g
rm -rf gpurun_out/$name   # the raw database is tens of MB; gpurun merges back at most 64 MiB
