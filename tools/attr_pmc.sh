#!/bin/bash
export ATTR_LIBRARY=0   # tools/attr_prof.py: the native step only
# SQ counters of the attribute step's kernels (separate --pmc passes with --kernel-trace only): tools/attr_prof.py 100
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d $root/gpurun_out/attr_pmc1 -o p -- python $root/tools/attr_prof.py 100 > $root/gpurun_out/attr_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM -d $root/gpurun_out/attr_pmc2 -o p -- python $root/tools/attr_prof.py 100 > $root/gpurun_out/attr_pmc2.log 2>&1
cd $root
for d in attr_pmc1 attr_pmc2; do echo "== $d"; tail -2 gpurun_out/$d.log | cut -c1-200; python tools/rocpd_pmc.py $(find gpurun_out/$d -name "*.db" | head -1) 2>&1 | grep -i "attr_conv\|tallsplit\|tail\|kernel\|error" | head -40; done
