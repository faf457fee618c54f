#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d $root/gpurun_out/knn_pmc1 -o p -- python $root/tools/knn_bench.py > $root/gpurun_out/knn_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $root/gpurun_out/knn_pmc2 -o p -- python $root/tools/knn_bench.py > $root/gpurun_out/knn_pmc2.log 2>&1
cd $root
for d in knn_pmc1 knn_pmc2; do echo "== $d"; tail -3 gpurun_out/$d.log | cut -c1-200; python tools/rocpd_pmc.py $(find gpurun_out/$d -name "*.db" | head -1) 2>&1 | grep -i "sim_select\|topk\|kernel\|error" | head -8; done
