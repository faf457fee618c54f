#!/bin/bash
# Round-3 evidence batch 2 (GPU box, via gpurun): the side figures DESIGN.md quotes, each into gpurun_out/r03_*.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -6) > gpurun_out/r03_pytest_gpu.log
(timeout 600 python -m pytest tests/test_schedule_trace_gpu.py -q -s 2>&1 | grep -E "worst phase-loss|passed|failed") > gpurun_out/r03_schedule_trace.log
timeout 600 python tools/c5_variance.py > gpurun_out/r03_c5_variance.log 2>&1
timeout 600 python tools/c5_skew.py > gpurun_out/r03_c5_skew.log 2>&1
timeout 1200 tools/c5_contig_ab.sh > gpurun_out/r03_c5_contig_ab.log 2>&1
ATTR_LIBRARY=0 timeout 300 tools/prof.sh r03_attr 9 tools/attr_prof.py 400 > gpurun_out/r03_attr_trace.md 2>&1
timeout 200 python tools/attr_prof.py 400 > gpurun_out/r03_attr.log 2>&1
timeout 300 python tools/epoch_bench.py > gpurun_out/r03_epoch.log 2>&1
timeout 300 python tools/oc_bench.py > gpurun_out/r03_oc.log 2>&1
OC_CFG=c5 timeout 400 python tools/oc_bench.py >> gpurun_out/r03_oc.log 2>&1
timeout 300 tools/prof.sh r03_oc_c2 8 tools/oc_bench.py > gpurun_out/r03_oc_trace_c2.md 2>&1
timeout 300 python tools/ae_bench.py > gpurun_out/r03_ae.log 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r03_gemm.md 2>&1
timeout 200 python tools/micro/dflat_wt.py > gpurun_out/r03_dflat_wt.log 2>&1
export MKE_BENCH_COMM=staged
for n in 2 4 8; do
  timeout 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n tools/multi_gpu_selftest.py 2>/dev/null | grep '^{' >> gpurun_out/r03_selftest_staged.log
done
ls gpurun_out | grep r03_ | wc -l
