#!/bin/bash
# Round-2 evidence batch (run on the GPU box via gpurun): writes everything under gpurun_out/ev_*
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -6) > gpurun_out/ev_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/ev_bench_c2_20steps.log 2>&1   # the driver's invocation, on the cold box
timeout 600 tools/pmc_passes.sh c2 > gpurun_out/ev_pmc_c2.log 2>&1
timeout 900 tools/pmc_passes.sh c5 > gpurun_out/ev_pmc_c5.log 2>&1
cp gpurun_out/r03_pmc_c2.json gpurun_out/r03_pmc_c5.json profiles/   # this box's copy only: the bench lines below quote the passes just taken
timeout 600 python bench.py > gpurun_out/ev_bench_c2.log 2>&1
timeout 900 python bench.py --config c5 > gpurun_out/ev_bench_c5.log 2>&1
timeout 400 tools/prof.sh ev_trace_c2 10 bench.py --no-cpu-baseline --no-variants > gpurun_out/ev_trace_c2.md 2>&1
timeout 600 tools/prof.sh ev_trace_c5 8 bench.py --config c5 --steps 300 --no-cpu-baseline > gpurun_out/ev_trace_c5.md 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/ev_gemm.md 2>&1
timeout 300 python tools/ae_bench.py > gpurun_out/ev_ae.log 2>&1
AE_ACT=tanh timeout 300 python tools/ae_bench.py >> gpurun_out/ev_ae.log 2>&1
AE_ROWS=20000 timeout 300 tools/prof.sh ev_ae_trace 14 tools/ae_bench.py > gpurun_out/ev_ae_trace.md 2>&1
timeout 300 python tools/oc_bench.py > gpurun_out/ev_oc.log 2>&1
OC_CHUNKS=2 timeout 300 python tools/oc_bench.py >> gpurun_out/ev_oc.log 2>&1
OC_CFG=c5 timeout 400 python tools/oc_bench.py >> gpurun_out/ev_oc.log 2>&1
timeout 300 tools/prof.sh ev_oc_c2 8 tools/oc_bench.py > gpurun_out/ev_oc_trace_c2.md 2>&1
OC_CFG=c5 timeout 400 tools/prof.sh ev_oc_c5 8 tools/oc_bench.py > gpurun_out/ev_oc_trace_c5.md 2>&1
timeout 300 python bench.py --force-sharded --steps 552 --no-cpu-baseline > gpurun_out/ev_bench_sharded_oc.log 2>&1
MKE_SHARD_MODE=rowfetch timeout 300 python bench.py --force-sharded --steps 184 --no-cpu-baseline > gpurun_out/ev_bench_sharded_rowfetch.log 2>&1
timeout 300 tools/prof.sh ev_attr 10 tools/attr_prof.py 400 > gpurun_out/ev_attr_trace.md 2>&1
timeout 200 python tools/attr_prof.py 400 > gpurun_out/ev_attr.log 2>&1
timeout 300 python tools/epoch_bench.py > gpurun_out/ev_epoch.log 2>&1
ls gpurun_out | grep ev_ | wc -l
