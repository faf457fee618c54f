#!/bin/bash
# Evidence for the relation-view step only (run on the GPU box via gpurun): writes under gpurun_out/ev_*
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/ev_bench_c2_20steps.log 2>&1   # the driver's invocation, on the cold box
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -6) > gpurun_out/ev_pytest.log
timeout 600 tools/pmc_passes.sh c2 > gpurun_out/ev_pmc_c2.log 2>&1
timeout 900 tools/pmc_passes.sh c5 > gpurun_out/ev_pmc_c5.log 2>&1
cp gpurun_out/r03_pmc_c2.json gpurun_out/r03_pmc_c5.json profiles/   # this box's copy only: the bench lines below quote the passes just taken
timeout 600 python bench.py > gpurun_out/ev_bench_c2.log 2>&1
timeout 900 python bench.py --config c5 > gpurun_out/ev_bench_c5.log 2>&1
timeout 400 tools/prof.sh ev_trace_c2 10 bench.py --no-cpu-baseline --no-variants > gpurun_out/ev_trace_c2.md 2>&1
timeout 600 tools/prof.sh ev_trace_c5 8 bench.py --config c5 --steps 300 --no-cpu-baseline > gpurun_out/ev_trace_c5.md 2>&1
timeout 300 python bench.py --force-sharded --steps 552 --no-cpu-baseline > gpurun_out/ev_bench_sharded_oc.log 2>&1
MKE_SHARD_MODE=rowfetch timeout 300 python bench.py --force-sharded --steps 184 --no-cpu-baseline > gpurun_out/ev_bench_sharded_rowfetch.log 2>&1
cat gpurun_out/ev_pytest.log
