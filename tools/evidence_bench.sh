#!/bin/bash
# Bench lines + full GPU test suite (run on the GPU box via gpurun): writes under gpurun_out/ev_*
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/ev_pytest_full.log 2>&1
grep -E "passed|failed|error" gpurun_out/ev_pytest_full.log | tail -3 > gpurun_out/ev_pytest.log
timeout 600 python bench.py > gpurun_out/ev_bench_c2.log 2>&1
timeout 900 python bench.py --config c5 > gpurun_out/ev_bench_c5.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/ev_bench_c2_20steps.log 2>&1
timeout 300 python bench.py --force-sharded --steps 552 --no-cpu-baseline > gpurun_out/ev_bench_sharded_oc.log 2>&1
MKE_SHARD_MODE=rowfetch timeout 300 python bench.py --force-sharded --steps 184 --no-cpu-baseline > gpurun_out/ev_bench_sharded_rowfetch.log 2>&1
timeout 300 python tools/oc_bench.py > gpurun_out/ev_oc.log 2>&1
OC_CFG=c5 timeout 400 python tools/oc_bench.py >> gpurun_out/ev_oc.log 2>&1
timeout 300 tools/prof.sh ev_oc_c2 8 tools/oc_bench.py > gpurun_out/ev_oc_trace_c2.md 2>&1
rm -f gpurun_out/ev_pytest_full.log.tmp
cat gpurun_out/ev_pytest.log
