#!/bin/bash
# Evidence batch for the relation-view step only (run on the GPU box via gpurun): writes under gpurun_out/ev_*
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4) > gpurun_out/ev_pytest.log
timeout 600 python bench.py > gpurun_out/ev_bench_c2.log 2>&1
timeout 900 python bench.py --config c5 > gpurun_out/ev_bench_c5.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/ev_bench_c2_20steps.log 2>&1
timeout 600 tools/pmc_passes.sh c2 > gpurun_out/ev_pmc_c2.log 2>&1
timeout 900 tools/pmc_passes.sh c5 > gpurun_out/ev_pmc_c5.log 2>&1
timeout 400 tools/prof.sh ev_trace_c2 10 bench.py --no-cpu-baseline --no-variants > gpurun_out/ev_trace_c2.md 2>&1
timeout 600 tools/prof.sh ev_trace_c5 8 bench.py --config c5 --steps 300 --no-cpu-baseline > gpurun_out/ev_trace_c5.md 2>&1
ls gpurun_out | grep ev_ | wc -l
