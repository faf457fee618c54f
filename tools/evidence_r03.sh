#!/bin/bash
# Round-3 evidence batch (run on the GPU box via gpurun): everything under gpurun_out/r03_*; copy what is to be judged into profiles/.
# Order matters: the PMC passes come first and their JSON is copied into profiles/ ON THE BOX, so that every bench line below
# quotes the byte counts of this very build (bench.py keys them by a hash of the kernel sources).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout 600 tools/pmc_passes.sh c2 > gpurun_out/r03_pmc_c2.log 2>&1
timeout 900 tools/pmc_passes.sh c5 --steps 40 > gpurun_out/r03_pmc_c5.log 2>&1
cp gpurun_out/r03_pmc_c2.json gpurun_out/r03_pmc_c5.json profiles/ 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_c2_20steps.json.log 2>&1   # the driver's invocation
timeout 600 python bench.py > gpurun_out/r03_bench_c2.json.log 2>&1
timeout 600 tools/prof.sh r03_trace_c2 10 bench.py --no-cpu-baseline --no-variants > gpurun_out/r03_kernel_trace_c2.md 2>&1
timeout 900 python bench.py --config c5 --steps 600 --no-cpu-baseline > gpurun_out/r03_bench_c5.json.log 2>&1
timeout 900 tools/prof.sh r03_trace_c5 8 bench.py --config c5 --steps 600 --no-cpu-baseline > gpurun_out/r03_kernel_trace_c5.md 2>&1
timeout 900 python bench.py --config c5 --force-sharded --steps 300 --no-cpu-baseline > gpurun_out/r03_bench_c5_sharded_g1.json.log 2>&1
timeout 300 python bench.py --force-sharded --steps 552 --no-cpu-baseline > gpurun_out/r03_bench_c2_sharded_g1.json.log 2>&1
ls -la gpurun_out | grep r03_
