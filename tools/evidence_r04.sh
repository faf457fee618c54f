#!/bin/bash
# Round-4 evidence batch (GPU box, via gpurun): everything under gpurun_out/r04_*; copy what is to be judged into profiles/.
# Order matters: the PMC passes come first and their JSON is copied into profiles/ ON THE BOX, so that every bench line below
# quotes the byte counts of this very build (bench.py keys them by a hash of the kernel sources).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
export MKE_ROUND=r04
timeout 600 tools/pmc_passes.sh c2 > gpurun_out/r04_pmc_c2.log 2>&1
timeout 900 tools/pmc_passes.sh c5 --steps 40 > gpurun_out/r04_pmc_c5.log 2>&1
cp gpurun_out/r04_pmc_c2.json gpurun_out/r04_pmc_c5.json profiles/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_c2_20steps.json.log 2>&1   # the driver's invocation (C5 variant inside)
timeout 600 python bench.py > gpurun_out/r04_bench_c2.json.log 2>&1
timeout 600 tools/prof.sh r04_trace_c2 10 bench.py --no-cpu-baseline --no-variants > gpurun_out/r04_kernel_trace_c2.md 2>&1
timeout 900 python bench.py --config c5 --steps 600 --no-cpu-baseline > gpurun_out/r04_bench_c5.json.log 2>&1
timeout 900 tools/prof.sh r04_trace_c5 8 bench.py --config c5 --steps 600 --no-cpu-baseline > gpurun_out/r04_kernel_trace_c5.md 2>&1
timeout 900 python bench.py --config c5 --force-sharded --steps 300 --no-cpu-baseline > gpurun_out/r04_bench_c5_sharded_g1.json.log 2>&1
timeout 300 python bench.py --force-sharded --steps 552 --no-cpu-baseline > gpurun_out/r04_bench_c2_sharded_g1.json.log 2>&1
timeout 300 tools/gap_table.sh r04_gap_c2 > gpurun_out/r04_gap_table_c2.md 2>&1
ATTR_LIBRARY=0 timeout 300 tools/prof.sh r04_attr_trace 8 tools/attr_prof.py 400 > gpurun_out/r04_attr_trace.md 2>&1
timeout 300 python tools/attr_prof.py 400 > gpurun_out/r04_attr.log 2>&1
timeout 300 python tools/knn_bench.py > gpurun_out/r04_knn.log 2>&1
timeout 300 python tools/ae_bench.py > gpurun_out/r04_ae.log 2>&1
for w in 1 8; do timeout 300 tools/prof.sh r04_oc_g${w}_c2_trace 40 tools/oc_rank_compute.py --world $w --config c2 > gpurun_out/r04_oc_g${w}_c2_trace.md 2>&1; done
timeout 600 tools/prof.sh r04_oc_g8_c5_trace 40 tools/oc_rank_compute.py --world 8 --config c5 > gpurun_out/r04_oc_g8_c5_trace.md 2>&1
# the driver's N > 1 shape with no launcher around it, two ranks sharing this GPU (collectives staged: a dry run, labelled as such)
MKE_BENCH_COMM=staged timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r04_bench_gpus2_staged.json.log 2>&1
ls -la gpurun_out | grep r04_
