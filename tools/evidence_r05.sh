#!/bin/bash
# Round-5 evidence batch (GPU box, via gpurun): everything under gpurun_out/r05_*; copy what is to be judged into profiles/.
# Order matters: the PMC passes come first and their JSON is copied into profiles/ ON THE BOX, so that every bench line below
# quotes the byte counts of this very build (bench.py keys them by a hash of the kernel sources).
# usage: tools/evidence_r05.sh [part]   (part: all | core | sharded | rest | tests)
part=${1:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
export MKE_ROUND=r05
o=gpurun_out
if [ $part = all ] || [ $part = core ]; then
timeout 900 tools/pmc_passes.sh c2 > $o/r05_pmc_c2.log 2>&1
timeout 1200 tools/pmc_passes.sh c5 --steps 40 > $o/r05_pmc_c5.log 2>&1
cp $o/r05_pmc_c2.json $o/r05_pmc_c5.json profiles/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $o/r05_bench_c2_20steps.json.log 2>&1   # the driver's invocation (C5 + Zipf variants inside)
timeout 900 python bench.py > $o/r05_bench_c2.json.log 2>&1
timeout 900 tools/prof.sh r05_trace_c2 10 bench.py --no-cpu-baseline --no-variants --windows 2 > $o/r05_kernel_trace_c2.md 2>&1
timeout 900 python bench.py --config c5 --steps 100 --no-cpu-baseline > $o/r05_bench_c5.json.log 2>&1
timeout 900 tools/prof.sh r05_trace_c5 8 bench.py --config c5 --steps 100 --windows 6 --no-cpu-baseline > $o/r05_kernel_trace_c5.md 2>&1
timeout 600 tools/prof.sh r05_trace_c2_zipf 10 bench.py --no-cpu-baseline --no-variants --zipf 1.0 --steps 184 --windows 3 > $o/r05_kernel_trace_c2_zipf.md 2>&1
MKE_BENCH_HOT=0 timeout 600 python bench.py --zipf 1.0 --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $o/r05_bench_c2_zipf_nohub.json.log 2>&1
timeout 600 python bench.py --zipf 1.0 --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $o/r05_bench_c2_zipf.json.log 2>&1
timeout 300 tools/gap_table.sh r05_gap_c2 > $o/r05_gap_table_c2.md 2>&1
fi
if [ $part = all ] || [ $part = sharded ]; then
# the sharded step on one GPU: (i) one-rank path without collectives, (ii) the G > 1 path forced over a one-rank RCCL group through
# multike_amd/rccl.py (collectives on the compute stream) and (iii) over torch.distributed
timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/r05_bench_c2_sharded_g1.json.log 2>&1
MKE_OC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/r05_bench_c2_sharded_g1_rccl.json.log 2>&1
MKE_OC_COMM=torch MKE_OC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/r05_bench_c2_sharded_g1_torchdist.json.log 2>&1
timeout 900 python bench.py --config c5 --force-sharded --steps 100 --windows 5 --no-cpu-baseline > $o/r05_bench_c5_sharded_g1.json.log 2>&1
MKE_OC_FORCE_COLLECTIVES=1 timeout 900 python bench.py --config c5 --force-sharded --steps 100 --windows 5 --no-cpu-baseline > $o/r05_bench_c5_sharded_g1_rccl.json.log 2>&1
MKE_OC_COMM=torch MKE_OC_FORCE_COLLECTIVES=1 timeout 900 python bench.py --config c5 --force-sharded --steps 100 --windows 5 --no-cpu-baseline > $o/r05_bench_c5_sharded_g1_torchdist.json.log 2>&1
# rank 0 of 8: per-kernel tables (rocprofv3), then the schedule with the modelled wire
timeout 400 tools/prof.sh r05_oc_g8_c2_trace 40 tools/oc_rank_compute.py --world 8 --config c2 > $o/r05_oc_g8_c2_trace.md 2>&1
timeout 800 tools/prof.sh r05_oc_g8_c5_trace 40 tools/oc_rank_compute.py --world 8 --config c5 > $o/r05_oc_g8_c5_trace.md 2>&1
for cfg in c2 c5; do
  timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --chunks 1 --prefetch 2>/dev/null | tail -1 > $o/r05_oc_${cfg}_nowire.json
  for ch in 1 2 3; do
    timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --chunks $ch --wire-gbps 376 --latency-us 15 --prefetch 2>/dev/null | tail -1 > $o/r05_oc_${cfg}_ch${ch}_wire.json
  done
done
timeout 600 python tools/oc_rank_compute.py --world 8 --config c2 --zipf 1.0 --chunks 1 --prefetch 2>/dev/null | tail -1 > $o/r05_oc_c2_zipf_nowire.json
MKE_BENCH_COMM=staged timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > $o/r05_bench_gpus2_staged.json.log 2>&1
fi
if [ $part = all ] || [ $part = rest ]; then
ATTR_LIBRARY=0 timeout 300 tools/prof.sh r05_attr_trace 8 tools/attr_prof.py 400 > $o/r05_attr_trace.md 2>&1
timeout 300 python tools/attr_prof.py 400 > $o/r05_attr.log 2>&1
timeout 300 python tools/knn_bench.py > $o/r05_knn.log 2>&1
timeout 300 python tools/ae_bench.py > $o/r05_ae.log 2>&1
timeout 600 python tools/full_run.py 100000 200 ITC > $o/r05_full_run.log 2>&1
timeout 600 bash tools/gpu_idle.sh > $o/r05_gpu_idle.log 2>&1
timeout 300 python tools/hbm_map.py --n 120 > $o/r05_hbm_map_box2.log 2>&1
for u in 1 2 3; do timeout 300 python bench.py --config c5 --steps 100 --windows 4 --no-cpu-baseline --no-variants > $o/r05_bench_c5_placed_$u.json.log 2>/dev/null; done
fi
if [ $part = all ] || [ $part = tests ]; then
( echo "python -m pytest tests -q -m gpu   (final tree of round 5, fresh MI355X box)"; timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -4 ) > $o/r05_pytest_gpu.log 2>&1
( echo "round-5 record sweeps on the final tree: python tools/fuzz_step.py 3000 5; python tools/fuzz_aux.py 1500 5; python tools/fuzz_model.py 60 5; python tools/fuzz_oc.py 60 5; python tools/fuzz_sharded.py 10 5"
  timeout 900 python tools/fuzz_step.py 3000 5 2>&1 | tail -3
  timeout 900 python tools/fuzz_aux.py 1500 5 2>&1 | tail -6
  timeout 900 python tools/fuzz_model.py 60 5 2>&1 | tail -3
  timeout 900 python tools/fuzz_oc.py 60 5 2>&1 | tail -3
  timeout 900 python tools/fuzz_sharded.py 10 5 2>&1 | tail -3 ) > $o/r05_fuzz.log 2>&1
fi
ls -la $o | grep r05_ | head -80
