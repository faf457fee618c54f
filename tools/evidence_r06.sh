#!/bin/bash
# Round-6 evidence batch (GPU box, via gpurun): everything under gpurun_out/r06_*; copy what is to be judged into profiles/.
# Order matters: the PMC passes come first and their JSON is copied into profiles/ ON THE BOX, so that every bench line below
# quotes the byte counts of this very build (bench.py keys them by a hash of the kernel sources) — the headline `roofline.frac`
# is the counter-based one when that file is there.
# usage: tools/evidence_r06.sh [part]   (part: all | core | sharded | rest | tests)
part=${1:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
export MKE_ROUND=r06
o=gpurun_out
if [ $part = all ] || [ $part = core ]; then
timeout 900 tools/pmc_passes.sh c2 > $o/r06_pmc_c2.log 2>&1
timeout 1200 tools/pmc_passes.sh c5 --steps 40 > $o/r06_pmc_c5.log 2>&1
cp $o/r06_pmc_c2.json $o/r06_pmc_c5.json profiles/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $o/r06_bench_c2_20steps.json.log 2>&1   # the driver's invocation (C5 + Zipf variants inside)
timeout 900 python bench.py > $o/r06_bench_c2.json.log 2>&1
timeout 900 tools/prof.sh r06_trace_c2 10 bench.py --no-cpu-baseline --no-variants --windows 2 > $o/r06_kernel_trace_c2.md 2>&1
timeout 900 python bench.py --config c5 --steps 100 --no-cpu-baseline > $o/r06_bench_c5.json.log 2>&1
timeout 900 tools/prof.sh r06_trace_c5 8 bench.py --config c5 --steps 100 --windows 6 --no-cpu-baseline > $o/r06_kernel_trace_c5.md 2>&1
timeout 600 python bench.py --zipf 1.0 --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $o/r06_bench_c2_zipf.json.log 2>&1
fi
if [ $part = all ] || [ $part = sharded ]; then
# the sharded step on one GPU: (i) one rank without collectives — entity-major (default) and the atomics form, (ii) the G > 1 path forced
# over a one-rank RCCL group: the native loop (mke_oc_steps calling RCCL's entry points) and the Python loop
timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/r06_bench_c2_sharded_g1.json.log 2>&1
MKE_OC_EM=0 timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/r06_bench_c2_sharded_g1_atomics.json.log 2>&1
MKE_OC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/r06_bench_c2_sharded_g1_rccl.json.log 2>&1
MKE_OC_NATIVE=0 MKE_OC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/r06_bench_c2_sharded_g1_rccl_pyloop.json.log 2>&1
timeout 900 python bench.py --config c5 --force-sharded --steps 100 --windows 5 --no-cpu-baseline > $o/r06_bench_c5_sharded_g1.json.log 2>&1
MKE_OC_FORCE_COLLECTIVES=1 timeout 900 python bench.py --config c5 --force-sharded --steps 100 --windows 5 --no-cpu-baseline > $o/r06_bench_c5_sharded_g1_rccl.json.log 2>&1
# rank 0 of 8: per-kernel tables (rocprofv3) of the entity-major step and of the atomics form, then the schedule with the modelled wire
timeout 400 tools/prof.sh r06_oc_g8_c2_trace 40 tools/oc_rank_compute.py --world 8 --config c2 --native 0 > $o/r06_oc_g8_c2_trace.md 2>&1
timeout 800 tools/prof.sh r06_oc_g8_c5_trace 40 tools/oc_rank_compute.py --world 8 --config c5 --native 0 > $o/r06_oc_g8_c5_trace.md 2>&1
timeout 400 tools/prof.sh r06_oc_g8_c2_atomics_trace 40 tools/oc_rank_compute.py --world 8 --config c2 --native 0 --em 0 > $o/r06_oc_g8_c2_atomics_trace.md 2>&1
timeout 800 tools/prof.sh r06_oc_g8_c5_atomics_trace 40 tools/oc_rank_compute.py --world 8 --config c5 --native 0 --em 0 > $o/r06_oc_g8_c5_atomics_trace.md 2>&1
for cfg in c2 c5; do
  for em in 1 0; do
    timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --native 0 --em $em 2>/dev/null | tail -1 > $o/r06_oc_${cfg}_em${em}_kernels.json
  done
  timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --chunks 1 --steps 230 --prefetch 2>/dev/null | tail -1 > $o/r06_oc_${cfg}_nowire.json
  for ch in 1 2 3; do
    timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --chunks $ch --steps 230 --wire-gbps 376 --latency-us 15 --prefetch 2>/dev/null | tail -1 > $o/r06_oc_${cfg}_ch${ch}_wire.json
  done
  timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --chunks 1 --steps 230 --wire-gbps 376 --latency-us 15 --prefetch --em 0 2>/dev/null | tail -1 > $o/r06_oc_${cfg}_ch1_wire_atomics.json
done
for z in "--zipf 1.0" "--rel-zipf 1.0"; do for em in 1 0; do
  timeout 600 python tools/oc_rank_compute.py --world 8 --config c2 --native 0 --steps 40 --em $em $z 2>/dev/null | tail -1 >> $o/r06_oc_c2_zipf_kernels.jsonl
done; done
timeout 600 python tools/oc_rank_compute.py --world 1 --config c2 --steps 100 --native 0 2>/dev/null | tail -1 > $o/r06_oc_c2_world1_em.json
MKE_BENCH_COMM=staged timeout 900 python bench.py --gpus 8 --steps 6 --warmup 2 > $o/r06_bench_gpus8_staged.json.log 2>&1
fi
if [ $part = all ] || [ $part = rest ]; then
ATTR_LIBRARY=0 timeout 300 tools/prof.sh r06_attr_trace 8 tools/attr_prof.py 400 > $o/r06_attr_trace.md 2>&1
timeout 300 python tools/attr_prof.py 400 > $o/r06_attr.log 2>&1
timeout 300 python tools/knn_bench.py > $o/r06_knn.log 2>&1
timeout 300 python tools/ae_bench.py > $o/r06_ae.log 2>&1
timeout 600 python tools/full_run.py 100000 200 ITC > $o/r06_full_run.log 2>&1
fi
if [ $part = all ] || [ $part = tests ]; then
( echo "python -m pytest tests -q -m gpu   (final tree of round 6, fresh MI355X box)"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -4 ) > $o/r06_pytest_gpu.log 2>&1
( echo "round-6 record sweeps on the final tree: python tools/fuzz_step.py 1500 6; python tools/fuzz_aux.py 800 6; python tools/fuzz_model.py 30 6; python tools/fuzz_oc.py 60 6; python tools/fuzz_sharded.py 8 6"
  timeout 900 python tools/fuzz_step.py 1500 6 2>&1 | tail -3
  timeout 900 python tools/fuzz_aux.py 800 6 2>&1 | tail -6
  timeout 900 python tools/fuzz_model.py 30 6 2>&1 | tail -3
  timeout 2400 python tools/fuzz_oc.py 60 6 2>&1 | tail -3
  timeout 900 python tools/fuzz_sharded.py 8 6 2>&1 | tail -3 ) > $o/r06_fuzz.log 2>&1
fi
ls -la $o | grep r06_ | head -100
