#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_passes.sh <config: c2|c5> [bench args...]
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only, no other trace domain) over a short
# bench.py run, then tools/pmc_to_json.py -> gpurun_out/r04_pmc_<config>.{json,md} (copy into profiles/ to commit).
cfg=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmc_${cfg}_$c
  rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/pmc_${cfg}_$c -o p -- \
    python $root/bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-variants "$@" \
    > $root/gpurun_out/pmc_${cfg}_$c.log 2>&1
done
cd $root
python tools/pmc_to_json.py $cfg gpurun_out/pmc_${cfg}_FETCH_SIZE gpurun_out/pmc_${cfg}_WRITE_SIZE gpurun_out/pmc_${cfg}_FETCH_SIZE.log
rm -rf gpurun_out/pmc_${cfg}_FETCH_SIZE gpurun_out/pmc_${cfg}_WRITE_SIZE
