#!/bin/bash
# Timing-only ablations of the evaluator's sweep (results are wrong in variants 1-3): which part of the tile loop costs what
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp multike_amd/libmultike_hip.so /tmp/new.so
mkdir -p tools/ab; cp /tmp/new.so tools/ab/libbase.so
for v in 1 2 3; do     # scratch builds of the evaluator with parts of the tile loop compiled out (mke_simtile.h, SIMT_ABLATE)
  [ -f tools/ab/libab$v.so ] && continue
  (cd multike_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-math-errno -DSIMT_ABLATE=$v -c mke_eval.hip -o /tmp/eval_ab$v.o \
     && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v mke_eval.o) /tmp/eval_ab$v.o -o ../../tools/ab/libab$v.so) &
done; wait
for v in base ab1 ab2 ab3; do
  cp tools/ab/lib$v.so multike_amd/libmultike_hip.so
  rm -rf gpurun_out/abl
  rocprofv3 --kernel-trace --stats -d gpurun_out/abl -o ev -- python tools/eval_bench.py 60000 75 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d gpurun_out/abl -o ev256 -- python tools/eval_bench.py 30000 256 > /dev/null 2>&1
  for db in $(find gpurun_out/abl -name "*.db" | sort); do python tools/rocpd_summary.py $db 8 | grep "k_align_rank" | cut -c1-140 | sed "s/^/$v /"; done
done
rm -rf gpurun_out/abl
cp /tmp/new.so multike_amd/libmultike_hip.so
