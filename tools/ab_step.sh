#!/bin/bash
# A/B of the attribute step on ONE box: tools/ab/libbase.so (a build of the committed tree, made by the caller) against the working
# tree's library, alternating, un-profiled step time (tools/attr_prof.py) and rocprofv3 kernel times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ATTR_LIBRARY=0
cp multike_amd/libmultike_hip.so /tmp/new.so
for rep in 1 2 3; do
  for v in base new; do
    [ $v = base ] && cp tools/ab/libbase.so multike_amd/libmultike_hip.so || cp /tmp/new.so multike_amd/libmultike_hip.so
    echo "$v: $(python tools/attr_prof.py 600 2>&1 | grep 'us/step' | sed 's/.*: //')"
  done
done
for v in base new; do
  [ $v = base ] && cp tools/ab/libbase.so multike_amd/libmultike_hip.so || cp /tmp/new.so multike_amd/libmultike_hip.so
  rm -rf gpurun_out/abs; rocprofv3 --kernel-trace --stats -d gpurun_out/abs -o t -- python tools/attr_prof.py 300 > /dev/null 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/abs -name "*.db" | head -1) 7 | grep "mke" | cut -c1-140 | sed "s/^/$v /"
done
rm -rf gpurun_out/abs; cp /tmp/new.so multike_amd/libmultike_hip.so
