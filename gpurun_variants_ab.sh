#!/bin/bash
mkdir -p gpurun_out/r04i
for cfg in c2 c5; do for q in 0 1; do
  python tools/oc_rank_compute.py --world 8 --config $cfg --set oc_score_quarter=$q > gpurun_out/r04i/oc_w8_${cfg}_q$q.log 2>&1
  grep "^{" gpurun_out/r04i/oc_w8_${cfg}_q$q.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg q=$q', {k:round(v,1) for k,v in d['phase_us'].items()}, round(d['wall_us_per_step_loopback'],1))"
done; done
for q in 0 1; do python tools/oc_rank_compute.py --world 4 --config c2 --set oc_score_quarter=$q 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 world 4 q=$q', {k:round(v,1) for k,v in d['phase_us'].items()}, round(d['wall_us_per_step_loopback'],1))"; done
