o=gpurun_out/r05l; mkdir -p $o
timeout 1200 python -m pytest tests/test_distributed_oc_gpu.py tests/test_rccl_gpu.py tests/test_distributed_gpu.py tests/test_distributed_model_gpu.py tests/test_distributed_run_gpu.py tests/test_abi.py -x -q -m gpu > $o/pytest.log 2>&1
tail -5 $o/pytest.log
for cfg in c2 c5; do
  timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --chunks 1 --prefetch 2>$o/oc_${cfg}_nowire.err | tail -1 > $o/oc_${cfg}_nowire.json
  for ch in 1 2; do
    timeout 600 python tools/oc_rank_compute.py --world 8 --config $cfg --chunks $ch --wire-gbps 376 --latency-us 15 --prefetch 2>/dev/null | tail -1 > $o/oc_${cfg}_ch${ch}_wire.json
  done
done
timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/bench_c2_sharded_g1.json.log 2>$o/bench_c2_sharded.err
MKE_OC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --force-sharded --steps 184 --windows 5 --no-cpu-baseline > $o/bench_c2_sharded_g1_rccl.json.log 2>>$o/bench_c2_sharded.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05l/oc_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['phase_us'], 'wall', round(d['wall_us_per_step_loopback'],1), 'host', round(d['host_us_per_step'],1), 'cap', d['capacity_vectors'])
    except Exception as e: print(f, 'ERR', e)
for f in sorted(glob.glob('gpurun_out/r05l/bench_*.json.log')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('host_us_per_step'))
    except Exception as e: print(f,'ERR',e)
PY
