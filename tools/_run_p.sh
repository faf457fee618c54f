o=gpurun_out/r05p; mkdir -p $o
cp gpurun_out/r05o/hbm_map.log $o/ 2>/dev/null
timeout 600 python -m pytest tests/test_distributed_oc_gpu.py -x -q -m gpu -k "both_vectors or one_rank_equals" > $o/pytest.log 2>&1; tail -2 $o/pytest.log
for rep in 1 2 3; do
  for pl in 0 1; do
    MKE_PLACE=$pl timeout 600 python bench.py --config c5 --steps 100 --windows 4 --no-cpu-baseline --no-variants > $o/bench_c5_place${pl}_$rep.json.log 2>$o/err_${pl}_$rep.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05p/bench_c5_place*.json.log')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], 'ms/step', round(d['ms_per_step']*1e3,1), 'score us', round(r.get('launch_us', r.get('kernel_us', 0)),1) if isinstance(r,dict) else '', [k for k in r.keys()][:0])
    except Exception as e: print(f,'ERR',e)
PY
