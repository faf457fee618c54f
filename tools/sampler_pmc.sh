#!/bin/bash
# SQ counters of the epoch sampler (k_neg_sample) over a short bench.py run; run on the GPU box via gpurun.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass -d $root/gpurun_out/smp_$tag -o p -- python $root/bench.py --steps 190 --warmup 5 --no-cpu-baseline --no-variants > $root/gpurun_out/smp_$tag.log 2>&1
done
cd $root
python - <<'PY'
import glob, sqlite3
for d in sorted(glob.glob("gpurun_out/smp_*")):
    dbs = sorted(glob.glob(d+"/**/*.db", recursive=True))
    if not dbs: continue
    c = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, d.grid_size_x, count(*), avg(e.value), avg(d.end-d.start) from {pe} e join {ip} i on e.pmc_id=i.id "
         f"join {kd} d on d.event_id=e.event_id join {ks} s on d.kernel_id=s.id group by 1,2,3 order by 1,3,2")
    for r in c.execute(q):
        if "neg_sample" in r[0]:
            print(f"{r[0][8:36]} grid {r[2]} | {r[1]} | n={r[3]} | {r[4]:.5g} | {r[5]/1e3:.1f} us")
PY
rm -rf gpurun_out/smp_*/
