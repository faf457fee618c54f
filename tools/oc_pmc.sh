#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -c1-10 | tr ' ' '_')
  OC_CFG=c5 rocprofv3 --kernel-trace --pmc $pass -d $root/gpurun_out/ocp_$tag -o p -- python $root/tools/oc_bench.py > $root/gpurun_out/ocp_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc $pass -d $root/gpurun_out/fup_$tag -o p -- python $root/bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline > $root/gpurun_out/fup_$tag.log 2>&1
done
cd $root
python - <<'PY'
import glob, sqlite3
for d in sorted(glob.glob("gpurun_out/ocp_*")+glob.glob("gpurun_out/fup_*")):
    if not glob.glob(d+"/**/*.db", recursive=True): continue
    db = sorted(glob.glob(d+"/**/*.db", recursive=True))[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value) from {pe} e join {ip} i on e.pmc_id=i.id "
         f"join {kd} d on d.event_id=e.event_id join {ks} s on d.kernel_id=s.id group by 1,2 order by 1,2")
    for r in c.execute(q):
        if "oc_score" in r[0] or "triple_score" in r[0]:
            print(f"{d[11:]} | {r[0][8:40]} | {r[1]} | {r[2]} | {r[3]:.4g}")
PY
rm -rf gpurun_out/ocp_* gpurun_out/fup_*
