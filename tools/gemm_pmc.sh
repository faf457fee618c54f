#!/bin/bash
# PMC passes over tools/gemm_bench.py (the auto-encoder's GEMM shapes): matrix-pipe busy cycles, wait breakdown, LDS conflicts
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU -d $root/gpurun_out/gemm_pmc1 -o p -- python $root/tools/gemm_bench.py > $root/gpurun_out/gemm_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $root/gpurun_out/gemm_pmc2 -o p -- python $root/tools/gemm_bench.py > $root/gpurun_out/gemm_pmc2.log 2>&1
cd $root
python - <<'PY'
import glob, sqlite3
for d in ("gemm_pmc1", "gemm_pmc2"):
    db = sorted(glob.glob(f"gpurun_out/{d}/**/*.db", recursive=True))[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, d.grid_size_x, i.name, count(*), avg(e.value) from {pe} e join {ip} i on e.pmc_id=i.id "
         f"join {kd} d on d.event_id=e.event_id join {ks} s on d.kernel_id=s.id group by 1,2,3 order by 1,2,3")
    print("==", d)
    for r in c.execute(q):
        if "gemm" in r[0] or "Cijk" in r[0]:
            print(f"| `{r[0][:60]}` grid {r[1]} | {r[2]} | {r[3]} | {r[4]:.4g} |")
PY
rm -rf gpurun_out/gemm_pmc1 gpurun_out/gemm_pmc2
