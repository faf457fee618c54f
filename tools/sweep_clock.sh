#!/bin/bash
# Effective shader clock of the evaluator's MFMA sweep: GRBM_GUI_ACTIVE (and SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES) per dispatch
# over the dispatch's duration, for the shipped kernel and the bare-MFMA ablation (tools/ab/libab3.so: no epilogue / staging /
# barrier), on random and on ZERO inputs (the DVFS give-back of MI355X_MICROARCH.md: zero data clocks higher).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp multike_amd/libmultike_hip.so /tmp/new.so
[ -f tools/ab/libab3.so ] || { echo "run tools/sweep_ablate.sh first (it builds tools/ab/libab3.so)"; exit 1; }
for v in new ab3; do
  [ $v = new ] && cp /tmp/new.so multike_amd/libmultike_hip.so || cp tools/ab/lib$v.so multike_amd/libmultike_hip.so
  for data in random zero; do
    rm -rf gpurun_out/clk
    MKE_EVAL_BENCH_ZERO=$([ $data = zero ] && echo 1 || echo 0) rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/clk -o p -- python tools/eval_bench.py 60000 75 > /dev/null 2>&1
    db=$(find gpurun_out/clk -name "*.db" | head -1)
    python - "$db" "$v" "$data" <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda p: [x for x in tabs if x.startswith(p)][0]
pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
rows = list(c.execute(f"select d.id, d.end - d.start, i.name, sum(e.value), count(*) from {pe} e join {ip} i on e.pmc_id=i.id join {kd} d on d.event_id=e.event_id "
                      f"join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_align_rank%' group by d.id, i.name order by d.id"))
by = {}
for did, dur, name, val, n in rows:
    by.setdefault(did, {"dur": dur})[name] = (val, n)
for did, r in list(by.items())[1:]:
    g = r.get("GRBM_GUI_ACTIVE", (0, 1)); b = r.get("SQ_BUSY_CYCLES", (0, 1)); m = r.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 1))
    print(f"{sys.argv[2]} {sys.argv[3]}: {r['dur'] / 1e3:.0f} us; GRBM_GUI_ACTIVE {g[0] / g[1]:.3e} per instance ({g[1]} inst) => {g[0] / g[1] / r['dur']:.2f} GHz; "
          f"SQ_BUSY_CYCLES {b[0] / b[1]:.3e} ({b[1]}) => {b[0] / b[1] / r['dur']:.2f} GHz; MFMA busy {m[0]:.3e} = {m[0] / 1024 / (g[0] / g[1]):.2f} of the clock-cycles per SIMD")
P
  done
done
rm -rf gpurun_out/clk
cp /tmp/new.so multike_amd/libmultike_hip.so
