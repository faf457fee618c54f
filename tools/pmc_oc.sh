#!/bin/bash
# usage (GPU box, via gpurun): tools/pmc_oc.sh <c2|c5>   -> gpurun_out/r05_pmc_oc_g8_<cfg>.md
# Memory-side bytes of what rank 0 of 8 launches per global step (tools/oc_rank_compute.py): two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; kernel trace only).  FETCH_SIZE is corrected as tools/pmc_to_json.py does it: calibrated on the update
# launch of the same run, whose read byte count is known (rows visited x 3 row reads).
cfg=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmc_oc_${cfg}_$c
  rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/pmc_oc_${cfg}_$c -o p -- \
    python $root/tools/oc_rank_compute.py --world 8 --config $cfg --steps 30 > $root/gpurun_out/pmc_oc_${cfg}_$c.log 2>&1
done
cd $root
python - $cfg <<'PY'
import glob, json, os, sqlite3, sys
cfg = sys.argv[1]
def per_kernel(dbdir):
    db = sorted(glob.glob(os.path.join(dbdir, "**", "*.db"), recursive=True))[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value), min(e.value), max(e.value), avg(d.end - d.start) from {pe} e "
         f"join {ip} i on e.pmc_id=i.id join {kd} d on d.event_id=e.event_id join {ks} s on d.kernel_id=s.id "
         f"group by s.kernel_name, i.name order by 4 desc")
    return [r for r in c.execute(q) if r[0].startswith("_ZN3mke")]
f = per_kernel(f"gpurun_out/pmc_oc_{cfg}_FETCH_SIZE"); w = per_kernel(f"gpurun_out/pmc_oc_{cfg}_WRITE_SIZE")
line = json.loads([l for l in open(f"gpurun_out/pmc_oc_{cfg}_FETCH_SIZE.log") if l.startswith('{"tool"')][-1])
stride = 80 if cfg == "c2" else 256
pick = lambda rows, needle: max([x for x in rows if needle in x[0]], key=lambda x: x[2])
uw = pick(w, "k_rows_update_multi"); uf = pick(f, "k_rows_update_multi")
rows_upd = uw[3] * 1024 / (3 * stride * 4)            # the update launch writes 3 rows per visited row: WRITE_SIZE needs no correction
corr = rows_upd * 3 * stride * 4 / (uf[3] * 1024)
md = [f"# r05 — memory-side bytes of rank 0 of 8's launches per global step, {cfg.upper()} (`tools/pmc_oc.sh {cfg}`)", "",
      "Two separate passes, each `rocprofv3 --kernel-trace --pmc <COUNTER>` over `tools/oc_rank_compute.py --world 8`; per dispatch; durations are the",
      "PROFILED ones (counter collection lengthens a launch).", "",
      f"Read correction {corr:.3f}: the update launch visits {rows_upd:,.0f} rows per step by its WRITE_SIZE (3 rows written per visited row, uncorrected), i.e. reads "
      f"{rows_upd * 3 * stride * 4 / 1e6:.1f} MB, against FETCH_SIZE {uf[3] * 1024 / 1e6:.1f} MB.", "",
      "| kernel | dispatches | read MB (corrected) | written MB | total MB | profiled us | TB/s at the profiled duration |", "|---|---|---|---|---|---|---|"]
for needle in ("k_oc_bases", "k_oc_score", "k_oc_apply", "k_rows_update_multi"):
    try:
        a, b = pick(f, needle), pick(w, needle)
    except ValueError:
        continue
    rd, wr, us = a[3] * 1024 * corr / 1e6, b[3] * 1024 / 1e6, a[6] / 1e3
    md.append(f"| `{a[0][:60]}` | {a[2]} | {rd:.1f} | {wr:.1f} | {rd + wr:.1f} | {us:.1f} | {(rd + wr) / us:.2f} |")
md += ["", f"unprofiled phase times of the same tool: {json.dumps({k: round(v, 1) for k, v in line['phase_us'].items()})}"]
open(f"gpurun_out/r05_pmc_oc_g8_{cfg}.md", "w").write("\n".join(md) + "\n")
print("\n".join(md))
PY
rm -rf gpurun_out/pmc_oc_${cfg}_FETCH_SIZE gpurun_out/pmc_oc_${cfg}_WRITE_SIZE
