#!/bin/bash
# usage (GPU box): tools/c5_tlb.sh "<COUNTER ...>"   — tools/c5_variance.py under one rocprofv3 --pmc pass; prints, per dispatch
# ORDER (the six allocations follow each other), the counter sums and the duration of k_triple_score in blocks of 60 dispatches
root=${GRAFT_REPO_ROOT:-$(pwd)}
ctrs="$1"
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/tlbp
rocprofv3 --kernel-trace --pmc $ctrs -d $root/gpurun_out/tlbp -o p -- python $root/tools/c5_variance.py > $root/gpurun_out/tlbp.log 2>&1
cd $root
grep "allocation" gpurun_out/tlbp.log
python - <<'PY'
import glob, sqlite3
db = sorted(glob.glob("gpurun_out/tlbp/**/*.db", recursive=True))[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda p: [x for x in tabs if x.startswith(p)][0]
pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
rows = list(c.execute(f"select d.start, d.end - d.start, i.name, sum(e.value) from {pe} e join {ip} i on e.pmc_id=i.id join {kd} d on d.event_id=e.event_id "
                      f"join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_triple_score%' group by d.id, i.name order by d.start"))
names = sorted({r[2] for r in rows})
by = {}
for st, dur, nm, v in rows:
    by.setdefault(st, {"dur": dur})[nm] = v
ds = [by[k] for k in sorted(by)]
print("dispatches of k_triple_score:", len(ds), "counters:", names)
blk = 60
for a in range(0, len(ds), blk):
    seg = ds[a:a + blk]
    line = f"dispatches {a:4d}-{a + len(seg) - 1:4d}: avg {sum(x['dur'] for x in seg) / len(seg) / 1e3:7.1f} us"
    for nm in names:
        line += f"  {nm} {sum(x.get(nm, 0) for x in seg) / len(seg):12.0f}"
    print(line)
PY
rm -rf gpurun_out/tlbp
