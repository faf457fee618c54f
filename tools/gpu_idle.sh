cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fr
rocprofv3 --kernel-trace -d gpurun_out/fr -o t -- python tools/full_run.py 100000 120 ITC > gpurun_out/fr.log 2>&1
tail -6 gpurun_out/fr.log | cut -c1-300
db=$(find gpurun_out/fr -name "*.db" | head -1)
python - "$db" <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
print(len(rows), "dispatches in the process")
first = next(i for i, r in enumerate(rows) if "k_neg_sample" in r[2])      # the drivers' run(): from the first epoch's sampler launch
rows = rows[first:]
print(len(rows), "dispatches from the first training epoch on")
# union of busy intervals (two streams overlap)
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
prev_n = rows[0][2][:40]
short = lambda x: x.replace("_ZN2at6native", "at::").replace("_ZN3mke", "mke::")[:34]
for k_, (s, e, n) in enumerate(rows[1:], 1):
    n = n[:40]
    if s > cur_e:
        after = " ".join(short(r[2]) for r in rows[k_ + 1:k_ + 5]) if s - cur_e > 1e7 else ""   # what the host enqueued next: names the call
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e, prev_n + "  ->  " + n + ("  | then " + after if after else ""))); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_n = n
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"span {span/1e9:.2f} s, GPU busy (union) {busy/1e9:.2f} s, idle {(span-busy)/1e9:.2f} s")
import collections
big = sorted(gaps, reverse=True)[:40]
print("largest gaps (ms) and the kernel that ended them:")
for g, at, n in big: print(f"  {g/1e6:8.2f} ms at +{(at-rows[0][0])/1e9:7.3f} s : {n}")
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<10us" if g < 1e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else "<10ms" if g < 1e7 else ">=10ms"] += g
print({k: round(v/1e9, 3) for k, v in hist.items()})
tot = collections.Counter(); cnt = collections.Counter()
for s_, e_, n_ in rows:
    key = n_.split("(")[0][:48]
    tot[key] += e_ - s_; cnt[key] += 1
print("kernel time by name (sum over both streams; the union above counts overlap once):")
for k_, v_ in tot.most_common(14):
    print(f"  {v_/1e9:6.3f} s  {cnt[k_]:7d} x {v_/cnt[k_]/1e3:7.1f} us  {k_}")
P
rm -rf gpurun_out/fr
