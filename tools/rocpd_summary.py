#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel trace) into a per-kernel stats table (markdown)."""
import sqlite3
import sys


def summarise(db, top=12, by_grid=False):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    name = "s.kernel_name || ' grid ' || d.grid_size_x || 'x' || d.grid_size_y || 'x' || d.grid_size_z" if by_grid else "s.kernel_name"
    q = (f"select {name}, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"sum(d.end-d.start), max(s.arch_vgpr_count), max(s.sgpr_count) from {kd} d join {ks} s on d.kernel_id=s.id "
         f"group by 1 order by 6 desc")
    rows = list(c.execute(q))
    tot = sum(r[5] for r in rows)
    out = ["| kernel | calls | avg us | min us | max us | % GPU time | vgpr | sgpr |", "|---|---|---|---|---|---|---|---|"]
    for r in rows[:top]:
        out.append(f"| `{r[0][:80] if not by_grid else r[0][:60] + r[0][r[0].rindex(' grid '):]}` | {r[1]} | {r[2] / 1e3:.2f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | "
                   f"{100 * r[5] / tot:.1f} | {r[6]} | {r[7]} |")
    span = list(c.execute(f"select min(start), max(end) from {kd}"))[0]
    out.append("")
    out.append(f"total kernel time {tot / 1e6:.2f} ms over a {(span[1] - span[0]) / 1e6:.2f} ms span, {sum(r[1] for r in rows)} dispatches")
    return "\n".join(out)


def dump_dispatches(db, substr, csv_path):
    """Per-dispatch rows (start ns, end ns, duration ns, grid) of every kernel whose name contains `substr`: small enough to
    commit under profiles/, so that a summary's average can be recomputed by a reader."""
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.end - d.start, d.grid_size_x from {kd} d join {ks} s "
                          f"on d.kernel_id = s.id where s.kernel_name like ? order by d.start", (f"%{substr}%",)))
    with open(csv_path, "w") as f:
        f.write("kernel,start_ns,end_ns,duration_ns,grid_x\n")
        for r in rows:
            f.write(f"\"{r[0][:70]}\",{r[1]},{r[2]},{r[3]},{r[4]}\n")
    return len(rows)


def gap_table(db, sampler_ordinal=3, steps=20):
    """Where a native-loop step's time sits, from the dispatch timestamps of `bench.py --steps 20 --warmup 5` (the driver's
    line): the timed region = the `sampler_ordinal`-th k_neg_sample launch (1: pre-warm epoch, 2: --warmup steps, 3: timed
    steps), the first step's stand-alone k_count_refs, then `steps` x (k_triple_score, k_rows_update_multi).  Per dispatch:
    duration and the idle gap in front of it (start - previous end); sums over the region."""
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    short = lambda n: ("k_" + n.split("k_", 1)[1].split("I", 1)[0].split("E", 1)[0]) if "k_" in n else n[:40]
    samp = [i for i, r in enumerate(rows) if "k_neg_sample" in r[0]]
    if len(samp) < sampler_ordinal:
        return f"only {len(samp)} sampler launches in the trace"
    i0 = samp[sampler_ordinal - 1]
    # the region ends with the steps-th k_rows_update_multi after i0
    n_upd, i1 = 0, i0
    for i in range(i0, len(rows)):
        if "k_rows_update_multi" in rows[i][0]:
            n_upd += 1
            if n_upd == steps:
                i1 = i
                break
    reg = rows[i0:i1 + 1]
    out = ["| # | kernel | duration us | gap in front us |", "|---|---|---|---|"]
    by = {}
    prev_end = None
    for k, (n, st, en) in enumerate(reg):
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        out.append(f"| {k} | `{short(n)}` | {(en - st) / 1e3:.2f} | {gap:.2f} |")
        a = by.setdefault(short(n), [0, 0.0, 0.0])
        a[0] += 1; a[1] += (en - st) / 1e3; a[2] += gap
        prev_end = en
    wall = (reg[-1][2] - reg[0][1]) / 1e3
    first_score = next(k for k, r in enumerate(reg) if "k_triple_score" in r[0])
    steady = (reg[-1][2] - reg[first_score][1]) / 1e3
    out += ["", "| kernel | launches | sum of durations us | sum of gaps in front us | per step us (duration + gap) |", "|---|---|---|---|---|"]
    for n, (cnt, du, ga) in by.items():
        out.append(f"| `{n}` | {cnt} | {du:.1f} | {ga:.1f} | {(du + ga) / steps:.2f} |")
    out += ["", f"region: {len(reg)} dispatches, {wall:.1f} us from the sampler's start to the last update's end = {wall / steps:.2f} us per step; "
                f"from the first score launch on: {steady:.1f} us = {steady / steps:.2f} us per step",
            f"kernel time {sum(v[1] for v in by.values()):.1f} us + idle gaps {sum(v[2] for v in by.values()):.1f} us (the profiler's per-dispatch "
            f"overhead sits in the gaps: they are an upper bound of what the unprofiled run idles)"]
    return "\n".join(out)


def pmc_table(db, needle="mke"):
    """Per (kernel, counter) averages of a `rocprofv3 --kernel-trace --pmc ...` run (kernels whose name contains `needle`)."""
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value) from {pe} e join {ip} i on e.pmc_id=i.id "
         f"join {kd} d on d.event_id=e.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, i.name")
    by = {}
    for name, ctr, n, avg in c.execute(q):
        if needle in name:
            by.setdefault(name, {})[ctr] = (n, avg)
    ctrs = sorted({k for v in by.values() for k in v})
    out = ["| kernel | dispatches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
    for name, v in sorted(by.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
        n = max(x[0] for x in v.values())
        out.append(f"| `{name[:70]}` | {n} | " + " | ".join(f"{v[k][1]:.3g}" if k in v else "" for k in ctrs) + " |")
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--pmc":           # rocpd_summary.py <db> --pmc [kernel-name substring]
        print(pmc_table(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 else "mke"))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "--gaps":          # rocpd_summary.py <db> --gaps [sampler ordinal] [steps]
        print(gap_table(sys.argv[1], int(sys.argv[3]) if len(sys.argv) > 3 else 3, int(sys.argv[4]) if len(sys.argv) > 4 else 20))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "--dump":          # rocpd_summary.py <db> --dump <kernel substring> <csv>
        print(dump_dispatches(sys.argv[1], sys.argv[3], sys.argv[4]), "dispatches written to", sys.argv[4])
        sys.exit(0)
    print(summarise(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12, by_grid=len(sys.argv) > 3))
