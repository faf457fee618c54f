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


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--dump":          # rocpd_summary.py <db> --dump <kernel substring> <csv>
        print(dump_dispatches(sys.argv[1], sys.argv[3], sys.argv[4]), "dispatches written to", sys.argv[4])
        sys.exit(0)
    print(summarise(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12, by_grid=len(sys.argv) > 3))
