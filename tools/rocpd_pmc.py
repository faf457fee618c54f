#!/usr/bin/env python3
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd .db (one `--pmc` pass)."""
import sqlite3
import sys


def summarise(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value), min(e.value), max(e.value) from {pe} e "
         f"join {ip} i on e.pmc_id=i.id join {kd} d on d.event_id=e.event_id join {ks} s on d.kernel_id=s.id "
         f"group by s.kernel_name, i.name order by 4 desc")
    out = ["| kernel | counter | dispatches | avg | min | max |", "|---|---|---|---|---|---|"]
    for r in c.execute(q):
        if r[0].startswith("_ZN3mke"):
            out.append(f"| `{r[0][:70]}` | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} |")
    return "\n".join(out)


if __name__ == "__main__":
    print(summarise(sys.argv[1]))
