#!/usr/bin/env python3
"""rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a bench.py run -> r03_pmc_<config>.json / .md.

Reading rules (MI355X_MICROARCH.md §HBM): unit KB (x 1024 = bytes); on gfx950 FETCH_SIZE under-reports reads (exactly 1/2
for 16 B/lane streams, "other widths uncalibrated: calibrate on a known byte count in your own access pattern"), WRITE_SIZE
needs no correction.  Calibration: k_rows_update_multi, whose read byte count is known from the bench line (rows it
visited x 3 row reads + the flag scan + the next step's id streams it counts)."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("MKE_ROUND", "r04")
sys.path.insert(0, ROOT)


def per_kernel(dbdir):
    db = sorted(glob.glob(os.path.join(dbdir, "**", "*.db"), recursive=True))[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value), min(e.value), max(e.value) from {pe} e "
         f"join {ip} i on e.pmc_id=i.id join {kd} d on d.event_id=e.event_id join {ks} s on d.kernel_id=s.id "
         f"group by s.kernel_name, i.name order by 4 desc")
    return [r for r in c.execute(q) if r[0].startswith("_ZN3mke")]


def pick(rows, needle):
    r = [x for x in rows if needle in x[0]]
    return max(r, key=lambda x: x[2]) if r else None


def main():
    cfg, d_fetch, d_write, log = sys.argv[1:5]
    import bench
    line = [l for l in open(log) if l.startswith('{"metric"')][-1]
    b = json.loads(line)
    fetch, write = per_kernel(d_fetch), per_kernel(d_write)
    sf, sw = pick(fetch, "k_triple_score"), pick(write, "k_triple_score")
    uf, uw = pick(fetch, "k_rows_update_multi"), pick(write, "k_rows_update_multi")
    c = b["config"]
    stride = (c["dim"] + 15) // 16 * 16
    rows = b["roofline"]["update_kernel"]["touched_rows_last_step"]
    ids = c["batch"] * (2 + 2 * c["neg"]) * 4
    known_read = rows * 3 * stride * 4 + c["n_ent"] * 4 + c["n_rel"] * 4 + ids
    corr = known_read / (uf[3] * 1024)
    traffic = sf[3] * 1024 * corr + sw[3] * 1024
    out = {"kernel": "k_triple_score", "config": cfg, "workload": c["workload"],
           "kernel_source_sha": bench.kernel_source_hash(),
           "fetch_size_kb": sf[3], "write_size_kb": sw[3], "fetch_correction": corr,
           "fetch_correction_basis": f"k_rows_update_multi: {rows} rows x 3 x {stride * 4} B + flag scan + id streams = "
                                     f"{known_read} B known vs FETCH_SIZE {uf[3]:.1f} KB",
           "traffic_bytes_per_launch": int(traffic), "dispatches": sf[2],
           "update_kernel": {"fetch_size_kb": uf[3], "write_size_kb": uw[3],
                             "known_write_bytes": rows * 3 * stride * 4}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{RND}_pmc_{cfg}.json"), "w") as f:
        json.dump(out, f, indent=1)
    md = [f"# {RND} — PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --config {cfg} --steps 40 --warmup 5`", "",
          f"workload: {c['workload']}; kernel sources sha {out['kernel_source_sha']}", "",
          "Two separate passes, each `rocprofv3 --kernel-trace --pmc <COUNTER>`; per dispatch, unit KB.", "",
          "| kernel | counter | dispatches | avg | min | max |", "|---|---|---|---|---|---|"]
    for r in fetch + write:
        md.append(f"| `{r[0][:72]}` | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} |")
    alg = b["roofline"]["alg_bytes_per_triple"] * b["roofline"]["triples_per_launch"]
    md += ["", f"* read correction {corr:.3f} ({out['fetch_correction_basis']}); WRITE_SIZE of the same kernel "
               f"{uw[3] * 1024 / 1e6:.1f} MB vs {rows * 3 * stride * 4 / 1e6:.1f} MB known",
           f"* `k_triple_score`: {sf[3] * 1024 * corr / 1e6:.1f} MB read + {sw[3] * 1024 / 1e6:.1f} MB written = "
           f"**{traffic / 1e6:.1f} MB per launch** against {alg / 1e6:.1f} MB algorithmic "
           f"({traffic / alg:.2f}x); bench.py divides it by the UNPROFILED launch duration (`roofline.achieved_counter`)"]
    with open(os.path.join(ROOT, "gpurun_out", f"{RND}_pmc_{cfg}.md"), "w") as f:
        f.write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
