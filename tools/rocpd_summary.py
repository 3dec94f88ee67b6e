#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) as the
`--stats` kernel table: calls, total / average / min / max duration per kernel, plus
any PMC counters collected.  Usage: rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | % | grid | wg | vgpr | sgpr | lds |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx, g, w, v, s, l in rows:
        lines.append(f"| {n[:70]} | {c} | {t/1e3:.1f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | "
                     f"{100*t/total:.1f} | {g} | {w} | {v} | {s} | {l} |")
    try:
        pmc = cur.execute(
            "select k.name, p.counter_name, count(*), avg(p.value), sum(p.value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error:
        try:
            cols = [r[1] for r in cur.execute("pragma table_info('pmc_events')")]
            pmc = [("(schema)", ",".join(cols), 0, 0, 0)]
        except sqlite3.Error:
            pmc = []
    if pmc:
        lines += ["", "| kernel | counter | dispatches | avg/dispatch | sum |", "|---|---|---|---|---|"]
        for n, cn, c, a, s in pmc:
            lines.append(f"| {n[:70]} | {cn} | {c} | {a:.1f} | {s:.1f} |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
