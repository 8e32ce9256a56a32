#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace): per-kernel calls / total / avg / min / max.

  python tools/prof_summary.py gpurun_out/prof/x_results.db [--csv out.csv] [--grid | --pmc | --gaps]
rocprofv3 in this image writes `*_results.db` by default; `--output-format csv` gives kernel_stats.csv directly.
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:90]


def pmc_summary(c, out_csv):
    """per-kernel average of every collected counter (rocprofv3 --pmc run): `counters_collection` view"""
    rows = c.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, count(*), avg(value), "
                     "min(value), max(value) from counters_collection group by 1,2,3,4,5 order by 7 desc").fetchall()
    out = ["kernel,counter,dispatches,avg,min,max"]
    for r in rows:
        out.append(f'"{short(r[0])} grid=({r[1]},{r[2]},{r[3]})",{r[4]},{r[5]},{r[6]:.1f},{r[7]:.1f},{r[8]:.1f}')
    txt = "\n".join(out)
    if out_csv:
        open(out_csv, "w").write(txt + "\n")
    print(txt)


def gaps_summary(c, out_csv):
    """idle time between consecutive kernels (next.start - prev.end), grouped by the (previous, next) kernel pair:
    what a launch boundary costs inside a graph, between a graph and an eager launch, ..."""
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    pairs = {}
    for (n0, _, e0), (n1, s1, _) in zip(rows, rows[1:]):
        pairs.setdefault((short(n0)[:48], short(n1)[:48]), []).append((s1 - e0) / 1e3)
    out = ["previous,next,count,median_gap_us,avg_gap_us,total_gap_ms"]
    for (a, b), g in sorted(pairs.items(), key=lambda kv: -sum(kv[1])):
        g.sort()
        out.append(f'"{a}","{b}",{len(g)},{g[len(g) // 2]:.2f},{sum(g) / len(g):.2f},{sum(g) / 1e3:.3f}')
    txt = "\n".join(out)
    if out_csv:
        open(out_csv, "w").write(txt + "\n")
    print(txt)


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    if "--gaps" in sys.argv:
        return gaps_summary(c, sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None)
    if "--pmc" in sys.argv:
        return pmc_summary(c, sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    by_grid = "--grid" in sys.argv
    key = "name, grid_x, grid_y, grid_z" if by_grid and "grid_x" in cols else "name"
    rows = c.execute(f"select {key}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {key} order by 3 desc").fetchall()
    total = sum(r[-4] for r in rows)
    out = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct"]
    for r in rows:
        nm = short(r[0]) + ("" if key == "name" else f" grid=({r[1]},{r[2]},{r[3]})")
        n, tot, avg, mn, mx = r[-5:]
        out.append(f'"{nm}",{n},{tot / 1e6:.3f},{avg / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * tot / total:.1f}')
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    out.append(f'"TOTAL kernel time",,{total / 1e6:.3f},,,,100.0')
    out.append(f'"first-kernel-start to last-kernel-end",,{(span[1] - span[0]) / 1e6:.3f},,,,')
    txt = "\n".join(out)
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
