#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace): per-kernel calls / total / avg / min / max.

  python tools/prof_summary.py gpurun_out/prof/x_results.db [--csv out.csv] [--grid]
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


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
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
