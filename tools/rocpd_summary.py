"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max
duration, like `--stats` CSV output.  Usage: python tools/rocpd_summary.py results.db [> profiles/x.md]"""

import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, k, tot, avg, mn, mx in rows:
        n = n if len(n) < 110 else n[:107] + "..."
        print(f"| `{n}` | {k} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
