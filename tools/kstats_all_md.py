"""rocprofv3 --kernel-trace --stats CSV -> markdown table of ALL kernels with their share of the total kernel time.
Usage: python tools/kstats_all_md.py <dir containing *kernel_stats.csv> [top N] [> profiles/x.md]"""

import csv
import glob
import os
import re
import sys


def main(d, top=16):
    hits = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not hits:
        print(f"no kernel_stats.csv under {d}")
        return
    with open(hits[0]) as f:
        rows = list(csv.DictReader(f))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print(f"Total kernel time {tot / 1e9:.3f} s over {calls} launches.\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        n = re.sub(r"\bvoid ", "", r["Name"])
        n = n if len(n) <= 120 else n[:117] + "..."
        t = float(r["TotalDurationNs"])
        print(f"| `{n}` | {r['Calls']} | {t / 1e6:.1f} | {t / int(r['Calls']) / 1e3:.1f} | {100 * t / tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16)
