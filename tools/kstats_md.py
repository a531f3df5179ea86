"""rocprofv3 --kernel-trace --stats CSV -> markdown table of the moq:: kernels (avg / min / max duration).
Usage: python tools/kstats_md.py <dir containing *kernel_stats.csv> [> profiles/x.md]"""

import csv
import glob
import os
import sys


def main(d):
    hits = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not hits:
        print(f"no kernel_stats.csv under {d}")
        return
    print("| kernel | calls | avg us | min us | max us |")
    print("|---|---|---|---|---|")
    with open(hits[0]) as f:
        rows = [r for r in csv.DictReader(f) if "moq" in (r.get("Name") or "")]
    for r in sorted(rows, key=lambda r: r["Name"]):
        n = r["Name"].replace("void ", "").replace("moq::", "")
        n = n[: n.index("(")] if "(" in n else n
        print(f"| `{n}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
