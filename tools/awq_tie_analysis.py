"""Gram-vs-reference-structured loss statistics from `tools/awq_bench.py --search auto --tie-margin inf --dump F`
(every candidate scored by BOTH engines): how far the Gram score may be trusted when two candidates are close.
Prints the markdown that profiles/r02_awq_tie_margin.md holds.   Usage: python tools/awq_tie_analysis.py F [F2 ...]"""

import json
import sys

import numpy as np


def main():
    for path in sys.argv[1:]:
        d = json.load(open(path))
        lin = d["linears"]
        flips, absrel, spread, need = 0, [], [], []
        for l in lin:
            g, e = np.array(l["gram_loss"]), np.array(l["loss"])
            dd = e - g
            absrel.append(np.max(np.abs(dd) / e))
            spread.append((dd.max() - dd.min()) / e.min())
            flips += int(np.argmin(g) != np.argmin(e))
            need.append(g[int(np.argmin(e))] / g.min() - 1)
        print(f"### {path}: {len(lin)} linears x {len(lin[0]['loss'])} candidates\n")
        print(f"* plain Gram argmin != error-GEMM argmin on **{flips}** linears; smallest margin that would have "
              f"re-scored the true minimum: {max(need):.2e}")
        print(f"* max |loss_gemm - loss_gram| / loss_gemm over all candidates: {max(absrel):.2e}")
        print(f"* max spread of (loss_gemm - loss_gram) over ALL candidate pairs of a linear, relative to its best loss: "
              f"{max(spread):.2e} (median {np.median(spread):.2e})\n")
        print("| margin | linears with a near-tie | candidates re-scored | worst pair spread among the re-scored |")
        print("|---|---|---|---|")
        for m in (1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 3e-2):
            cnt = tot = 0
            worst = 0.0
            for l in lin:
                g, e = np.array(l["gram_loss"]), np.array(l["loss"])
                c = np.where(g <= g.min() * (1 + m))[0]
                if len(c) > 1:
                    cnt += 1
                    tot += len(c)
                    dd = (e - g)[c]
                    worst = max(worst, (dd.max() - dd.min()) / e.min())
            print(f"| {m:g} | {cnt} | {tot} | {worst:.2e} |")
        print()
        print("| score gap of a candidate pair (relative to the best score) | pairs | max spread of d / gap |")
        print("|---|---|---|")
        rows, overturned = [], 0.0
        for l in lin:
            g, e = np.array(l["gram_loss"]), np.array(l["loss"])
            dd = e - g
            for i in range(len(g)):
                for j in range(i + 1, len(g)):
                    gap, spread = abs(g[i] - g[j]) / g.min(), abs(dd[i] - dd[j]) / g.min()
                    if gap > 0:
                        rows.append((gap, spread / gap))
                    if spread >= gap:
                        overturned = max(overturned, gap)
        rows = np.array(rows)
        for lo, hi in ((0, 1e-3), (1e-3, 5e-3), (5e-3, 2e-2), (2e-2, 1e-1), (1e-1, 10)):
            sel = rows[(rows[:, 0] >= lo) & (rows[:, 0] < hi)]
            if len(sel):
                print(f"| [{lo:g}, {hi:g}) | {len(sel)} | {sel[:, 1].max():.3f} |")
        print(f"\nlargest score gap a rounding spread could overturn (spread >= gap): {overturned:.2e}\n")


if __name__ == "__main__":
    main()
