"""Reduce one rocprofv3 --pmc pass (counter_collection.csv, one row per dispatch and counter) to a small JSON:
for the kernels matching `pattern`, the counter values in dispatch order.  Run ON the GPU box before the CSVs are
deleted (a pass over a 30-kernel process writes tens of MB; gpurun merges back at most 64 MiB).
Usage: python tools/pmc_reduce.py <pass dir> <kernel substring> <out.json>"""

import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(d, pattern, out):
    hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    res = {"source": d, "kernel": pattern, "dispatches": []}
    if hits:
        per = defaultdict(dict)
        order = []
        with open(hits[0]) as f:
            for r in csv.DictReader(f):
                if pattern not in (r.get("Kernel_Name") or ""):
                    continue
                did = int(r.get("Dispatch_Id") or r.get("Correlation_Id") or 0)
                if did not in per:
                    order.append(did)
                per[did][r["Counter_Name"]] = per[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                per[did]["_grid"] = r.get("Grid_Size")
        res["dispatches"] = [dict(per[d_], _id=d_) for d_ in sorted(order)]
    with open(out, "w") as f:
        json.dump(res, f)
    print(f"{out}: {len(res['dispatches'])} dispatches")


if __name__ == "__main__":
    main(*sys.argv[1:4])
