#!/usr/bin/env python3
"""L2 (TCC) request counters per kernel function from one rocprofv3 PMC pass (TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum):
requests per launch, hit rate, and requests x 128 B / launch as an upper estimate of the bytes the XCDs' L2s served
(a wide load's request is a 128-byte line; stores and atomics count as requests too)."""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import base


def main():
    d, out = sys.argv[1:3]
    tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = base(row["Kernel_Name"])
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "TCC_REQ_sum": n[k] += 1
    res = {}
    for k, c in sorted(tot.items()):
        if n[k] < 2: continue
        hit, miss, req = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0), c.get("TCC_REQ_sum", 0.0)
        res[k] = {"launches": n[k], "req_per_launch": req / n[k], "hit_rate": hit / (hit + miss) if hit + miss else None,
                  "req_x128B_per_launch": req / n[k] * 128}
        print(k[:60].ljust(60), f"req/launch {req / n[k]:12.0f}  hit {res[k]['hit_rate']}  ~MB {req / n[k] * 128 / 1e6:8.1f}")
    json.dump({"source": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum (one pass)", "kernels": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
