#!/usr/bin/env python3
"""Where the wave-cycles of each kernel class go (rocprofv3 SQ counters, one pass):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \\
        -d <dir> --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile
    python tools/pmc_sq.py <dir> <out.json>

WAIT_ANY = parked in s_waitcnt / barrier, WAIT_INST_ANY = issue-stalled, ACTIVE_INST_ANY = issuing (disjoint, quad-cycle
units, MI355X_MICROARCH.md 'rocprofv3 PMC slots'); MFMA busy is in cycles summed over SIMDs."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import base


def main():
    d, out = sys.argv[1:3]
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = base(row["Kernel_Name"])
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "SQ_WAVE_CYCLES":
                    n[k] += 1
    res = {}
    for k, c in tot.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if not wc or n[k] < 4:
            continue
        res[k] = {"launches": n[k], "wave_cycles_per_launch": wc / n[k],
                  "parked_frac": c.get("SQ_WAIT_ANY", 0) / wc, "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                  "issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                  "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n[k],
                  "lds_bank_conflict_frac": (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
                  "busy_cycles_per_launch": c.get("SQ_BUSY_CYCLES", 0) / n[k]}
        r = res[k]
        print(f"{k:28s} n={n[k]:4d} parked {r['parked_frac']:.2f} issue-stall {r['issue_stall_frac']:.2f} issuing {r['issuing_frac']:.2f} "
              f"lds-conflict {r['lds_bank_conflict_frac'] if r['lds_bank_conflict_frac'] is None else round(r['lds_bank_conflict_frac'], 3)}")
    json.dump({"source": "rocprofv3 --pmc SQ_* (one pass) on `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile`", "kernels": res},
              open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
