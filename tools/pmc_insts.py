#!/usr/bin/env python3
"""Instruction mix per kernel function (rocprofv3 SQ instruction counters, one pass, per launch and per wave):
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d <dir> --output-format csv -- <bench>
    python tools/pmc_insts.py <dir> <out.json>
Price list (MI355X, measured by tools/experiments/bench_pipe_overlap.cpp): one wave64 VALU instruction occupies its SIMD's VALU
issue for ~4.3 cycles whatever the number of waves; a 16x16x32 bf16 MFMA 16 cycles of the matrix pipe (half of them also block VALU
issue); one 16-byte-per-lane vector-memory instruction ~17 cycles of the CU's 64 B/clk L1 path."""
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
                k = row["Kernel_Name"]
                k = base(k) + ("<" + k.split("<", 1)[1][:40] if "<" in k and ("convblock" in k or "enc_" in k) else "")
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "SQ_WAVES": n[k] += 1
    res = {}
    for k, c in sorted(tot.items()):
        if n[k] < 2 or not c.get("SQ_WAVES"): continue
        w = c["SQ_WAVES"]
        r = {"launches": n[k], "waves_per_launch": w / n[k]}
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if name in c: r[name.lower() + "_per_wave"] = c[name] / w
        res[k] = r
        print(k[:70].ljust(70), " ".join(f"{kk[9:-9]}={vv:8.1f}" for kk, vv in r.items() if kk.endswith("_per_wave")))
    json.dump({"source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_* (one pass), per wave", "kernels": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
