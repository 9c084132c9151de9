#!/usr/bin/env python3
"""Memory-side bytes per launch and kernel class from two rocprofv3 PMC passes of the same command
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, collected separately: they do not fit one pass on gfx950).

    python tools/pmc_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is
doubled (MI355X_MICROARCH.md, HBM section).  Kernels are grouped by their template name."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def base(name):
    """kernel template name: `convblock_kernel` from either the demangled or the Itanium-mangled symbol"""
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1))
        start = m.end()
        return name[start:start + n]
    name = re.sub(r"\(.*$", "", name)
    name = name.split("<")[0]
    name = name.replace("void ", "").split("::")[-1]
    return name.replace(".kd", "").strip()


def collect(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = base(row["Kernel_Name"])
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt)):
        n = fc.get(k) or wc.get(k)
        fetch = ft.get(k, 0.0) * 1024 * 2 / max(fc.get(k, 0), 1)
        write = wt.get(k, 0.0) * 1024 / max(wc.get(k, 0), 1)
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                      "hbm_bytes_per_launch": fetch + write}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash   # the csrc/ hash these passes were taken at: bench.py marks the figure stale when it differs
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 1 --warmup 1 "
                         "--no-cpu-baseline --no-kernel-profile --no-fp32`; KiB -> bytes, FETCH_SIZE x2 per MI355X_MICROARCH.md (HBM section)",
               "kernel_source_hash": kernel_source_hash(), "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:32s} launches {v['launches']:5d}  fetch {v['fetch_bytes_per_launch']/1e6:8.2f} MB  write {v['write_bytes_per_launch']/1e6:8.2f} MB")


if __name__ == "__main__":
    main()
