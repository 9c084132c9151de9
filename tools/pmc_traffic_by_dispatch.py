#!/usr/bin/env python3
"""Memory-side bytes PER LAUNCH POSITION of one denoiser call, against what the launch must move.

    python tools/pmc_traffic_by_dispatch.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <out.json> [B L Lt]

The per-function averages of tools/pmc_traffic.py (enc_bc_kernel: 45 MB per launch against 30 MB "algorithmic", the 1.49x of the
round-4 review) mix four different launches.  Here every dispatch is attributed to its POSITION in the denoiser call (the four
enc_bc launches: enc3.bc | enc5.bc -> pool -> att_dense -> att0.a | att0.bc -> att1.a | att1.bc; the six ConvBlocks; enc5.a) and
compared with a prediction made of two parts:

  activations : every tensor the launch reads or writes, once (bf16): the algorithmic count of dhw_api.cpp;
  weights x 8 : the launch's packed weights, fetched once PER XCD.  The eight XCDs have private 4 MiB L2s (MI355X_MICROARCH.md,
                "L2 (per XCD)": not shared, not coherent), every kernel here has workgroups on all eight, and each of them needs the
                whole layer's weights — so the fabric delivers eight copies per launch whatever the kernel does.  FETCH_SIZE counts
                fabric requests, Infinity-Cache hits included (same guide, HBM section), so these copies show up as "traffic" although
                the 20 MB of weights never leave the 256 MB Infinity Cache.

FETCH_SIZE x 2 and KiB -> bytes as in tools/pmc_traffic.py."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import base  # noqa: E402

ES = 2  # bf16


def per_dispatch(d, counter):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter:
                    rows.append((int(row["Dispatch_Id"]), base(row["Kernel_Name"]), float(row["Counter_Value"])))
    agg = defaultdict(float)   # (some rocprofv3 builds emit one row per XCD / SE instance: sum them per dispatch)
    name = {}
    for did, k, v in rows:
        agg[did] += v
        name[did] = k
    return [(did, name[did], agg[did]) for did in sorted(agg)]


def predictions(B, L, Lt, chain5=True):
    """position -> (label, activation bytes, weight bytes) for the reference dims c = (128, 192, 256), num_layers = 2"""
    c1, c2, c3, dt = 128, 192, 256, 384
    rows = lambda lv: B * L // lv   # noqa: E731

    def cb_w(cin, cout, up=0):       # conv1 + conv_skip (3 taps x cin), conv2 (3 taps x cout / 2), fc (+ the fused skip_conv of a decoder block)
        return (3 * cin * cout // 2 + 3 * cin * cout + 3 * (cout // 2) * cout + cout * cout + 3 * up * cin) * ES

    def cb_a(r, cin, cout, pool=False, up=0, out_f32=False, heads=False):
        a = r * (up + cin // 2 if up else cin) * ES
        a += 0 if heads else r * cout * (4 if out_f32 else ES) * (1.5 if pool else 1.0)
        return a

    def enc_a(r, d, lk, chained):
        """reads x (taken from LDS when chained behind the previous kernel's tile) + the text K / V of the step, writes x2 + [q2 | k2 | v2];
        weights: q1, dense1, qkv2 = 5 d^2, + the fp32 PE.W bias tables [Lk][d] and [Lk][2 d] (shared by all samples: fetched per XCD too)"""
        return r * d * ES * (4 if chained else 5) + B * Lt * d * 2 * ES, 5 * d * d * ES + lk * 3 * d * 4

    def enc_bc(r, d, pool=False):   # reads x2 + [q2|k2|v2], writes out (+ pool); weights: dense2, ffn = 5 d^2
        return r * d * ES * (5 + (0.5 if pool else 0.0)), 5 * d * d * ES

    # (enc4 continues into enc5.a since round 5 on the asymmetric 32-row tiles: DHW_CHAIN_CONV default; pass chain5=False for older builds)
    a3c, a5, aac = enc_a(rows(2), c2, L // 2, True), enc_a(rows(4), c3, L // 4, chain5), enc_a(rows(8), dt, L // 8, True)
    conv = [("enc1", rows(1) * 2 * 4 + rows(1) * c1 * ES * 1.5, cb_w(c1, c1)),     # (reads the 2-float strokes, writes out + pool)
            ("enc2 -> enc3.a", cb_a(rows(2), c1, c2) + a3c[0], cb_w(c1, c2) + a3c[1]),
            ("enc4 -> enc5.a", cb_a(rows(4), c2, c3) + a5[0], cb_w(c2, c3) + a5[1]) if chain5 else ("enc4", cb_a(rows(4), c2, c3), cb_w(c2, c3)),
            ("dec3", cb_a(rows(4), dt, c3, up=c3), cb_w(dt, c3, up=c3)),
            ("dec2", cb_a(rows(2), c3, c2, up=c2), cb_w(c3, c2, up=c2)),
            ("dec1 (+ heads, scheduler step)", cb_a(rows(1), c2, c1, up=c1, heads=True) + rows(1) * 2 * 4 * 2, cb_w(c2, c1, up=c1))]
    bc3, bc5, bca = enc_bc(rows(2), c2, pool=True), enc_bc(rows(4), c3, pool=True), enc_bc(rows(8), dt)
    encbc = [("enc3.bc", bc3[0], bc3[1]),
             ("enc5.bc -> pool -> att_dense -> att0.a", bc5[0] + rows(8) * dt * ES + aac[0], bc5[1] + c3 * dt * ES + aac[1]),
             ("att0.bc -> att1.a", bca[0] + aac[0], bca[1] + aac[1]),
             ("att1.bc", bca[0], bca[1])]
    out = {"convblock_kernel": conv, "enc_bc_kernel": encbc}
    if not chain5:
        out["enc_a_kernel"] = [("enc5.a", a5[0], a5[1])]
    return out


def main():
    fdir, wdir, out = sys.argv[1:4]
    B, L, Lt = (int(x) for x in sys.argv[4:7]) if len(sys.argv) >= 7 else (64, 488, 30)
    pred = predictions(B, L, Lt, chain5=os.environ.get("DHW_CHAIN_CONV", "3") not in ("0", "1"))
    res = {}
    for counter, d, scale in (("FETCH_SIZE", fdir, 2048.0), ("WRITE_SIZE", wdir, 1024.0)):
        count = defaultdict(int)
        for did, k, v in per_dispatch(d, counter):
            if k not in pred:
                continue
            pos = count[k] % len(pred[k])
            count[k] += 1
            e = res.setdefault(k, [dict(label=p[0], activation_bytes=p[1], weight_bytes=p[2], fetch=0.0, write=0.0, n_fetch=0, n_write=0) for p in pred[k]])[pos]
            e["fetch" if counter == "FETCH_SIZE" else "write"] += v * scale
            e["n_fetch" if counter == "FETCH_SIZE" else "n_write"] += 1
    table = {}
    for k, entries in res.items():
        rows = []
        for e in entries:
            f = e["fetch"] / max(e["n_fetch"], 1)
            w = e["write"] / max(e["n_write"], 1)
            once, x8 = e["activation_bytes"] + e["weight_bytes"], e["activation_bytes"] + 8 * e["weight_bytes"]
            rows.append({"launch": e["label"], "dispatches": e["n_fetch"], "pmc_fetch_bytes": f, "pmc_write_bytes": w, "pmc_bytes": f + w,
                         "algorithmic_bytes_weights_once": once, "predicted_bytes_weights_per_xcd": x8,
                         "ratio_vs_algorithmic": (f + w) / once, "ratio_vs_predicted": (f + w) / x8})
            print(f"{k:18s} {e['label']:42s} PMC {(f + w) / 1e6:7.2f} MB (fetch {f / 1e6:6.2f} write {w / 1e6:6.2f})  algorithmic {once / 1e6:6.2f}  "
                  f"+ 7 more weight copies {x8 / 1e6:6.2f}   x{(f + w) / once:4.2f} / x{(f + w) / x8:4.2f}")
        tot_p, tot_o, tot_8 = (sum(r[c] for r in rows) for c in ("pmc_bytes", "algorithmic_bytes_weights_once", "predicted_bytes_weights_per_xcd"))
        table[k] = {"per_launch_position": rows, "sum_over_a_call": {"pmc_bytes": tot_p, "algorithmic_bytes_weights_once": tot_o,
                    "predicted_bytes_weights_per_xcd": tot_8, "ratio_vs_algorithmic": tot_p / tot_o, "ratio_vs_predicted": tot_p / tot_8}}
        print(f"{k:18s} {'SUM over one denoiser call':42s} PMC {tot_p / 1e6:7.2f} MB  algorithmic {tot_o / 1e6:6.2f}  with per-XCD weights {tot_8 / 1e6:6.2f}   "
              f"x{tot_p / tot_o:4.2f} / x{tot_p / tot_8:4.2f}")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per dispatch, grouped by launch position in the denoiser call; "
                         "prediction = activations once + packed weights once per XCD (8 private L2s)", "B": B, "L": L, "Lt": Lt, "kernels": table},
              open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
