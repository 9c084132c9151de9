"""Round-2 GPU tests (-m gpu): graph cache vs interleaved schedule lengths, the device noise generator on its own,
pen-lift bit flips of the bf16 path at the bench size, and BASELINE configs[3] (L=1000, T=1000, B=32) at full size."""
import json
import os

import numpy as np
import pytest
import torch

import dhg_amd
from dhg_amd import spec
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sd(nl=2, out_scale=None):
    d = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(nl).items()}
    if out_scale is not None:   # see test_config3_*: keeps the random-init T=1000 trajectory finite
        d["output_dense.weight"] = d["output_dense.weight"] * out_scale
        d["output_dense.bias"] = d["output_dense.bias"] * out_scale
    return d


def _model(prec, B, L, Lt, sd=None):
    m = dhg_amd.DiffusionModel(2, precision=prec, max_B=B, max_L=L, max_Lt=Lt).eval()
    m.load_state_dict(sd or _sd())
    return m


def test_cached_graphs_survive_interleaved_schedule_lengths():
    """One handle sampled at T = 8, 4, 16, 8: every schedule length owns its FiLM table, so the cached T=8
    graph must replay against the T=8 table after other lengths ran (a shared, re-grown table let it read T=4's)."""
    B, L, Lt = 3, 64, 6
    inp = spec.synthetic_inputs(B, L, Lt, seed=17, T=0)
    tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
    m = _model("bf16", B, L, Lt)
    got = [dhg_amd.sample(m, tx, sv, L=L, T=T, seed=5).cpu() for T in (8, 4, 16, 8, 4, 32, 16, 8)]
    for T, out in zip((8, 4, 16, 8, 4, 32, 16, 8), got):
        fresh = _model("bf16", B, L, Lt)
        want = dhg_amd.sample(fresh, tx, sv, L=L, T=T, seed=5).cpu()
        assert torch.equal(out, want), T
        del fresh


def test_device_noise_generator_moments_and_independence():
    """The Philox4x32-10 + Box-Muller generator read back directly (dhw_debug_randn), B=64 x L=488 x 2 = 62 464 draws per
    iteration: moments, normality (KS), serial / cross-component / cross-iteration correlation, shard invariance."""
    from scipy import stats
    B, L = 64, 488
    m = _model("bf16", B, L, 4)
    x = m.debug_randn(seed=123, first_sample=0, B=B, L=L, it=-1).numpy().astype(np.float64)
    z0 = m.debug_randn(seed=123, first_sample=0, B=B, L=L, it=0).numpy().astype(np.float64)
    n = x.size
    se = 1.0 / np.sqrt(n)
    for v in (x, z0):
        f = v.ravel()
        assert abs(f.mean()) < 5 * se
        assert abs(f.var() - 1.0) < 5 * np.sqrt(2.0 / n)
        assert abs(stats.skew(f)) < 5 * np.sqrt(6.0 / n)
        assert abs(stats.kurtosis(f)) < 5 * np.sqrt(24.0 / n)          # excess kurtosis: 0 for N(0,1), -1.2 for uniform
        assert stats.kstest(f, "norm").pvalue > 1e-3
        assert np.abs(f).max() > 3.5                                    # tails exist (62k draws: P(max < 3.5) ~ 1e-13)
        seq = v[..., 0]                                                 # along the stroke axis
        assert abs(np.corrcoef(seq[:, :-1].ravel(), seq[:, 1:].ravel())[0, 1]) < 5 * se * np.sqrt(2)   # lag-1
        assert abs(np.corrcoef(v[..., 0].ravel(), v[..., 1].ravel())[0, 1]) < 5 * se * np.sqrt(2)       # the Box-Muller pair
        assert abs(np.corrcoef(v[:-1].ravel(), v[1:].ravel())[0, 1]) < 5 * se                           # neighbouring samples
    assert abs(np.corrcoef(x.ravel(), z0.ravel())[0, 1]) < 5 * se       # iterations are independent streams
    other = m.debug_randn(seed=124, first_sample=0, B=B, L=L, it=-1).numpy()
    assert abs(np.corrcoef(x.ravel(), other.ravel())[0, 1]) < 5 * se    # seeds are independent streams
    # keyed by the GLOBAL sample index: a shard starting at prompt 40 draws exactly rows 40.. of the full batch
    shard = m.debug_randn(seed=123, first_sample=40, B=8, L=L, it=-1).numpy()
    assert np.array_equal(shard, x[40:48].astype(np.float32))


def test_device_noise_is_what_the_sampler_consumes():
    """dhw_sample(noise=NULL) == dhw_sample(noise = the generator's draws handed over explicitly), bit for bit."""
    B, L, Lt, T = 4, 64, 5, 3
    inp = spec.synthetic_inputs(B, L, Lt, seed=3, T=0)
    tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
    m = _model("fp32", B, L, Lt)
    a = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=77, first_sample=9).cpu()
    nz = torch.stack([m.debug_randn(77, 9, B, L, it) for it in range(-1, T)])
    b = dhg_amd.sample(m, tx, sv, L=L, T=T, noise=nz.cuda()).cpu()
    assert torch.equal(a, b)


def test_bf16_pen_flips_at_the_bench_size_are_confined_to_ties():
    """B=64, T=60, L=488 (the bench configuration), bf16 path vs this library's fp32 path (pinned to the reference at 1e-3 /
    identical pen bits by the goldens) on the same external noise.  north_star asks for pen-lift decisions bit-exact after
    rounding: fp32 mode is; bf16 mode differs ONLY where the fp32 probability itself sits within 0.012 of the 0.5
    threshold (bf16 weights and activations move p by up to ~0.015).  Measured (r2): 60 of 31 232 bits, all with
    |p - 0.5| < 0.0099; max |dp| 0.0151; trajectory within 0.6 % of max|x|."""
    B, L, Lt, T = 64, 488, 30, 60
    inp = spec.synthetic_inputs(B, L, Lt, seed=31, T=T)
    tx, sv, nz = (torch.from_numpy(inp[k]).cuda() for k in ("text", "style", "noise"))
    outs = {}
    for prec in ("fp32", "bf16"):
        m = _model(prec, B, L, Lt)
        outs[prec] = dhg_amd.sample(m, tx, sv, L=L, T=T, noise=nz).cpu().numpy()
        del m
    ref, got = outs["fp32"], outs["bf16"]
    flipped = np.round(ref[..., 2]) != np.round(got[..., 2])
    dist = np.abs(ref[..., 2][flipped] - 0.5)
    report = {"flipped": int(flipped.sum()), "of": int(flipped.size), "max_dist_to_half": float(dist.max()) if flipped.any() else 0.0,
              "max_abs_dp": float(np.abs(ref[..., 2] - got[..., 2]).max()),
              "traj_rel_err": float(np.abs(ref[..., :2] - got[..., :2]).max() / np.abs(ref[..., :2]).max())}
    print("bf16 pen flips at the bench size:", json.dumps(report))
    assert report["max_abs_dp"] < 0.03
    assert np.all(dist < 0.02), report                  # a flip only where the fp32 path's own p is a near-tie
    assert flipped.mean() < 0.005, report               # 0.19 % measured; ~0.5 % of all p lie within 0.005 of 0.5
    assert report["traj_rel_err"] < 0.02, report


OUT_SCALE = 0.05   # configs[3]: output_dense x 0.05 (see the test below)


def test_config3_long_schedule_full_size_properties():
    """BASELINE configs[3]: L=1000 strokes, Lt=62, T=1000 steps, B=32 prompts per GPU, bf16.  With torch-default random
    init the REFERENCE arithmetic itself overflows by step ~900 (the un-normalised skip path feeds |x| ~ 1e13 back through
    eps; the oracle reaches NaN too), so the eps head is scaled by 0.05 to keep the trajectory finite (|x| ends ~1e15,
    the schedule's own 1/sqrt(abar_T) gain).  Properties at full size: finite, deterministic, shard-invariant, the
    text plane's 16 chunks of 64 steps stitched correctly (plane on == plane off)."""
    B, L, Lt, T = 32, 1000, 62, 1000
    inp = spec.synthetic_inputs(B, L, Lt, seed=5, T=0)
    tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
    m = _model("bf16", B, L, Lt, _sd(out_scale=OUT_SCALE))
    full = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=3).cpu()
    assert full.shape == (B, L, 3) and torch.isfinite(full).all()
    assert ((full[..., 2] >= 0) & (full[..., 2] <= 1)).all()
    again = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=3).cpu()
    assert torch.equal(full, again)
    shard = dhg_amd.sample(m, tx[8:12].contiguous(), sv[8:12].contiguous(), L=L, T=T, seed=3, first_sample=8).cpu()
    assert torch.equal(shard, full[8:12])
    del m
    old = os.environ.get("DHW_PLANE")
    os.environ["DHW_PLANE"] = "0"
    try:
        m2 = _model("bf16", 4, L, Lt, _sd(out_scale=OUT_SCALE))
        noplane = dhg_amd.sample(m2, tx[8:12].contiguous(), sv[8:12].contiguous(), L=L, T=T, seed=3, first_sample=8).cpu()
    finally:
        if old is None:
            os.environ.pop("DHW_PLANE", None)
        else:
            os.environ["DHW_PLANE"] = old
    assert torch.equal(noplane, full[8:12])


def test_config3_long_schedule_matches_the_oracle(golden_dir):
    """BASELINE configs[3] (L=1000, Lt=62, T=1000) against the oracle, TEACHER FORCED (round 3; SURVEY 7 "hard parts"): the
    reverse process of a random-init model gains 1/sqrt(1-beta) per step (1e15 over this schedule), so the state is
    captured and reset to a seeded N(0,1) draw every 16 steps — in the oracle (ref_cpu.sample(teacher=...), fixture
    tests/golden/config3_oracle.npz written by oracle/make_config3_fixture.py) and in the library
    (dhw_debug_set_teacher).  |x| stays O(10), so the tolerances mean something, while all 1000 schedule indices, FiLM rows
    and the 16 text-plane chunks are exercised on two prompts with the plain synthetic weights:
    fp32 — every captured state and the final strokes within 1e-4 of max|x|, pen bits identical;
    bf16 — within 2 % of max|x|, pen bits may differ only where the oracle's probability is within 0.02 of 0.5."""
    g = np.load(os.path.join(golden_dir, "config3_oracle.npz"))
    B, L, Lt, T, every = (int(g[k]) for k in ("B", "L", "Lt", "T", "every"))
    assert (B, L, Lt, T, every) == (2, 1000, 62, 1000, 16)
    rs = int(g["row_stride"])
    ref_out, ref_cap = torch.from_numpy(g["out"]), torch.from_numpy(g["captures"])
    resets = torch.randn((T - 1) // every, B, L, 2, generator=torch.Generator().manual_seed(int(g["reset_seed"])))
    inp = spec.synthetic_inputs(B, L, Lt, seed=int(g["seed"]), T=T)
    tx, sv, nz = (torch.from_numpy(inp[k]) for k in ("text", "style", "noise"))
    scale = max(ref_out[..., :2].abs().max().item(), ref_cap.abs().max().item())
    assert scale < 1e3, scale          # the forced trajectory stays in a meaningful range
    rep = {"max_abs_x": scale}
    for prec, tol_x in (("fp32", 1e-4), ("bf16", 0.02)):
        m = _model(prec, B, L, Lt)
        cap = m.set_teacher(resets.cuda(), every)
        out = dhg_amd.sample(m, tx.cuda(), sv.cuda(), L=L, T=T, noise=nz.cuda()).cpu()
        cap = cap.cpu()[:, :, ::rs]
        m.set_teacher(None)
        e_cap = (cap - ref_cap).abs().amax(dim=(1, 2, 3)) / scale            # per 16-step segment
        e_out = (out[..., :2] - ref_out[..., :2]).abs().max().item() / scale
        flipped = out[..., 2].round() != ref_out[..., 2].round()
        dist = (ref_out[..., 2][flipped] - 0.5).abs()
        rep[prec] = {"rel_err_segments_max": e_cap.max().item(), "rel_err_final": e_out, "pen_bits_flipped": int(flipped.sum()),
                     "max_dist_to_half_of_flips": dist.max().item() if flipped.any() else 0.0,
                     "max_abs_dp": (out[..., 2] - ref_out[..., 2]).abs().max().item()}
        print("configs[3] teacher-forced vs oracle:", json.dumps(rep))
        assert torch.isfinite(out).all() and torch.isfinite(cap).all()
        assert e_cap.max().item() < tol_x and e_out < tol_x, rep
        if prec == "fp32":
            assert not flipped.any(), rep
        else:
            assert (dist < 0.02).all() and flipped.float().mean().item() < 0.01, rep
        del m


def test_bench_prints_exactly_one_json_line_with_rccl_initialised():
    """`bench.py --force-dist`: the multi-GPU form's process group (backend "nccl" = RCCL: barrier and max-over-ranks of the wall time)
    on a one-rank group, so the collective path runs on a 1-GPU box.  The GPU boxes export NCCL_DEBUG=VERSION and RCCL prints its
    banner to STDOUT: the rank must still put exactly ONE line on stdout, the JSON line (bench.py routes fd 1 to stderr)."""
    import subprocess
    import sys
    env = dict(os.environ, NCCL_DEBUG="VERSION", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-fp32",
                        "--no-train-step", "--no-longseq", "--no-kernel-profile"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["metric"].startswith("stroke-points")
