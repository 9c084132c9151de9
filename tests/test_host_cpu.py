"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol the headers
declare, the schedule is pinned to the reference's, the state_dict inventory and tokenizer match the
reference, and the product path fails loudly (never falls back) when no HIP device is present."""
import ctypes as C
import json
import math
import os
import re

import numpy as np
import pytest
import torch

import dhg_amd
from dhg_amd import _lib, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for h in ("dhw.h", "dhw_debug.h", "dhw_style.h", "dhw_train.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b(dhw_[a-z0-9_]+)\s*\(", src))
    return syms


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.lib()
    declared = _declared_symbols()
    assert {"dhw_create", "dhw_load", "dhw_forward", "dhw_sample", "dhw_destroy", "dhw_last_error"} <= declared
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes binding and headers disagree"
    assert b"gfx950" in lib.dhw_version()


def test_schedule_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "sched.npz"))
    beta, alpha = _lib.schedule(60)
    # linspace (fma on both halves) and the double-accumulated cumprod are reproduced bit for bit; torch's
    # vectorised expf (SLEEF u10) and libm's expf may disagree in the last bit on isolated entries
    ulp_b = np.abs(beta.view(np.int32).astype(np.int64) - g["beta"].view(np.int32).astype(np.int64))
    ulp_a = np.abs(alpha.view(np.int32).astype(np.int64) - g["alpha"].view(np.int32).astype(np.int64))
    assert ulp_b.max() <= 1 and (ulp_b != 0).sum() <= 2, ulp_b
    assert ulp_a.max() <= 1 and (ulp_a != 0).sum() <= 2, ulp_a
    assert np.allclose(dhg_amd.get_beta_set().numpy(), g["beta"], rtol=2e-7, atol=0)
    assert np.allclose(dhg_amd.get_alpha_set().numpy(), g["alpha"], rtol=1e-6, atol=0)


def test_schedule_generalises_in_T():
    b60, _ = _lib.schedule(60)
    b1000, a1000 = _lib.schedule(1000)
    assert abs(b1000[0] - b60[0]) < 1e-7 and abs(b1000[-1] - b60[-1]) < 1e-6
    assert np.all(np.diff(b1000) > 0) and np.all(np.diff(a1000) < 0) and np.isfinite(a1000).all()
    assert _lib.lib().dhw_schedule(0, None, None) < 0
    # every small schedule length against the reference's formula evaluated by torch (utils/nn.py:19-39, inference.py:81) — T = 1
    # included: a one-point torch.linspace is its start, which the symmetric two-sided evaluation alone got wrong until round 3
    for T in (1, 2, 3, 4, 5, 8, 16, 61):
        beta = 0.02 + torch.exp(torch.linspace(math.log(1e-5), math.log(0.4), T))
        alpha = torch.cumprod(1 - beta, dim=0)
        b, a = _lib.schedule(T)
        assert np.allclose(b, beta.numpy(), rtol=3e-7, atol=0), T
        assert np.allclose(a, alpha.numpy(), rtol=1e-6, atol=0), T


@pytest.mark.parametrize("nl", [2, 4])
def test_state_dict_inventory_matches_reference(golden_dir, nl):
    with open(os.path.join(golden_dir, "keys.json")) as f:
        ref = json.load(f)[str(nl)]
    ours = [[n, list(s)] for n, s, _ in spec.param_spec(nl)]
    assert ours == ref
    m = dhg_amd.DiffusionModel(nl)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref
    assert sum(p.numel() for p in m.parameters()) == (10028451 if nl == 2 else 14074275)
    with pytest.raises(RuntimeError):  # strict load, as the reference's checkpoint loader
        m.load_state_dict({"bogus.weight": torch.zeros(1)})


def test_synthetic_weights_are_portable_and_order_independent():
    a = spec.synthetic_state_dict(2, seed=0)
    b = spec.synthetic_state_dict(4, seed=0)
    for k in a:
        assert np.array_equal(a[k], b[k]), k   # keyed by tensor name, not by position
    assert np.all(a["enc1.affine1.gamma_emb.bias"] == 1)
    assert abs(float(a["enc1.fc.weight"].max()) - 1 / np.sqrt(128)) < 1e-3
    s0 = spec.synthetic_inputs(4, 16, 5, seed=9)
    s1 = spec.synthetic_inputs_range(2, 2, 16, 5, seed=9)
    for k in ("text", "style", "strokes"):
        assert np.array_equal(s0[k][2:], s1[k])
    assert np.array_equal(s0["noise"][:, 2:], s1["noise"])


def test_tokenizer_known_answers(golden_dir):
    with open(os.path.join(golden_dir, "tokenizer.json")) as f:
        ka = json.load(f)
    tk = dhg_amd.Tokenizer()
    for c in ka:
        ids = tk.encode(c["prompt"])
        assert ids == c["ids"]
        assert dhg_amd.stroke_length(len(ids)) == c["L"]
    assert tk.decode(tk.encode("Hi there")[:-1]) == "Hi there"
    assert tk.vocab_size == 73


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_a_gpu():
    dims = _lib.DhwDims(2, 128, 192, 256, 1, 8, 1, 14, 0)
    h = C.c_void_p()
    rc = _lib.lib().dhw_create(C.byref(h), C.byref(dims), 0)
    assert rc == -3 and not h.value
    assert b"HIP device" in _lib.lib().dhw_last_error(None)
    m = dhg_amd.DiffusionModel(2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 8, 2), torch.ones(1, 3, dtype=torch.long), torch.ones(1, 1), torch.zeros(1, 14, 1280))


def test_token_ids_outside_the_vocabulary_raise_like_nn_embedding():
    """reference text_style.py:71 `nn.Embedding(73, ...)` raises IndexError; the kernels would clamp silently"""
    from dhg_amd.model import check_token_ids
    check_token_ids(torch.tensor([[0, 1, 72]]))
    for bad in ([[73]], [[-1, 3]], [[5, 1000]]):
        with pytest.raises(IndexError):
            check_token_ids(torch.tensor(bad))
    # the per-prompt cache of validated tensors: same object + same version is not re-read, an in-place write is; entries are weak
    seen = []
    t = torch.tensor([[1, 2, 3]])
    check_token_ids(t, seen)
    check_token_ids(t, seen)
    assert len(seen) == 1
    t[0, 1] = 99                       # (bumps _version)
    with pytest.raises(IndexError):
        check_token_ids(t, seen)
    del t
    import gc
    gc.collect()
    assert all(ref() is None for ref, _, _ in seen)   # a validated prompt tensor is not kept alive by the cache
    m = dhg_amd.DiffusionModel(2)
    with pytest.raises(IndexError):    # checked before the device is touched
        m(torch.zeros(1, 8, 2), torch.tensor([[73]]), torch.ones(1, 1), torch.zeros(1, 14, 1280))
    with pytest.raises(IndexError):
        dhg_amd.sample(m, torch.tensor([[1, 99]]), torch.zeros(1, 14, 1280), L=8)


def test_create_rejects_bad_dims():
    l = _lib.lib()
    h = C.c_void_p()
    for bad in (dict(c1=64), dict(c3=128), dict(c2=100), dict(c2=204), dict(c2=0), dict(max_L=10), dict(precision=7), dict(S=0)):
        kw = dict(num_layers=2, c1=128, c2=192, c3=256, max_B=1, max_L=8, max_Lt=1, S=14, precision=0)
        kw.update(bad)
        assert l.dhw_create(C.byref(h), C.byref(_lib.DhwDims(**kw)), 0) == -1, bad
    assert l.dhw_create(None, None, 0) == -1


def test_no_exception_crosses_the_c_abi():
    """include/dhw.h: "nothing throws across the ABI".  Round 4's logs hold three host-process aborts from a std::out_of_range
    (std::map::at on a workspace name) that left dhw_forward.  dhw_debug_raise throws INSIDE the guarded body of an entry point
    — the same std::map::at miss, a std::bad_alloc, a non-std exception —: each must come back as DHW_ERR_INTERNAL (-5) with
    a message, and this process must still be alive to assert it."""
    l = _lib.lib()
    for kind, needle in ((1, b"map::at"), (2, b"bad_alloc"), (3, b"unknown C++ exception")):
        rc = l.dhw_debug_raise(None, kind)
        assert rc == -5, (kind, rc)
        msg = l.dhw_last_error(None)
        assert b"dhw_debug_raise: internal error" in msg and needle in msg, msg
    assert l.dhw_debug_raise(None, 99) == -1          # an ordinary argument error still reads as one
    assert b"gfx950" in l.dhw_version()                # the library (and the interpreter) survived


def test_every_entry_point_runs_behind_the_barrier():
    """Structural: every extern "C" function with a multi-line body in the three API files opens with its file's *_GUARD macro
    (csrc/abi_guard.h), so a new entry point cannot be added without the barrier."""
    d = os.path.join(ROOT, "diffusion-handwriting-generation.pytorch_amd", "csrc")
    total = 0
    for f, guard in (("dhw_api.cpp", "DHW_GUARD("), ("dhw_style_api.cpp", "STYLE_GUARD("), ("dhw_train_api.cpp", "TRAIN_GUARD(")):
        src = open(os.path.join(d, f)).read()
        blocks = src.split('extern "C" {')[1:]
        assert blocks, f
        for blk in blocks:
            blk = blk.split('}  // extern "C"')[0]
            for m in re.finditer(r"^(?:int|int64_t) (dhw_\w+)\([^{;]*\{\n(.*?)^\}", blk, flags=re.S | re.M):
                name, body = m.group(1), m.group(2)
                assert body.lstrip().startswith(guard), f"{f}: {name} is not behind {guard}...)"
                total += 1
    assert total >= 50, total


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "diffusion-handwriting-generation.pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src and "/root/reference" not in src, f


def test_xcd_swizzle_is_a_bijection_with_contiguous_ranges_per_xcd():
    """Every fused kernel derives (sample, row tile) from this id: it must visit each tile exactly once for any grid,
    and the blocks the dispatcher deals to one XCD (blockIdx % 8) must get one contiguous id range."""
    f = _lib.lib().dhw_debug_xcd_swizzle
    for nwg in (1, 2, 7, 8, 9, 15, 64, 100, 192, 255, 256, 257, 1001):
        ids = [f(i, nwg) for i in range(nwg)]
        assert sorted(ids) == list(range(nwg)), nwg
        for x in range(min(8, nwg)):
            mine = sorted(ids[i] for i in range(x, nwg, 8))
            assert mine == list(range(mine[0], mine[0] + len(mine))), (nwg, x)


def test_kernel_source_hash_covers_csrc():
    """bench.py marks the committed PMC traffic figures stale when the sampler's kernel sources change: every file under csrc/
    must be either hashed (SAMPLER_SOURCES) or explicitly listed as not part of the sampling path (NON_SAMPLER_SOURCES)."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(bench)
    d = os.path.join(ROOT, "diffusion-handwriting-generation.pytorch_amd", "csrc")
    files = {f for f in os.listdir(d) if os.path.isfile(os.path.join(d, f))}
    assert not (bench.SAMPLER_SOURCES & bench.NON_SAMPLER_SOURCES)
    unknown = files - bench.SAMPLER_SOURCES - bench.NON_SAMPLER_SOURCES
    assert not unknown, f"csrc files in neither list of bench.py: {sorted(unknown)}"
    missing = (bench.SAMPLER_SOURCES | bench.NON_SAMPLER_SOURCES) - files
    assert not missing, f"bench.py lists csrc files that do not exist: {sorted(missing)}"
    h1 = bench.kernel_source_hash()
    assert len(h1) == 16 and h1 == bench.kernel_source_hash()
