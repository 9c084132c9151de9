"""The whole training step in HIP (dhg_amd/train_model.py over include/dhw_train.h dhw_op_*): each generic operation against
torch's CPU operator + autograd on the same seeded inputs, then DiffusionModel forward + loss.backward() against the fixture
the imported reference generated (tests/golden/model_grad.npz, oracle/make_golden_r2.py model_grad_fixture): outputs, losses
and ALL 323 parameter gradients (norm + projection on a fixed random direction each, a few small ones element-wise)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dhg_amd
from dhg_amd import spec, train, train_model as tm

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def probe(index: int, n: int) -> np.ndarray:
    """The fixed random direction oracle/make_golden_r2.py projected parameter gradient number ``index`` on."""
    return np.random.Generator(np.random.PCG64([77, index])).standard_normal(n).astype(np.float32)


def _var(t):
    return tm.Var(t.detach().to(DEV, torch.float32).contiguous())


def _close(a, b, tol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-6)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert float((a - b).abs().max()) <= tol * scale, (float((a - b).abs().max()), scale)


def _run(build, inputs, dout_seed=9):
    """build(tape, *Vars) -> Var;  the same function on torch tensors gives the reference via autograd."""
    g = torch.Generator().manual_seed(dout_seed)
    tape = tm.Tape(torch.device(DEV))
    vs = [_var(t) for t in inputs]
    y = build(tape, *vs)
    dy = torch.randn(y.d.shape, generator=g)
    y.g = dy.to(DEV)
    tape.backward()
    torch.cuda.synchronize()
    return y, vs, dy


def test_linear_and_conv3_in_all_three_directions():
    g = torch.Generator().manual_seed(1)
    # (40, 72): not multiples of 16 / 32 — the GEMM's edge masking, one launch per tap; (64, 96): the three taps merged into
    # one contraction (dhw_gemm_desc.taps) and, for the weight gradient, into the batch index
    # (32, 32) at L = 4 / 2 / 1: samples shorter than a K step (the deepest level of an L = 8 .. 32 batch)
    # (5, 96, 64, 128), (6, 160, 128, 64): whole 64-row tiles and whole K steps — the unmasked K loops of the Conv1d forms, whose
    # only per-element test is the sample-edge one (first / last tile and K slice included), next to a ragged last tile
    for B, L, Cin, Cout in ((3, 20, 40, 72), (2, 24, 64, 96), (5, 4, 32, 32), (3, 2, 32, 64), (4, 1, 64, 32), (5, 96, 64, 128), (6, 160, 128, 64)):
        x = torch.randn(B * L, Cin, generator=g, requires_grad=True)
        W = torch.randn(Cout, Cin, 3, generator=g, requires_grad=True)
        b = torch.randn(Cout, generator=g, requires_grad=True)
        y, (vx, vW, vb), dy = _run(lambda t, x, W, b: t.conv3(x, W, b, L), (x, W, b))
        ref = F.conv1d(x.view(B, L, Cin).transpose(1, 2), W, b, padding="same").transpose(1, 2).reshape(B * L, Cout)
        ref.backward(dy)
        _close(y.d, ref)
        _close(vx.g, x.grad)
        _close(vW.g, W.grad)
        _close(vb.g, b.grad)

    x = torch.randn(50, 2, generator=g, requires_grad=True)                    # K = 2: input_dense
    W = torch.randn(130, 2, generator=g, requires_grad=True)
    b = torch.randn(130, generator=g, requires_grad=True)
    y, (vx, vW, vb), dy = _run(lambda t, x, W, b: t.linear(x, W, b), (x, W, b))
    ref = F.linear(x, W, b)
    ref.backward(dy)
    for mine, r in ((y.d, ref), (vx.g, x.grad), (vW.g, W.grad), (vb.g, b.grad)):
        _close(mine, r)


def test_attention_with_padding_mask():
    g = torch.Generator().manual_seed(2)
    B, H, D, Lq, Lk = 2, 3, 64, 24, 10
    q = torch.randn(B * Lq, H * D, generator=g, requires_grad=True)
    k = torch.randn(B * Lk, H * D, generator=g, requires_grad=True)
    v = torch.randn(B * Lk, H * D, generator=g, requires_grad=True)
    mask = torch.zeros(B, Lk)
    mask[0, 7:] = 1
    mask[1, 9:] = 1
    y, (vq, vk, vv), dy = _run(lambda t, q, k, v: t.attention(q, k, v, B, H, mask.to(DEV)), (q, k, v))
    split = lambda a, Ln: a.view(B, Ln, H, D).transpose(1, 2)   # noqa: E731
    ref = F.scaled_dot_product_attention(split(q, Lq), split(k, Lk), split(v, Lk), attn_mask=mask[:, None, None, :] * -1e9)
    ref = ref.transpose(1, 2).reshape(B * Lq, H * D)
    ref.backward(dy)
    for mine, r in ((y.d, ref), (vq.g, q.grad), (vk.g, k.grad), (vv.g, v.grad)):
        _close(mine, r)


def test_elementwise_norm_and_resampling_ops():
    g = torch.Generator().manual_seed(3)
    B, L, Cc = 2, 12, 96
    x = torch.randn(B * L, Cc, generator=g, requires_grad=True)
    gam = torch.randn(B, Cc, generator=g, requires_grad=True)
    bet = torch.randn(B, Cc, generator=g, requires_grad=True)
    pe = tm.positional_encoding(L, Cc, 4.0)
    keep = (torch.rand(B * L, Cc, generator=g) > 0.3).float()

    def build(t, x, gam, bet):
        h = t.film(t.layernorm(t.silu(x)), gam, bet, B)
        h = t.add(t.add_rows(h, pe.to(DEV), B), x)                      # fan-in on x
        h = t.dropout(h, keep.to(DEV), 0.3)
        h = t.resample(2, t.resample(0, h))                             # pool then upsample
        return t.sigmoid(h)

    y, (vx, vg, vb), dy = _run(build, (x, gam, bet))
    h = F.layer_norm(F.silu(x), (Cc,), eps=1e-6).view(B, L, Cc) * gam[:, None] + bet[:, None]
    h = (h + pe[None]).reshape(B * L, Cc) + x
    h = h * keep / 0.7
    h = F.avg_pool1d(h.view(B, L, Cc).transpose(1, 2), 2)
    h = F.interpolate(h, scale_factor=2, mode="nearest").transpose(1, 2).reshape(B * L, Cc)
    ref = torch.sigmoid(h)
    ref.backward(dy)
    for mine, r in ((y.d, ref), (vx.g, x.grad), (vg.g, gam.grad), (vb.g, bet.grad)):
        _close(mine, r)


def test_residual_adds_fused_into_gemm_film_and_layernorm_passes():
    """``addend`` of linear / conv3 (dhw_gemm_desc.addend), film_cols and ln_film_cols: y = op(x) + r, d r = d y, with fan-in on r
    (model.py:44-58 and cnn.py:87 chain exactly these), and ``silu_out`` (dhw_gemm_desc.act_out, act_out of dhw_op_ln_film): SiLU(y) as
    a second output of the same pass, ``pe`` of ln_film_cols: y + PE[l] as a third; sizes with edge tiles (scalar output path) and interior tiles (vector path)."""
    g = torch.Generator().manual_seed(11)
    for B, L, Cin, Cout in ((2, 24, 64, 96), (4, 64, 128, 128)):
        R = B * L
        x = torch.randn(R, Cin, generator=g, requires_grad=True)
        r = torch.randn(R, Cout, generator=g, requires_grad=True)
        W = torch.randn(Cout, Cin, generator=g, requires_grad=True)
        W3 = torch.randn(Cout, Cout, 3, generator=g, requires_grad=True)
        b = torch.randn(Cout, generator=g, requires_grad=True)
        b3 = torch.randn(Cout, generator=g, requires_grad=True)
        table = torch.randn(B, 4 * Cout, generator=g, requires_grad=True)     # gamma | beta for the FiLM, gamma | beta for LN + FiLM
        pe = tm.positional_encoding(L, Cout, 2.0)

        def build(t, x, r, W, W3, b, b3, table):
            h, ha = t.linear(x, W, b, addend=r, silu_out=True)                 # x W^T + b + r, and SiLU of it from the same pass
            h2, h2a = t.conv3(ha, W3, b3, L, addend=h, silu_out=True)          # both outputs of the Linear are consumed
            h3 = t.film_cols(h2a, table, 0, Cout, B, act=True, addend=r)       # second consumer of r: its gradient is a fan-in
            y, ya, yp = t.ln_film_cols(h3, table, 2 * Cout, 3 * Cout, B, addend=h2, silu_out=True, pe=pe.to(DEV))   # + SiLU(y), y + PE
            return t.add(t.add(y, ya), yp)

        y, vs, dy = _run(build, (x, r, W, W3, b, b3, table))
        h = F.linear(x, W, b) + r
        h2 = F.conv1d(F.silu(h).view(B, L, Cout).transpose(1, 2), W3, b3, padding="same").transpose(1, 2).reshape(R, Cout) + h
        ga, be, ga2, be2 = (table[:, i * Cout:(i + 1) * Cout] for i in range(4))
        h3 = F.silu(F.silu(h2).view(B, L, Cout) * ga[:, None] + be[:, None]).reshape(R, Cout) + r
        yr = (F.layer_norm(h3, (Cout,), eps=1e-6).view(B, L, Cout) * ga2[:, None] + be2[:, None]).reshape(R, Cout) + h2
        ref = yr + F.silu(yr) + (yr.view(B, L, Cout) + pe[None]).reshape(R, Cout)
        ref.backward(dy)
        _close(y.d, ref)
        for v, t_ref in zip(vs, (x, r, W, W3, b, b3, table)):
            _close(v.g, t_ref.grad)


def test_gemm_kernel_paths_against_float64_matmul():
    """dhw_op_gemm called directly over a sweep that reaches every path of the kernel: 64- and 32-row tiles, interior (unmasked K loop,
    16-byte output) and ragged tiles, K of whole and partial steps, the three operand orientations (k-major / m-major LDS tiles),
    split-K with atomics (accumulating outputs), addend, act_out, dsilu_of and rowsum."""
    import ctypes as C
    import itertools
    from dhg_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(5)
    for ((M, N, K), form), bf16 in itertools.product(itertools.product(((64, 64, 128), (200, 64, 50), (1920, 384, 96), (1920, 128, 480), (15360, 128, 128), (96, 160, 2), (128, 128, 3840)),
                                                                       ("AB", "ATB", "ABT")), (0, 1)):
        if bf16 and (M, N, K) in ((15360, 128, 128), (128, 128, 3840)) and form != "AB":
            continue   # (the mixed-precision mode — operands rounded to bf16 at staging, fp32 accumulation — on a subset: same paths)
        # reference for bf16 = 1: the same products of bf16-ROUNDED operands in float64 (what the kernel contracts), so only the fp32
        # accumulation order is left in the tolerance; the row sums (bias gradients) are sums of the rounded A, epilogues stay fp32
        rnd = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if bf16 else (lambda t: t)
        for acc in (0, 1):
            A = torch.randn(M, K, generator=g)
            Bm = torch.randn(K, N, generator=g)
            Cm = torch.randn(M, N, generator=g).to(DEV)
            D = torch.randn(M, N, generator=g)
            ref = rnd(A).double() @ rnd(Bm).double() + (Cm.cpu().double() if acc else 0)
            As, (sam, sak) = (A.t().contiguous(), (1, M)) if form == "ATB" else (A, (K, 1))
            Bs, (sbk, sbn) = (Bm.t().contiguous(), (1, K)) if form == "ABT" else (Bm, (N, 1))
            As, Bs, Dd = As.to(DEV), Bs.to(DEV), D.to(DEV)
            rs = torch.zeros(M, device=DEV)
            act = torch.empty(M, N, device=DEV) if not acc else None
            if not acc:
                ref = ref + D.double()                  # addend (non-accumulating launches); accumulating ones test dsilu_of instead:
            else:                                       # C += (A B) * SiLU'(D)
                sg = torch.sigmoid(D.double())
                ref = Cm.cpu().double() + (rnd(A).double() @ rnd(Bm).double()) * (sg * (1 + D.double() * (1 - sg)))
            d = _lib.GemmDesc(As.data_ptr(), sam, sak, 0, 0, 0, 0, Bs.data_ptr(), sbk, sbn, 0, 0, 0, 0, 0, Cm.data_ptr(), N, 1, 0, 0,
                              M, N, K, 1, 1, 0, 1, None, 1.0, acc, bf16)
            d.act_out = act.data_ptr() if act is not None else None
            d.addend = Dd.data_ptr() if not acc else None
            d.dsilu_of = Dd.data_ptr() if acc else None
            d.rowsum = rs.data_ptr()
            assert lib.dhw_op_gemm(C.byref(d), None) == 0
            torch.cuda.synchronize()
            tol = 2e-5 * max(float(ref.abs().max()), 1e-6)
            assert float((Cm.cpu().double() - ref).abs().max()) <= tol, (M, N, K, form, acc, bf16)
            rsum = rnd(A).double().sum(1)
            assert float((rs.cpu().double() - rsum).abs().max()) <= 2e-5 * max(float(rsum.abs().max()), 1.0), (M, N, K, form, acc, bf16)
            if act is not None:
                assert float((act.cpu().double() - F.silu(ref)).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0), (M, N, K, form, bf16)


def test_embedding_gather_and_scatter():
    g = torch.Generator().manual_seed(4)
    table = torch.randn(73, 48, generator=g, requires_grad=True)
    ids = torch.randint(0, 73, (2, 9), generator=g)
    y, (vt,), dy = _run(lambda t, table: t.embedding(ids.to(DEV).view(-1), table), (table,))
    ref = F.embedding(ids.view(-1), table)
    ref.backward(dy)
    _close(y.d, ref)
    _close(vt.g, table.grad)


@pytest.mark.parametrize("fixture", ["model_grad.npz", "model_grad_drop.npz"])
def test_whole_model_gradients_match_the_reference_autograd(golden_dir, fixture):
    """model_grad.npz: drop_rate 0.0 as configs/best.yml trains; model_grad_drop.npz: the class default 0.1 (model.py:71) with the
    twelve EncoderLayer.drop masks the reference drew replayed in call order."""
    f = np.load(os.path.join(golden_dir, fixture))
    B, L, Lt, S = (int(f[k]) for k in ("B", "L", "Lt", "S"))
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    names = [str(n) for n in f["names"]]
    assert names == [n for n in sd if n in set(names)] and len(names) == 323
    inp = spec.synthetic_inputs(B, L, Lt, S=S, seed=int(f["seed"]), pad=int(f["pad"]))
    keep = torch.from_numpy(np.unpackbits(f["keep"])[:B * S * 1280].reshape(B, S, 1280).astype(np.float32))
    eps, pen, alphas = (torch.from_numpy(f[k]) for k in ("eps", "pen", "alphas"))

    drop_rate, masks = 0.0, None
    if "drop_rate" in f.files:
        drop_rate = float(f["drop_rate"])
        bits, sizes = np.unpackbits(f["enc_keep"]), f["enc_keep_sizes"]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        masks = [torch.from_numpy(bits[offs[i]:offs[i + 1]].astype(np.float32)) for i in range(len(sizes))]
    model = tm.TrainModel({k: sd[k] for k in names}, num_layers=2, device=DEV, drop_rate=drop_rate)
    x_pert = train.perturb(torch.from_numpy(inp["strokes"]), eps, alphas)
    _close(x_pert, torch.from_numpy(f["x_pert"]), 1e-6)
    score, pen_pred = model.forward(x_pert, torch.from_numpy(inp["text"]), torch.sqrt(alphas), torch.from_numpy(inp["style"]), keep, masks)
    _close(score, torch.from_numpy(f["score"]), 5e-5)
    _close(pen_pred, torch.from_numpy(f["pen_pred"]), 5e-5)
    out, d_score, d_pen = train.loss_fn(eps, score, pen, pen_pred, alphas)
    assert np.allclose(out.cpu().numpy(), f["loss"], rtol=2e-5)
    model.backward(d_score, d_pen)
    torch.cuda.synchronize()

    # every parameter gradient: norm and projection on the fixture's random direction, relative to the gradient's own norm
    # (floored at 1e-4 of the largest: a few gradients — key biases under the softmax's shift invariance — are exactly 0)
    norms, dots = f["norms"], f["dots"]
    floor = 1e-4 * norms.max()
    worst = (0.0, "")
    for i, n in enumerate(names):
        gr = model.p[n].g
        assert gr is not None, f"no gradient reached {n}"
        gr = gr.cpu().double().numpy().ravel()
        scale = max(norms[i], floor)
        e_norm = abs(np.linalg.norm(gr) - norms[i]) / scale
        e_dot = abs(np.dot(gr, probe(i, gr.size).astype(np.float64)) - dots[i]) / scale
        worst = max(worst, (max(e_norm, e_dot), n))
        assert e_norm < 2e-3 and e_dot < 2e-3, (n, e_norm, e_dot, norms[i])
    print("worst parameter-gradient error (relative to its norm):", worst)
    for k in f.files:
        if k.startswith("g_"):
            _close(model.p[k[2:]].g, torch.from_numpy(f[k]), 1e-3)


@pytest.mark.parametrize("B", [4, 32])
def test_gradients_at_the_benchmarked_shape_match_the_oracle_autograd(B):
    """BASELINE configs[4] at the shape `bench.py --train` runs (L=480, Lt=50, S=14; B=32 is the bench's per-GPU batch itself — its
    GEMM shapes pick other tile sizes and split-K factors than B=4's): forward + loss + backward through the library against oracle/ref_cpu.forward +
    the reference's loss (loss.py:29-37) under torch autograd on the CPU, same seeded weights / inputs / eps / abar / style
    keep-mask.  The small fixtures (model_grad.npz: B=2, L=64, Lt=10) never reach the split-K weight-gradient path over
    thousands of rows with its fp32 atomics, nor the multi-block attention of L/2 = 240 keys; this does.  Every one of the
    323 parameter gradients: norm and projection on a fixed random direction, relative to the gradient's own norm
    (floored at 1e-4 of the largest).  Tolerance 2e-4 (fp32 accumulation order over 1920-row contractions; measured 6.6e-6)."""
    from oracle import ref_cpu
    L, Lt, S = 480, 50, 14
    sd_np = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    inp = spec.synthetic_inputs(B, L, Lt, S=S, seed=21, pad=3)
    g = torch.Generator().manual_seed(5)
    eps = torch.randn(B, L, 2, generator=g)
    pen = (torch.rand(B, L, generator=g) < 0.1).float()
    alphas = torch.rand(B, 1, generator=g) * 0.9 + 0.05
    keep = (torch.rand(B, S, 1280, generator=g) >= 0.3).float()
    strokes, text, style = (torch.from_numpy(inp[k]) for k in ("strokes", "text", "style"))

    # --- oracle: CPU restatement of the reference under autograd (Dropout(0.3) on the style input = keep / 0.7, text_style.py:83,92)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd_np.items()}
    x_pert = ref_cpu.perturb(strokes, eps, alphas)
    score_ref, pen_ref = ref_cpu.forward(sd, x_pert, text, torch.sqrt(alphas), style * keep / 0.7)
    loss_ref, sl_ref, pl_ref = ref_cpu.loss_fn(eps, score_ref, pen, pen_ref, alphas)
    loss_ref.backward()

    # --- library
    model = tm.TrainModel(sd_np, num_layers=2, device=DEV, drop_rate=0.0)
    xp = train.perturb(strokes, eps, alphas)
    _close(xp, x_pert, 1e-6)
    score, pen_pred = model.forward(xp, text, torch.sqrt(alphas), style, keep, None)
    _close(score, score_ref, 1e-4)
    _close(pen_pred, pen_ref, 1e-4)
    out, d_score, d_pen = train.loss_fn(eps, score, pen, pen_pred, alphas)
    assert np.allclose(out.cpu().numpy(), [loss_ref.item(), sl_ref.item(), pl_ref.item()], rtol=5e-5), (out, loss_ref)
    model.backward(d_score, d_pen)
    torch.cuda.synchronize()

    names = list(model.names)
    assert len(names) == 323
    ref_norms = np.array([float(sd[n].grad.double().norm()) for n in names])
    floor = 1e-4 * ref_norms.max()
    worst = (0.0, "")
    for i, n in enumerate(names):
        gr = model.p[n].g
        assert gr is not None, f"no gradient reached {n}"
        gr = gr.cpu().double().numpy().ravel()
        rf = sd[n].grad.double().numpy().ravel()
        pr = probe(i, gr.size).astype(np.float64)
        scale = max(ref_norms[i], floor)
        e_norm = abs(np.linalg.norm(gr) - ref_norms[i]) / scale
        e_dot = abs(np.dot(gr, pr) - np.dot(rf, pr)) / scale
        e_max = np.abs(gr - rf).max() / scale
        worst = max(worst, (max(e_norm, e_dot), n))
        assert e_norm < 2e-4 and e_dot < 2e-4 and e_max < 5e-3, (n, e_norm, e_dot, e_max, ref_norms[i])
    print("configs[4] shape: worst parameter-gradient error (relative to its norm):", worst)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_train_steps_follow_the_reference_trajectory(golden_dir, mode):
    """tests/golden/train_traj.npz: four updates of the reference's train_step (train.py:26-67) with its optimizer stack
    (Adam + InvSqrtScheduledOptim + clip_grad_norm_(100), configs/best.yml) on one fixed batch.  The native step — perturb,
    forward, loss, backward, clip, Adam at the Noam rate — must reproduce each step's three losses, the unclipped gradient
    norm, and every parameter's displacement after the last update."""
    f = np.load(os.path.join(golden_dir, "train_traj.npz"))
    B, L, Lt, S, steps = (int(f[k]) for k in ("B", "L", "Lt", "S", "steps"))
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    model = tm.TrainModel(sd, num_layers=2, device=DEV)
    opt = train.Adam(model.parameters())
    inp = spec.synthetic_inputs(B, L, Lt, S=S, seed=int(f["seed"]), pad=int(f["pad"]))
    strokes3 = torch.cat([torch.from_numpy(inp["strokes"]), torch.from_numpy(f["pen"])[..., None]], dim=-1)
    batch = {"strokes": strokes3, "text": torch.from_numpy(inp["text"]), "style": torch.from_numpy(inp["style"])}
    eps, alphas = torch.from_numpy(f["eps"]), torch.from_numpy(f["alphas"])
    keeps = np.unpackbits(f["keep"])[:steps * B * S * 1280].reshape(steps, B, S, 1280).astype(np.float32)
    losses = []
    graphed = tm.GraphedTrainStep(model, opt, B, L, Lt, S, warmup=int(f["warmup"]), device_rng=False) if mode == "graph" else None
    for step in range(1, steps + 1):
        keep = torch.from_numpy(keeps[step - 1])
        if graphed is not None:      # the hipGraph is captured at step 1 and replayed with new inputs / learning rate afterwards
            out = graphed(batch, None, step, eps=eps, alphas=alphas, style_keep=keep)
            norm = graphed.grad_norm()
        else:
            out = tm.train_step(model, opt, batch, None, step, eps=eps, alphas=alphas, style_keep=keep, warmup=int(f["warmup"]))
            norm = model.last_grad_norm
        losses.append(out.cpu().numpy())
        assert abs(norm - f["grad_norms"][step - 1]) < 2e-3 * f["grad_norms"][step - 1], (step, norm)
    losses = np.array(losses)
    print("losses:", losses[:, 0], "reference:", f["losses"][:, 0])
    assert np.allclose(losses, f["losses"], rtol=2e-3), (losses, f["losses"])
    delta = np.array([float((model.p[n].d.cpu() - torch.from_numpy(sd[n])).norm()) for n in model.names])
    assert np.allclose(delta, f["delta"], rtol=2e-2, atol=1e-6), np.abs(delta / np.maximum(f["delta"], 1e-12) - 1).max()


def test_device_draws_are_standard_normal_and_drop_30_percent():
    """dhw_train_draw: eps ~ N(0,1), keep-mask mean 0.7, a new draw index gives new numbers, the same index the same."""
    import ctypes as C
    from dhg_amd import _lib
    B, L, n_keep = 16, 480, 16 * 14 * 1280
    eps, keep = torch.empty(B, L, 2, device=DEV), torch.empty(n_keep, device=DEV)
    rng = torch.tensor([7, 3], dtype=torch.int64, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def draw():
        assert _lib.lib().dhw_train_draw(rng.data_ptr(), B, L, eps.data_ptr(), n_keep, n_keep // B, 0.3, keep.data_ptr(), st) == 0
        return eps.cpu().clone(), keep.cpu().clone()
    e1, k1 = draw()
    e1b, k1b = draw()
    rng[1] = 4
    e2, k2 = draw()
    assert torch.equal(e1, e1b) and torch.equal(k1, k1b) and not torch.equal(e1, e2) and not torch.equal(k1, k2)
    n = e1.numel()
    assert abs(float(e1.mean())) < 4 / n ** 0.5 and abs(float(e1.var()) - 1) < 6 * (2 / n) ** 0.5
    assert set(k1.unique().tolist()) == {0.0, 1.0} and abs(float(k1.mean()) - 0.7) < 4 * (0.21 / n_keep) ** 0.5
    assert abs(float((k1 * k2).mean()) - 0.49) < 5e-3        # independent across draws


def test_two_rank_update_equals_the_single_rank_update_on_the_whole_batch(tmp_path):
    """Data parallelism of the training step (BASELINE configs[4]): two ranks, each with half of a seeded batch, averaging their
    flat gradient buffers (gloo here: both ranks share the box's one GPU; RCCL on a real node), must land on the parameters one
    rank reaches with the whole batch — the losses are batch means, so the mean of the two ranks' gradients IS the whole-batch
    gradient.  The pair runs the BUCKETED path (five graph segments, each bucket's all-reduce issued behind its segment:
    train.GradBucketReducer); a second pair with DHW_TRAIN_BUCKETS=0 runs the single flat all-reduce and must agree with it."""
    import socket
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "ddp_worker.py")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    single = subprocess.run([sys.executable, worker, "0", "1", port, str(tmp_path / "w1.npz")], env=env, timeout=300)
    assert single.returncode == 0
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path / "w2.npz")], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port2 = str(s.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port2, str(tmp_path / "w2flat.npz")], env=dict(env, DHW_TRAIN_BUCKETS="0")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a, b, bf = np.load(tmp_path / "w1.npz"), np.load(tmp_path / "w2.npz"), np.load(tmp_path / "w2flat.npz")
    assert int(a["segments"]) == 0 and int(b["segments"]) == 5 and int(bf["segments"]) == 0     # one graph / five segments / one graph + flat all-reduce
    # the same sums, bucket by bucket or in one piece (two runs differ at rounding level: fp32 atomics in the split-K weight gradients)
    d2 = np.linalg.norm((b["flat"] - bf["flat"]).astype(np.float64))
    m2 = np.linalg.norm((bf["flat"] - spec_flat()).astype(np.float64))
    assert m2 > 1e-3 and d2 < 1e-3 * m2, ("bucketed and flat gradient all-reduce disagree", d2, m2)
    # rank 0 of the pair reports the loss of ITS half; the parameters are what must agree
    assert np.isfinite(b["losses"]).all()
    # Adam's early updates are ~ lr * g / (|g| + 1e-8): elements whose gradient is at rounding level move by a
    # summation-order-dependent amount, so the comparison is in the L2 norm of the whole displacement
    diff = np.linalg.norm((a["flat"] - b["flat"]).astype(np.float64))
    moved = np.linalg.norm((a["flat"] - spec_flat()).astype(np.float64))
    print("parameter difference 2 ranks vs 1 (L2):", diff, "displacement (L2):", moved)
    assert moved > 1e-3 and diff < 1e-3 * moved


def spec_flat():
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    return np.concatenate([np.asarray(v, np.float32).ravel() for v in sd.values()])


def test_gradient_buckets_are_final_when_their_marker_fires():
    """The overlapped all-reduce (train.GradBucketReducer) reads bucket i of the flat gradient buffer as soon as the tape's marker
    i fires.  That is only right if no later backward kernel adds into that range: snapshot every bucket at its marker and compare
    with the buffer after the whole sweep, bit for bit — and the buckets must tile the buffer, every tensor in its rule's bucket."""
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    B, L, Lt = 2, 64, 10
    inp = spec.synthetic_inputs(B, L, Lt, S=14, seed=9, pad=2)
    model = tm.TrainModel(sd, num_layers=2, device="cuda", drop_rate=0.1)
    assert len(model.bucket_ranges) == tm.N_BUCKETS and model.bucket_ranges[0][0] == 0 and model.bucket_ranges[-1][1] == model.flat.numel()
    for k in model.names:
        a, b = model.bucket_ranges[tm.grad_bucket(k)]
        assert a <= model.offset[k] and model.offset[k] + model.p[k].d.numel() <= b, k
    g = torch.Generator().manual_seed(9)
    x = torch.from_numpy(inp["strokes"])
    snaps, order = {}, []

    def hook(i):
        torch.cuda.synchronize()
        a, b = model.bucket_ranges[i]
        snaps[i] = model.flat_grad[a:b].clone()
        order.append(i)
    for trial in range(2):          # (twice: the second sweep runs on a buffer that held the first one's gradients)
        snaps.clear(), order.clear()
        model.zero_grad()
        score, pen = model.forward(x, torch.from_numpy(inp["text"]), torch.rand(B, 1, generator=g) * 0.8 + 0.1, torch.from_numpy(inp["style"]))
        model.bucket_hook = hook
        model.backward(torch.randn(score.shape, generator=g).cuda(), torch.randn(pen.shape, generator=g).cuda())
        model.bucket_hook = None
        torch.cuda.synchronize()
        assert order == [0, 1, 2, 3]
        for i, snap in snaps.items():
            a, b = model.bucket_ranges[i]
            assert torch.equal(snap, model.flat_grad[a:b]), f"bucket {i} received gradient after its marker (trial {trial})"
            assert float(snap.abs().sum()) > 0
        a, b = model.bucket_ranges[-1]
        assert float(model.flat_grad[a:b].abs().sum()) > 0


def test_device_drawn_encoder_dropout_trains():
    """drop_rate 0.1 with the masks drawn by dhw_op_keep_mask inside a graph-replayed step: the masks change from update to
    update (two replays with the same inputs but different draw indices give different losses), the same index reproduces."""
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    B, L, Lt = 2, 64, 10
    inp = spec.synthetic_inputs(B, L, Lt, S=14, seed=5, pad=1)
    g = torch.Generator().manual_seed(5)
    batch = {"strokes": torch.cat([torch.from_numpy(inp["strokes"]), (torch.rand(B, L, 1, generator=g) < 0.1).float()], dim=-1),
             "text": torch.from_numpy(inp["text"]), "style": torch.from_numpy(inp["style"])}
    eps, alphas = torch.randn(B, L, 2, generator=g), torch.rand(B, 1, generator=g) * 0.9 + 0.05
    keep = (torch.rand(B, 14, 1280, generator=g) >= 0.3).float()

    def first_losses(indices):
        model = tm.TrainModel(sd, num_layers=2, device=DEV, drop_rate=0.1)
        opt = train.Adam(model.parameters())
        step = tm.GraphedTrainStep(model, opt, B, L, Lt, warmup=10 ** 9, device_rng=False)     # lr ~ 0: the weights stay put
        return [float(step(batch, None, k, eps=eps, alphas=alphas, style_keep=keep)[0]) for k in indices]
    a = first_losses([1, 2, 1])
    b = first_losses([1])
    # (the loss sums are fp32 atomics over blocks: the same masks give the same loss up to summation order, not bit for bit)
    assert abs(a[0] - b[0]) < 2e-6 * abs(a[0]) and abs(a[0] - a[1]) > 1e-4 and abs(a[0] - a[2]) < 1e-4, (a, b)


def test_fit_trains_and_its_checkpoint_feeds_the_sampler(tmp_path):
    """train.py / fit(): the reference's loop shape (train.py:84-134) on synthetic batches — log lines in its format,
    checkpoint_<n>.pth and model_final.pth as torch-saved state_dicts with the reference's keys, which the sampling side
    (load_model -> sample) takes as they are."""
    cfg = tmp_path / "cfg.yml"
    cfg.write_text("""
experiment: {seed: 1}
dataset_args: {max_seq_len: 64, max_text_len: 8}
training_args: {steps: 4, batch_size: 2, warmup_steps: 100, clip_grad: 100.0, dropout: 0.1, att_layers_num: 2, channels: 128, log_freq: 2, save_freq: 3}
optimizer: {type: torch.optim.Adam, params: {lr: 0.0003, weight_decay: 0.00001, betas: [0.9, 0.98]}}
""")
    lines = []
    tm.fit(cfg, None, tmp_path / "run", log=lines.append)
    assert len(lines) == 2 and lines[0].startswith("Step 2 | Loss: ") and " | Score: " in lines[1] and " | Pen: " in lines[1]
    ck = torch.load(tmp_path / "run" / "checkpoint_3.pth", weights_only=True)
    assert set(ck) == {"meta", "state_dict"} and len(ck["state_dict"]) == 323            # save_checkpoint's form (checkpoint.py:244)
    assert dhg_amd.find_checkpoint(tmp_path / "run").name == "model_final.pth"
    sd = torch.load(tmp_path / "run" / "model_final.pth", weights_only=True)
    ref_keys = list(spec.synthetic_state_dict(2))
    assert list(sd) == ref_keys and all(torch.isfinite(v).all() for v in sd.values())
    moved = max(float((sd[k] - torch.from_numpy(spec.synthetic_state_dict(2, seed=0)[k])).abs().max()) for k in ref_keys)
    assert moved > 0
    model = dhg_amd.load_model(cfg, tmp_path / "run" / "model_final.pth")
    inp = spec.synthetic_inputs(2, 64, 8, seed=2)
    out = dhg_amd.sample(model, torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda(), L=64, T=2, seed=3)
    assert out.shape == (2, 64, 3) and bool(torch.isfinite(out).all())


@pytest.mark.parametrize("num_layers,B,L,Lt,S", [(2, 3, 72, 7, 14), (4, 2, 40, 12, 14), (2, 1, 8, 1, 1)])
def test_training_forward_equals_the_inference_forward_when_nothing_is_dropped(num_layers, B, L, Lt, S):
    """The two native forwards of the same model — the training one (generic fp32 ops, torch layouts) and the sampler's fused
    fp32 kernels (packed weights; pinned to the reference by the goldens) — on shapes the gradient fixtures do not cover:
    num_layers = 4 (the class default, model.py:66), odd token counts, the smallest legal L, a one-row style.  A keep-mask of
    0.7 everywhere makes Dropout(0.3) the identity (0.7 / (1 - 0.3) = 1)."""
    sd = spec.synthetic_state_dict(num_layers, 128, 192, 256, seed=4)
    inp = spec.synthetic_inputs(B, L, Lt, S=S, seed=8, pad=min(2, Lt - 1))
    sigma = torch.linspace(0.2, 0.9, B).reshape(B, 1)
    strokes, text, style = (torch.from_numpy(inp[k]) for k in ("strokes", "text", "style"))
    infer = dhg_amd.DiffusionModel(num_layers, precision="fp32").eval()
    infer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    eps_ref, pen_ref, _ = infer(strokes.cuda(), text.cuda(), sigma.cuda(), style.cuda())
    model = tm.TrainModel(sd, num_layers=num_layers, device=DEV)
    score, pen = model.forward(strokes, text, sigma, style, torch.full((B, S, 1280), 0.7))
    _close(score, eps_ref, 2e-5)
    _close(pen, pen_ref, 2e-5)


def test_bf16_mixed_precision_gradients_stay_close_and_training_still_learns(golden_dir):
    """precision="bf16": GEMM operands rounded to bf16 in the kernel, fp32 accumulation and fp32 everything else.  Against the
    reference's fp32 gradients (model_grad.npz): the whole gradient vector within 2 % in L2, the loss within 1e-3; and ten
    updates on one fixed batch at a real learning rate bring the loss down like the fp32 step does."""
    f = np.load(os.path.join(golden_dir, "model_grad.npz"))
    B, L, Lt, S = (int(f[k]) for k in ("B", "L", "Lt", "S"))
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    names = [str(n) for n in f["names"]]
    inp = spec.synthetic_inputs(B, L, Lt, S=S, seed=int(f["seed"]), pad=int(f["pad"]))
    keep = torch.from_numpy(np.unpackbits(f["keep"])[:B * S * 1280].reshape(B, S, 1280).astype(np.float32))
    eps, pen, alphas = (torch.from_numpy(f[k]) for k in ("eps", "pen", "alphas"))
    grads = {}
    for prec in ("fp32", "bf16"):
        model = tm.TrainModel({k: sd[k] for k in names}, num_layers=2, device=DEV, precision=prec)
        x_pert = train.perturb(torch.from_numpy(inp["strokes"]), eps, alphas)
        score, pen_pred = model.forward(x_pert, torch.from_numpy(inp["text"]), torch.sqrt(alphas), torch.from_numpy(inp["style"]), keep)
        out, d_score, d_pen = train.loss_fn(eps, score, pen, pen_pred, alphas)
        model.backward(d_score, d_pen)
        grads[prec] = (model.flat_grad.cpu().double().numpy().copy(), float(out[0]))
    rel = np.linalg.norm(grads["bf16"][0] - grads["fp32"][0]) / np.linalg.norm(grads["fp32"][0])
    print("bf16 vs fp32 gradient, relative L2:", rel, "loss", grads["bf16"][1], grads["fp32"][1])
    assert rel < 2e-2 and abs(grads["bf16"][1] - grads["fp32"][1]) < 1e-3 * abs(grads["fp32"][1]) + 1e-3

    g = torch.Generator().manual_seed(5)
    batch = {"strokes": torch.cat([torch.from_numpy(inp["strokes"]), pen[..., None]], dim=-1), "text": torch.from_numpy(inp["text"]),
             "style": torch.from_numpy(inp["style"])}
    finals = {}
    for prec in ("fp32", "bf16"):
        model = tm.TrainModel(sd, num_layers=2, device=DEV, precision=prec)
        opt = train.Adam(model.parameters())
        losses = [float(tm.train_step(model, opt, batch, None, k, eps=eps, alphas=alphas, style_keep=keep, warmup=2000)[0]) for k in range(1, 41)]
        finals[prec] = (losses[0], losses[-1])
    print("loss first -> last after 40 updates:", finals)
    assert finals["bf16"][1] < 0.8 * finals["bf16"][0] and abs(finals["bf16"][1] - finals["fp32"][1]) < 0.15 * finals["fp32"][0]


_SWITCH_WORKER = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from dhg_amd import spec, train, train_model as tm
B, L, Lt = 4, 64, 10
sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
model = tm.TrainModel(sd, num_layers=2, device=torch.device("cuda", 0))
inp = spec.synthetic_inputs(B, L, Lt, S=14, seed=5, pad=3)
g = torch.Generator().manual_seed(5)
x = torch.from_numpy(inp["strokes"]); sig = torch.linspace(0.2, 0.9, B).reshape(B, 1)
keep = (torch.rand(B, 14, 1280, generator=g) >= 0.3).float()
score, pen = model.forward(x, torch.from_numpy(inp["text"]), sig, torch.from_numpy(inp["style"]), style_keep=keep)
w1 = torch.randn(score.shape, generator=g).cuda(); w2 = torch.randn(pen.shape, generator=g).cuda()
model.zero_grad(); model.backward(w1, w2)
torch.cuda.synchronize()
print(json.dumps({"launches": model.last_launches, "score": float(score.double().abs().sum()), "gnorm": float(model.flat_grad.double().norm()),
                  "probe": model.flat_grad[::9973].double().cpu().tolist()}))
"""


def test_training_launch_switches_give_the_same_gradients(tmp_path):
    """The round-4 launch mergers (a layer's weight + data gradient in one launch, grouped q / k / v, FiLM riding on the ConvBlock
    GEMMs) and the 16-byte element-wise kernels only change WHICH launch computes a value: with each switch off the forward output and
    the whole gradient vector agree to rounding (the 16-byte forms associate LayerNorm's sums differently: 1e-5 relative), and the
    default issues fewer launches than every switched-off run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, env in (("default", {}), ("no_pair", {"DHW_SGEMM_PAIR": "0"}), ("no_group", {"DHW_SGEMM_GROUP": "0"}), ("no_vec4", {"DHW_TRAIN_VEC4": "0"}),
                      ("no_rider", {"DHW_TRAIN_FILM_RIDER": "0"})):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", _SWITCH_WORKER, root], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    ref = runs["default"]
    for name, v in runs.items():
        assert abs(v["score"] - ref["score"]) <= 1e-5 * abs(ref["score"]), name
        assert abs(v["gnorm"] - ref["gnorm"]) <= 1e-5 * ref["gnorm"], name
        assert np.allclose(v["probe"], ref["probe"], rtol=2e-4, atol=1e-6 * ref["gnorm"]), name
    assert all(runs[k]["launches"] > ref["launches"] for k in ("no_pair", "no_group", "no_rider")), {k: v["launches"] for k, v in runs.items()}
