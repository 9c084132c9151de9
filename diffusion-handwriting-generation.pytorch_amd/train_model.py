"""The whole training step of ``DiffusionModel`` on the MI355X (BASELINE configs[4]).

``TrainModel`` runs the reference's ``DiffusionModel.forward`` in train mode (model.py:133-199) and the backward pass its
``loss.backward()`` would run (train.py:55-60), as a chain of hand-written fp32 HIP operations (include/dhw_train.h
``dhw_op_*``: one strided exact-f32 MFMA GEMM for every Linear / Conv1d / attention product in all three directions, plus
FiLM, LayerNorm, softmax, resampling, embedding and activation kernels).  torch supplies device memory and streams only:
no torch operator touches an activation or a gradient.  The op order mirrors the reference module by module so that the
recorded tape, replayed in reverse, accumulates exactly autograd's gradients; ``tests/test_gpu_train.py`` pins every one
of the 323 parameter gradients against a fixture the imported reference generated (tests/golden/model_grad.npz).

Parameters keep the reference's ``state_dict`` names and torch layouts, so a reference checkpoint loads unchanged and a
trained one can be handed straight to ``dhg_amd.DiffusionModel`` for sampling.
"""
from __future__ import annotations

import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib
from .train import Adam, GradBucketReducer, _stream, _tcheck, allreduce_grads, get_alphas, loss_fn, noam_lr, perturb

_F = 4  # bytes per element


# (The weight-gradient GEMMs run on the main stream.  A second stream beside the data-gradient chain — rounds 2 and 3,
# DHW_TRAIN_WGRAD_SIDE — measured slower (17.3 vs 15.1, later 9.93 vs 9.12 ms per update: the GEMMs already fill the CUs, parallel
# graph branches only interleave them) and, once gradient buffers were handed from an op's output to its addend, raced with the
# main stream's writes into those buffers; the switch and the side stream were removed in round 4.)
# DHW_TRAIN_FUSE_DSILU=0: SiLU's backward as its own pass again instead of a factor in the consuming GEMM's data-gradient output (A/B)
FUSE_DSILU = os.environ.get("DHW_TRAIN_FUSE_DSILU", "1") != "0"
# DHW_TRAIN_FILM_RIDER=0: a ConvBlock's affine (+ SiLU) (+ conv_skip) as its own pass behind the GEMM again instead of a further output of the GEMM (A/B)
FILM_RIDER = os.environ.get("DHW_TRAIN_FILM_RIDER", "1") != "0"
# DHW_TRAIN_BUCKETS=0: one flat SUM all-reduce behind the whole backward (rounds 2-4) instead of bucketed all-reduces overlapped with its tail (A/B)
GRAD_BUCKETS = os.environ.get("DHW_TRAIN_BUCKETS", "1") != "0"

# Gradient buckets, in the order the backward sweep completes them (TrainModel lays the flat buffers out in this order, so a
# bucket is ONE contiguous range = one all-reduce):
#   0  decoder ConvBlocks + skip convolutions          final once the sweep is back at the end of the bottleneck layers
#   1  bottleneck EncoderLayers + att_dense            ... at the end of enc5
#   2  encoder ConvBlocks / EncoderLayers + input_dense ... in front of input_dense
#   3  text side (TextStyleEncoder, every layer's text_dense: evaluated once, first, for all layers) + all FiLM Linears (their
#      gradient is ONE kernel over the whole [B, 2 x 9280] table, the last but two of the sweep)
#   4  sigma MLP (the first module of the forward) + the two heads' odd-sized tensors (kept last so that every other tensor starts
#      16-byte aligned: the GEMMs take the 16-byte-load forms and film_table reads weight rows as f32x4)
N_BUCKETS = 5


def grad_bucket(name: str) -> int:
    if name.startswith(("sigma_ffn.", "output_dense.", "pen_lifts_dense.")):
        return 4
    if ".gamma_emb." in name or ".beta_emb." in name or name.startswith("text_style_model.") or ".text_dense." in name:
        return 3
    if name.startswith(("dec1.", "dec2.", "dec3.", "skip_conv")):
        return 0
    if name.startswith(("att_layers.", "att_dense.")):
        return 1
    if name.startswith(("enc1.", "enc2.", "enc3.", "enc4.", "enc5.", "input_dense.")):
        return 2
    raise ValueError(f"no gradient bucket rule for parameter {name}")

class Var:
    """A node of the tape: a device tensor and its lazily allocated (zero-initialised) gradient.  ``leaf``: a network input
    whose gradient nobody needs (the backward sweep skips the kernels that would only produce it)."""
    __slots__ = ("d", "g", "leaf", "pre")

    def __init__(self, d: torch.Tensor, leaf: bool = False):
        self.d = d
        self.g = None
        self.leaf = leaf
        self.pre = None     # self = SiLU(pre): a GEMM that consumes self sends its data gradient straight to pre (dhw_gemm_desc.dsilu_of)

    def grad(self) -> torch.Tensor:
        if self.g is None:
            self.g = torch.zeros_like(self.d)
        return self.g


def positional_encoding(length: int, dim: int, pos_factor: float) -> torch.Tensor:
    """PosEmbeddings.forward(arange(length)) (attention.py:14-23) -> [length, dim] on the host (a constant table)."""
    half = dim // 2
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = torch.arange(length)[:, None] * freq[None, :] * pos_factor
    return torch.cat((e.sin(), e.cos()), dim=-1).float()


class Tape:
    """Forward ops append their backward closure; ``backward()`` replays them in reverse.  Every backward kernel ADDS into
    the input's gradient buffer (zero-initialised on first touch), which is how autograd's fan-in sums arise."""

    def __init__(self, device, bf16: bool = False):
        self.dev = device
        self.bf16 = int(bf16)   # GEMM operands rounded to bf16 inside the kernel (fp32 accumulation); everything else stays fp32
        self.lib = _lib.lib()
        self.main = torch.cuda.current_stream(device)
        self.st = _stream(device)
        self.steps = []
        self.launches = 0
        self.flops = 0          # algorithmic FLOPs of every GEMM issued (2 M N K per batch entry)

    # ---- raw kernel wrappers ---------------------------------------------------------------------------------------
    def new(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, device=self.dev, dtype=torch.float32)

    def gemm(self, *args, **kw):
        """One GEMM launch; arguments as ``gemm_desc``."""
        d = self.gemm_desc(*args, **kw)
        _tcheck(self.lib.dhw_op_gemm(C.byref(d), self.st))
        self.launches += 1

    def gemm_pair(self, first, second):
        """Two independent GEMMs — ``first`` / ``second``: (args, kwargs) of ``gemm_desc``, or None — as ONE launch where the
        library can pair them (dhw_op_gemm2: a layer's weight gradient, first, with its data gradient)."""
        if first is None or second is None:
            one = first if second is None else second
            return self.gemm(*one[0], **one[1])
        d0, d1 = self.gemm_desc(*first[0], **first[1]), self.gemm_desc(*second[0], **second[1])
        n = self.lib.dhw_op_gemm2(C.byref(d0), C.byref(d1), self.st)     # (returns the launches it issued: 1, or 2 where the forms do not pair)
        _tcheck(n)
        self.launches += n

    def gemm_group(self, items):
        """Up to six independent GEMMs — ``items``: (args, kwargs) of ``gemm_desc`` (None entries are skipped), dispatched in that
        order — as one launch where the library can group them (dhw_op_gemm_group)."""
        items = [it for it in items if it is not None]
        for i in range(0, len(items), 6):
            chunk = items[i:i + 6]
            if len(chunk) == 1:
                self.gemm(*chunk[0][0], **chunk[0][1])
                continue
            descs = [self.gemm_desc(*a, **k) for a, k in chunk]
            arr = (_lib.GemmDesc * len(descs))(*descs)
            n = self.lib.dhw_op_gemm_group(arr, len(descs), self.st)
            _tcheck(n)
            self.launches += n

    def gemm_desc(self, A, a_off, sam, sak, Bm, b_off, sbk, sbn, Cm, c_off, scm, scn, M, N, K, *, bias=None, alpha=1.0, acc=False,
                  nzo=1, nzi=1, za=(0, 0), zb=(0, 0), zc=(0, 0), a_shift=0, b_shift=0, lr=0, taps=1, a_tap_shift=0, sbt=0, b_z_shift=0,
                  rowsum=None, addend=None, act_out=None, dsilu_of=None, film=None):
        """See dhw_gemm_desc (include/dhw_train.h).  Extents are checked here, on the host, before the launch.  ``film``: (gamma
        pointer, beta pointer, table row stride, rows per sample, act, out tensor, addend tensor or None) — FiLM (+ SiLU) (+ addend)
        of the written value as a further output of the pass."""
        Kt = K // taps

        def extent(off, s0, n0, s1, n1, z, tap_stride=0):
            return off + (n0 - 1) * s0 + (n1 - 1) * s1 + (nzo - 1) * z[0] + (nzi - 1) * z[1] + (taps - 1) * tap_stride
        if extent(a_off, sam, M, sak, Kt, za) >= A.numel() or extent(b_off, sbk, Kt, sbn, N, zb, sbt) >= Bm.numel() \
                or extent(c_off, scm, M, scn, N, zc) >= Cm.numel() or min(a_off, b_off, c_off) < 0:
            raise ValueError("gemm operand extents exceed their buffers")
        if max(A.numel(), Bm.numel(), Cm.numel()) >= 2 ** 31:
            raise ValueError("gemm operands are addressed with 32-bit element offsets")
        if (a_shift or a_tap_shift) and (lr < 1 or M % lr):
            raise ValueError("a row-shifted gemm needs whole samples of lr rows")
        if (b_shift or b_z_shift) and (lr < 1 or Kt % lr):
            raise ValueError("a row-shifted gemm needs whole samples of lr rows")
        d = _lib.GemmDesc(A.data_ptr() + a_off * _F, sam, sak, za[0], za[1], a_shift, a_tap_shift,
                          Bm.data_ptr() + b_off * _F, sbk, sbn, zb[0], zb[1], sbt, b_shift, b_z_shift,
                          Cm.data_ptr() + c_off * _F, scm, scn, zc[0], zc[1],
                          M, N, K, nzo, nzi, lr, taps, bias.data_ptr() if bias is not None else None, alpha, int(acc), self.bf16,
                          act_out.data_ptr() if act_out is not None else None, addend.data_ptr() if addend is not None else None,
                          dsilu_of.data_ptr() if dsilu_of is not None else None, rowsum.data_ptr() if rowsum is not None else None)
        if film is not None:
            fg, fb, fps, frows, fact, fout, fadd = film
            if fout.numel() != Cm.numel() or scn != 1 or nzo * nzi != 1 or acc:
                raise ValueError("a FiLM output rides on an unbatched row-major GEMM without accumulation")
            d.film_gamma, d.film_beta, d.film_pstride, d.film_rows, d.film_act = fg, fb, fps, frows, int(fact)
            d.film_out, d.film_addend = fout.data_ptr(), fadd.data_ptr() if fadd is not None else None
        d._keep = (A, Bm, Cm, bias, act_out, addend, dsilu_of, rowsum, film)   # (the operands outlive the descriptor's use)
        self.flops += 2 * M * N * K * nzo * nzi
        return d

    def into(self, v: "Var"):
        """(gradient buffer of v, accumulate flag): the first writer overwrites a fresh buffer, later ones add."""
        if v.g is not None:
            return v.g, 1
        if isinstance(v, _ViewVar):
            base, _ = self.into(v.base)
            v.g = base.view(*v.shape)
        else:
            v.g = torch.empty_like(v.d)
        return v.g, 0

    def call(self, fn, *args):
        _tcheck(getattr(self.lib, fn)(*args, self.st))
        self.launches += 1

    def _grad_to(self, v: "Var", dy: torch.Tensor):
        """d v (+)= dy for an addend v of an op whose output gradient buffer dy is dead afterwards: v takes the buffer if it has
        no gradient yet (later fan-in adds go into it), else one add."""
        if v.leaf:
            return
        if v.g is None and not isinstance(v, _ViewVar):
            v.g = dy
            return
        dv, acc = self.into(v)
        self.call("dhw_op_add", dy.data_ptr(), None, dy.numel(), dv.data_ptr(), acc)

    def _silu_of(self, y: "Var", buf: torch.Tensor) -> "Var":
        """The Var of SiLU(y) whose values the kernel that produced y wrote into ``buf`` in the same pass (``silu_out`` of linear /
        conv3 / ln_film_cols); its backward is the stand-alone SiLU's (recorded after y's, so it runs first and adds into y.g)."""
        ya = Var(buf)
        ya.pre = y if FUSE_DSILU else None
        n = buf.numel()

        def bwd():
            dx, acc = self.into(y)
            self.call("dhw_op_unary_bwd", 0, ya.g.data_ptr(), y.d.data_ptr(), n, dx.data_ptr(), acc)
        self.record(ya, bwd)
        return ya

    # ---- differentiable ops ------------------------------------------------------------------------------------------
    def linear(self, x: Var, W: Var, b: Var | None, addend: Var | None = None, silu_out: bool = False, film=None):
        """nn.Linear on rows: x [R, K], W [N, K] (torch layout) -> [R, N]; ``addend`` [R, N]: the residual added to the result
        in the GEMM's output pass (y = x W^T + b + addend).  ``film`` = (table, col_g, col_b, B, act, addend): returns
        film_cols(y, ...) instead, evaluated in the same pass."""
        R, K = x.d.shape
        N = W.d.shape[0]
        if film is not None and -(-R // 64) * -(-N // 64) < 64 and K >= 512:   # (split-K output: the FiLM as its own pass)
            return self.film_cols(self.linear(x, W, b, addend), *film)
        if (addend is not None or silu_out) and -(-R // 64) * -(-N // 64) < 64 and K >= 512:   # (split-K output: separate passes)
            y = self.linear(x, W, b)
            y = self.add(y, addend) if addend is not None else y
            return (y, self.silu(y)) if silu_out else y
        # few output tiles and a long contraction (sigma_ffn's 2048 -> 32): zero the output and let the GEMM split K over
        # workgroups with atomics, instead of one workgroup walking all of K
        split = -(-R // 64) * -(-N // 64) < 64 and K >= 512
        y = Var(torch.zeros(R, N, device=self.dev) if split else self.new(R, N))
        act = self.new(R, N) if silu_out else None     # ``silu_out``: also returns SiLU(y), written by the same GEMM pass
        fout, frider = self._film_rider(R, N, film) if film is not None else (None, None)
        self.gemm(x.d, 0, K, 1, W.d, 0, 1, K, y.d, 0, N, 1, R, N, K, bias=b.d if b is not None else None, acc=split,
                  addend=addend.d if addend is not None else None, act_out=act, film=frider)

        def bwd():
            dy = y.g
            dgrad = None
            if not x.leaf:
                if x.pre is not None:       # x = SiLU(u): d u (+)= (dy W) * SiLU'(u) in the GEMM's output pass, x's own backward has nothing to do
                    dx, acc = self.into(x.pre)
                    dgrad = ((dy, 0, N, 1, W.d, 0, K, 1, dx, 0, K, 1, R, K, N), dict(acc=acc, dsilu_of=x.pre.d))
                else:
                    dx, acc = self.into(x)
                    dgrad = ((dy, 0, N, 1, W.d, 0, K, 1, dx, 0, K, 1, R, K, N), dict(acc=acc))         # dx (+)= dy W
            dW, db = W.grad(), b.grad() if b is not None else None
            # dW += dy^T x, db += column sums of dy — in one launch with the data gradient (both read dy, neither the other's output)
            self.gemm_pair(((dy, 0, 1, N, x.d, 0, K, 1, dW, 0, K, 1, N, K, R), dict(acc=True, rowsum=db)), dgrad)
            if addend is not None:
                self._grad_to(addend, dy)
        self.record(y, bwd)
        if film is not None:
            return self._film_record(y, fout, film)
        return (y, self._silu_of(y, act)) if silu_out else y

    def linear_group(self, specs):
        """Independent nn.Linears — ``specs``: [(x, W, b)], e.g. the q / k / v projections of one attention — as ONE launch forward
        and, backward, one launch for their weight gradients and every data gradient whose destination no other member writes
        (two projections of the same input add into the same gradient buffer: the second one follows in its own launch)."""
        def plain(x, W):
            R, K = x.d.shape
            return not x.leaf and not (-(-R // 64) * -(-W.d.shape[0] // 64) < 64 and K >= 512)
        if not all(plain(x, W) for x, W, _ in specs):
            return [self.linear(x, W, b) for x, W, b in specs]
        ys, items = [], []
        for x, W, b in specs:
            (R, K), N = x.d.shape, W.d.shape[0]
            y = Var(self.new(R, N))
            ys.append(y)
            items.append(((x.d, 0, K, 1, W.d, 0, 1, K, y.d, 0, N, 1, R, N, K), dict(bias=b.d if b is not None else None)))
        self.gemm_group(items)

        def bwd():
            live = [(x, W, b, y) for (x, W, b), y in zip(specs, ys) if y.g is not None]
            group, later, written = [], [], set()
            for x, W, b, y in live:       # dW += dy^T x, db += column sums of dy
                (R, K), N = x.d.shape, W.d.shape[0]
                group.append(((y.g, 0, 1, N, x.d, 0, K, 1, W.grad(), 0, K, 1, N, K, R), dict(acc=True, rowsum=b.grad() if b is not None else None)))
            for x, W, b, y in live:       # dx (+)= dy W
                (R, K), N = x.d.shape, W.d.shape[0]
                dx, acc = self.into(x.pre if x.pre is not None else x)     # (x = SiLU(u): d u straight out of the GEMM's output pass)
                item = ((y.g, 0, N, 1, W.d, 0, K, 1, dx, 0, K, 1, R, K, N), dict(acc=acc, dsilu_of=x.pre.d if x.pre is not None else None))
                (later if dx.data_ptr() in written else group).append(item)
                written.add(dx.data_ptr())
            if group:
                self.gemm_group(group)
            for a, k in later:
                self.gemm(*a, **k)
        self.steps.append(bwd)
        return ys

    def conv3(self, x: Var, W: Var, b: Var, L: int, addend: Var | None = None, silu_out: bool = False, defer: list | None = None, film=None):
        """nn.Conv1d(k=3, padding='same') on C-last rows: x [B*L, Cin], W [Cout, Cin, 3] -> [B*L, Cout].  W (and its gradient) may
        have any strides: TrainModel keeps the Conv1d weights as [tap][Cout][Cin] in memory (unit stride along Cin), which makes the
        weight the 16-byte-load operand of all three GEMMs; a torch-contiguous W works too (scalar loads, stride-3 stores)."""
        R, Cin = x.d.shape
        Cout = W.d.shape[0]
        sco, sci, st = W.d.stride()
        y = Var(self.new(R, Cout))
        merged = Cin % 32 == 0 and Cout % 32 == 0      # the three taps as one contraction over K = 3 Cin (dhw_gemm_desc.taps)
        act = self.new(R, Cout) if silu_out and merged else None
        fout, frider = self._film_rider(R, Cout, film) if film is not None and merged else (None, None)   # (``film``: as in ``linear``)
        if merged:
            fwd = ((x.d, 0, Cin, 1, W.d, 0, sci, sco, y.d, 0, Cout, 1, R, Cout, 3 * Cin),
                   dict(bias=b.d, taps=3, a_shift=-1, a_tap_shift=1, sbt=st, lr=L, addend=addend.d if addend is not None else None, act_out=act, film=frider))
            if defer is not None:      # (the caller launches it together with other independent GEMMs: gemm_group)
                defer.append(fwd)
            else:
                self.gemm(*fwd[0], **fwd[1])
        else:
            for t in range(3):
                self.gemm(x.d, 0, Cin, 1, W.d, t * st, sci, sco, y.d, 0, Cout, 1, R, Cout, Cin, bias=b.d if t == 0 else None, acc=t > 0,
                          a_shift=t - 1, lr=L, addend=addend.d if addend is not None and t == 0 else None)

        def bwd():
            dy, dW, db = y.g, W.grad(), b.grad()
            gco, gci, gt = dW.stride()
            fuse = merged and x.pre is not None    # x = SiLU(u): the merged data-gradient GEMM writes d u itself
            dx, acc = self.into(x.pre if fuse else x)
            # dx[r] += sum_t dy[r - (t-1)] W[:, :, t];  dW[:, :, t] += dy^T x[r + (t-1)]
            if merged:   # (weight gradient — the taps as the inner batch index, db riding along — and data gradient in one launch)
                self.gemm_pair(((dy, 0, 1, Cout, x.d, 0, Cin, 1, dW, 0, gco, gci, Cout, Cin, R),
                                dict(acc=True, nzi=3, zc=(0, gt), b_shift=-1, b_z_shift=1, lr=L, rowsum=db)),
                               ((dy, 0, Cout, 1, W.d, 0, sco, sci, dx, 0, Cin, 1, R, Cin, 3 * Cout),
                                dict(acc=acc, taps=3, a_shift=1, a_tap_shift=-1, sbt=st, lr=L, dsilu_of=x.pre.d if fuse else None)))
            else:
                for t in range(3):
                    self.gemm(dy, 0, Cout, 1, W.d, t * st, sco, sci, dx, 0, Cin, 1, R, Cin, Cout, acc=acc or t > 0, a_shift=1 - t, lr=L)
                    self.gemm(dy, 0, 1, Cout, x.d, 0, Cin, 1, dW, t * gt, gco, gci, Cout, Cin, R, acc=True, b_shift=t - 1, lr=L,
                              rowsum=db if t == 1 else None)
            if addend is not None:
                self._grad_to(addend, dy)
        self.record(y, bwd)
        if film is not None:
            return self._film_record(y, fout, film) if merged else self.film_cols(y, *film)
        if not silu_out:
            return y
        return (y, self._silu_of(y, act)) if act is not None else (y, self.silu(y))

    def unary(self, kind: int, x: Var) -> Var:
        y = Var(torch.empty_like(x.d), leaf=x.leaf)     # a function of inputs only needs no gradient either
        n = x.d.numel()
        self.call("dhw_op_unary", kind, x.d.data_ptr(), n, y.d.data_ptr())
        if kind == 0 and not x.leaf and FUSE_DSILU:
            y.pre = x
        saved = x.d if kind == 0 else y.d
        def bwd():
            if x.leaf:
                return
            dx, acc = self.into(x)
            self.call("dhw_op_unary_bwd", kind, y.g.data_ptr(), saved.data_ptr(), n, dx.data_ptr(), acc)
        self.record(y, bwd)
        return y

    def silu(self, x):
        return self.unary(0, x)

    def sigmoid(self, x):
        return self.unary(1, x)

    def add(self, a: Var, b: Var) -> Var:
        y = Var(torch.empty_like(a.d))
        n = a.d.numel()
        self.call("dhw_op_add", a.d.data_ptr(), b.d.data_ptr(), n, y.d.data_ptr(), 0)

        def bwd():
            # d a = d b = d y.  y's gradient buffer is dead after this step (y's consumers ran before it in the reverse sweep), so
            # the first input that has no gradient yet simply TAKES the buffer (later fan-in adds go into it); only the other input
            # needs a copy / an add.  One launch per residual add in the backward instead of two.
            taken = False
            for v in (a, b):
                if v.g is None and not taken and not isinstance(v, _ViewVar):
                    v.g = y.g
                    taken = True
                    continue
                dv, acc = self.into(v)
                self.call("dhw_op_add", y.g.data_ptr(), None, n, dv.data_ptr(), acc)
        self.record(y, bwd)
        return y

    def add_rows(self, x: Var, table: torch.Tensor, B: int) -> Var:
        R, Cc = x.d.shape
        y = Var(torch.empty_like(x.d))
        self.call("dhw_op_add_rows", x.d.data_ptr(), table.data_ptr(), B, R // B, Cc, y.d.data_ptr())
        def bwd():
            if x.g is None and not isinstance(x, _ViewVar):
                x.g = y.g          # (d x = d y, and y's buffer is dead after this step: take it instead of copying)
                return
            dx, acc = self.into(x)
            self.call("dhw_op_add", y.g.data_ptr(), None, x.d.numel(), dx.data_ptr(), acc)
        self.record(y, bwd)
        return y

    def film(self, x: Var, gamma: Var, beta: Var, B: int) -> Var:
        """x * gamma[b] + beta[b] (conditioning.py:23-26); gamma, beta [B, C]."""
        R, Cc = x.d.shape
        L = R // B
        y = Var(torch.empty_like(x.d))
        self.call("dhw_op_film", x.d.data_ptr(), gamma.d.data_ptr(), beta.d.data_ptr(), Cc, B, L, Cc, y.d.data_ptr())
        def bwd():
            dx, acc = self.into(x)
            self.call("dhw_op_film_bwd", y.g.data_ptr(), x.d.data_ptr(), gamma.d.data_ptr(), Cc, B, L, Cc, dx.data_ptr(), acc,
                      gamma.grad().data_ptr(), beta.grad().data_ptr())
        self.record(y, bwd)
        return y

    def _film_rider(self, R: int, Cc: int, spec):
        """``spec`` = (table, col_g, col_b, B, act, addend): the FiLM output tensor of a GEMM that evaluates film_cols in its own
        output pass, and the ``film`` tuple of ``gemm_desc``."""
        table, col_g, col_b, B, act, addend = spec
        out = self.new(R, Cc)
        base = table.d.data_ptr()
        return out, (base + col_g * _F, base + col_b * _F, table.d.shape[1], R // B, bool(act), out, addend.d if addend is not None else None)

    def _film_record(self, u: Var, out: torch.Tensor, spec) -> Var:
        """The backward of a FiLM output that rode on the GEMM producing ``u`` (same closure as film_cols)."""
        table, col_g, col_b, B, act, addend = spec
        R, Cc = u.d.shape
        L, TOT = R // B, table.d.shape[1]
        base = table.d.data_ptr()
        y = Var(out)

        def bwd():
            dx, acc = self.into(u)
            gbase = table.grad().data_ptr()
            self.call("dhw_op_film_act_bwd", y.g.data_ptr(), u.d.data_ptr(), base + col_g * _F, base + col_b * _F, TOT, B, L, Cc, int(act),
                      dx.data_ptr(), acc, gbase + col_g * _F, gbase + col_b * _F)
            if addend is not None:
                self._grad_to(addend, y.g)
        self.record(y, bwd)
        return y

    def film_cols(self, x: Var, table: Var, col_g: int, col_b: int, B: int, act: bool = False, addend: Var | None = None) -> Var:
        """``film`` with gamma / beta taken from columns [col, col + C) of a [B, TOT] table (dhw_op_film_table); ``act``: followed
        by SiLU in the same pass (the ConvBlock's ``SiLU(affine(conv(.)))``, cnn.py:70-80) — one launch each way instead of two,
        the FiLM output is recomputed in the backward instead of stored."""
        R, Cc = x.d.shape
        L, TOT = R // B, table.d.shape[1]
        y = Var(torch.empty_like(x.d))
        base = table.d.data_ptr()
        self.call("dhw_op_film_act", x.d.data_ptr(), base + col_g * _F, base + col_b * _F, TOT, B, L, Cc, int(act),
                  addend.d.data_ptr() if addend is not None else None, y.d.data_ptr())

        def bwd():
            dx, acc = self.into(x)
            gbase = table.grad().data_ptr()
            self.call("dhw_op_film_act_bwd", y.g.data_ptr(), x.d.data_ptr(), base + col_g * _F, base + col_b * _F, TOT, B, L, Cc, int(act),
                      dx.data_ptr(), acc, gbase + col_g * _F, gbase + col_b * _F)
            if addend is not None:
                self._grad_to(addend, y.g)
        self.record(y, bwd)
        return y

    def ln_film_cols(self, x: Var, table: Var, col_g: int, col_b: int, B: int, addend: Var | None = None, silu_out: bool = False,
                     pe: torch.Tensor | None = None):
        """LayerNorm followed by FiLM (every EncoderLayer / TextStyleEncoder pairs them: model.py:44-58, text_style.py:98-110) in one
        pass each way; the normalised rows are recomputed in the backward from the saved mean / rstd."""
        R, Cc = x.d.shape
        L, TOT = R // B, table.d.shape[1]
        y = Var(torch.empty_like(x.d))
        mean, rstd = self.new(R), self.new(R)
        base = table.d.data_ptr()
        act = torch.empty_like(x.d) if silu_out else None
        ype = torch.empty_like(x.d) if pe is not None else None     # ``pe`` [L, C]: also returns y + pe[l] (add_rows of the result) from the same pass
        self.call("dhw_op_ln_film", x.d.data_ptr(), B, L, Cc, base + col_g * _F, base + col_b * _F, TOT,
                  addend.d.data_ptr() if addend is not None else None, y.d.data_ptr(), act.data_ptr() if silu_out else None,
                  pe.data_ptr() if pe is not None else None, ype.data_ptr() if pe is not None else None, mean.data_ptr(), rstd.data_ptr())

        def bwd():
            dx, acc = self.into(x)
            gbase = table.grad().data_ptr()
            self.call("dhw_op_ln_film_bwd", y.g.data_ptr(), x.d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), base + col_g * _F, TOT, B, L, Cc,
                      dx.data_ptr(), acc, gbase + col_g * _F, gbase + col_b * _F)
            if addend is not None:
                self._grad_to(addend, y.g)
        self.record(y, bwd)
        outs = [y]
        if silu_out:
            outs.append(self._silu_of(y, act))
        if pe is not None:
            yp = Var(ype)
            self.record(yp, lambda: self._grad_to(y, yp.g))     # d y (+)= d (y + pe)
            outs.append(yp)
        return outs[0] if len(outs) == 1 else tuple(outs)

    def layernorm(self, x: Var) -> Var:
        R, Cc = x.d.shape
        y = Var(torch.empty_like(x.d))
        mean, rstd = self.new(R), self.new(R)
        self.call("dhw_op_layernorm", x.d.data_ptr(), R, Cc, y.d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
        def bwd():
            dx, acc = self.into(x)
            self.call("dhw_op_layernorm_bwd", y.g.data_ptr(), y.d.data_ptr(), rstd.data_ptr(), R, Cc, dx.data_ptr(), acc)
        self.record(y, bwd)
        return y

    def resample(self, mode: int, x: Var) -> Var:
        """mode 0: AvgPool1d(2); mode 2: Upsample(x2, nearest) — over the row axis of C-last rows (L even)."""
        R, Cc = x.d.shape
        Ro = R // 2 if mode == 0 else R * 2
        y = Var(self.new(Ro, Cc))
        self.call("dhw_op_resample", mode, x.d.data_ptr(), Ro, Cc, y.d.data_ptr(), 0)
        def bwd():
            dx, acc = self.into(x)
            self.call("dhw_op_resample", mode + 1, y.g.data_ptr(), R, Cc, dx.data_ptr(), acc)
        self.record(y, bwd)
        return y

    def embedding(self, ids: torch.Tensor, table: Var) -> Var:
        R = ids.numel()
        Cc = table.d.shape[1]
        y = Var(self.new(R, Cc))
        self.call("dhw_op_embedding", ids.data_ptr(), table.d.data_ptr(), R, Cc, y.d.data_ptr())
        self.record(y, lambda: self.call("dhw_op_embedding_bwd", ids.data_ptr(), y.g.data_ptr(), R, Cc, table.grad().data_ptr()))
        return y

    def attention(self, q: Var, k: Var, v: Var, B: int, H: int, mask: torch.Tensor | None) -> Var:
        """scaled_dp_attn over heads (attention.py:26-45, 77-87): q [B*Lq, H*D], k / v [B*Lk, H*D] -> [B*Lq, H*D].
        Heads are addressed by strides (no split / merge copies)."""
        HD = q.d.shape[1]
        D = HD // H
        Lq, Lk = q.d.shape[0] // B, k.d.shape[0] // B
        scale = 1.0 / math.sqrt(D)
        S = self.new(B, H, Lq, Lk)
        P = self.new(B, H, Lq, Lk)
        o = Var(self.new(B * Lq, HD))
        zq, zk, zs = (Lq * HD, D), (Lk * HD, D), (H * Lq * Lk, Lq * Lk)
        self.gemm(q.d, 0, HD, 1, k.d, 0, 1, HD, S, 0, Lk, 1, Lq, Lk, D, nzo=B, nzi=H, za=zq, zb=zk, zc=zs)             # S = Q K^T
        self.call("dhw_op_softmax", S.data_ptr(), B * H * Lq, Lk, H * Lq, mask.data_ptr() if mask is not None else None, scale, P.data_ptr())
        self.gemm(P, 0, Lk, 1, v.d, 0, HD, 1, o.d, 0, HD, 1, Lq, D, Lk, nzo=B, nzi=H, za=zs, zb=zk, zc=zq)             # O = P V

        def bwd():
            do = o.g
            dP = S   # the scores are dead after the softmax: reuse their buffer
            dv, av = self.into(v)
            self.gemm_group([((P, 0, 1, Lk, do, 0, HD, 1, dv, 0, HD, 1, Lk, D, Lq), dict(acc=av, nzo=B, nzi=H, za=zs, zb=zq, zc=zk)),     # dV (+)= P^T dO
                             ((do, 0, HD, 1, v.d, 0, 1, HD, dP, 0, Lk, 1, Lq, Lk, D), dict(nzo=B, nzi=H, za=zq, zb=zk, zc=zs))])          # dP = dO V^T
            self.call("dhw_op_softmax_bwd", dP.data_ptr(), P.data_ptr(), B * H * Lq, Lk, scale, dP.data_ptr())                # dS (in place)
            dq, aq = self.into(q)
            dk, ak = self.into(k)
            if dq.data_ptr() == dk.data_ptr():   # (q and k are the same tensor's gradient: the two updates of it stay ordered)
                self.gemm(dP, 0, Lk, 1, k.d, 0, HD, 1, dq, 0, HD, 1, Lq, D, Lk, acc=aq, nzo=B, nzi=H, za=zs, zb=zk, zc=zq)
                self.gemm(dP, 0, 1, Lk, q.d, 0, HD, 1, dk, 0, HD, 1, Lk, D, Lq, acc=1, nzo=B, nzi=H, za=zs, zb=zq, zc=zk)
            else:
                self.gemm_group([((dP, 0, Lk, 1, k.d, 0, HD, 1, dq, 0, HD, 1, Lq, D, Lk), dict(acc=aq, nzo=B, nzi=H, za=zs, zb=zk, zc=zq)),   # dQ (+)= dS K
                                 ((dP, 0, 1, Lk, q.d, 0, HD, 1, dk, 0, HD, 1, Lk, D, Lq), dict(acc=ak, nzo=B, nzi=H, za=zs, zb=zq, zc=zk))])  # dK (+)= dS^T Q
        self.record(o, bwd)
        return o

    def dropout(self, x: Var, keep: torch.Tensor, p: float) -> Var:
        """nn.Dropout(p) with the keep-mask supplied (1 = kept)."""
        y = Var(torch.empty_like(x.d), leaf=x.leaf)
        n = x.d.numel()
        scale = 1.0 / (1.0 - p)
        self.call("dhw_op_mask_mul", x.d.data_ptr(), keep.data_ptr(), scale, n, y.d.data_ptr(), 0)
        def bwd():
            if x.leaf:
                return
            dx, acc = self.into(x)
            self.call("dhw_op_mask_mul", y.g.data_ptr(), keep.data_ptr(), scale, n, dx.data_ptr(), acc)
        self.record(y, bwd)
        return y

    def backward(self):
        for step in reversed(self.steps):
            step()
        self.steps = []

    def mark(self, fn):
        """A position of the forward: ``fn`` runs when the backward sweep comes back to it, i.e. when every backward kernel of the ops
        recorded AFTER this point has been enqueued (the gradient-bucket markers)."""
        self.steps.append(fn)

    def record(self, out: Var, fn):
        """fn runs in the backward sweep iff a gradient reached ``out``."""
        self.steps.append(lambda: fn() if out.g is not None else None)


class TrainModel:
    """``DiffusionModel`` (model.py:61-199) for training: parameters as fp32 device tensors under the reference's names."""

    STYLE_DROP = 0.3   # text_style.py:88

    def __init__(self, state_dict: dict, num_layers: int = 2, device=None, drop_rate: float = 0.0, seed: int = 0, precision: str = "fp32"):
        """``drop_rate``: the EncoderLayers' dropout (model.py:23; configs/best.yml trains with 0.0, the class default is 0.1).
        ``precision``: "fp32" — exact-f32 MFMA everywhere, the mode the gradient fixtures pin — or "bf16": mixed precision, every
        GEMM (Linear / Conv1d / attention, forward and backward) contracts bf16-rounded operands with fp32 accumulation; weights,
        activations in memory, gradients, optimizer state and all non-GEMM arithmetic stay fp32 — with one exception: a bias
        gradient is summed inside its layer's weight-gradient GEMM from the bf16-rounded dy tile (dhw_gemm_desc.rowsum), so it
        carries the same operand rounding as that weight gradient (tests: the bf16 cases of the GEMM sweep)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision
        if not 0.0 <= drop_rate < 1.0:
            raise ValueError("drop_rate must be in [0, 1)")
        if not torch.cuda.is_available():
            raise RuntimeError("TrainModel needs the MI355X: the training step has no CPU path")
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.num_layers, self.drop_rate = num_layers, float(drop_rate)
        # {seed, draw index} of the device generator behind the dropout masks (and GraphedTrainStep's eps draw)
        self.seed = seed
        self.rng = torch.tensor([seed, 0], dtype=torch.int64, device=self.dev)
        self._masks, self._site = None, 3
        self.names = list(state_dict.keys())
        shapes = {k: tuple(np.shape(v)) for k, v in state_dict.items()}
        # Conv1d weights [Cout, Cin, 3] are kept as [tap][Cout][Cin] in the flat buffers (unit stride along Cin: the weight is then
        # the 16-byte-load operand of the forward, data-gradient and weight-gradient GEMMs alike; in torch's layout it was a stride-3
        # gather / scatter).  Their named views are the permuted views, so shapes and values are the reference's everywhere.
        conv_w = {k for k in shapes if len(shapes[k]) == 3 and shapes[k][2] == 3}
        host = {}
        for k, v in state_dict.items():
            h = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).detach().to("cpu", torch.float32)
            host[k] = (h.permute(2, 0, 1).contiguous() if k in conv_w else h).reshape(-1)
        # ONE flat fp32 buffer for all parameters and one for all gradients (40 MB each): a single memset clears the
        # gradients, a single kernel pair clips and applies Adam.  The tensors sit in it in the order the backward sweep COMPLETES
        # their gradients (grad_bucket; state_dict order inside a bucket, odd-sized tensors last), so each of the N_BUCKETS
        # gradient buckets is one contiguous range that can be all-reduced the moment its marker fires (GradBucketReducer).
        # self.names / state_dict() stay in the reference's order: the layout is private to the flat buffers.
        self.layout = sorted(self.names, key=lambda k: (grad_bucket(k), host[k].numel() % 4 != 0))   # (stable: state_dict order otherwise)
        self.flat = torch.cat([host[k] for k in self.layout]).to(self.dev)
        self.flat_grad = torch.zeros_like(self.flat)
        self.p, self.offset, off = {}, {}, 0
        self.bucket_ranges, self.bucket_hook = [], None     # [start, end) of each bucket; bucket_hook(i): called when bucket i < N_BUCKETS - 1 is final
        for b in range(N_BUCKETS):
            n_b = sum(host[k].numel() for k in self.layout if grad_bucket(k) == b)
            start = self.bucket_ranges[-1][1] if self.bucket_ranges else 0
            self.bucket_ranges.append((start, start + n_b))
        run = 0
        for k in self.layout:
            if run % 4 and host[k].numel() % 4 == 0:
                raise ValueError(f"parameter {k} would start unaligned in the flat buffer (an odd-sized tensor precedes it)")
            run += host[k].numel()
        for k in self.layout:
            n = host[k].numel()
            view = (lambda buf: buf[off:off + n].view(3, *shapes[k][:2]).permute(1, 2, 0)) if k in conv_w else (lambda buf: buf[off:off + n].view(shapes[k]))
            v = Var(view(self.flat))
            v.g = view(self.flat_grad)
            self.p[k] = v
            self.offset[k] = off
            off += n
        # every AffineTransformLayer's gamma / beta Linear as columns of one [B, TOT] table (dhw_op_film_table): column ->
        # (weight row, bias) offsets inside the flat buffer; film_cols[name] = (first gamma column, first beta column)
        woff, boff, self.film_cols, col = [], [], {}, 0
        for k in self.names:
            if k.endswith(".gamma_emb.weight"):
                base, Cc = k[:-len(".gamma_emb.weight")], shapes[k][0]
                if shapes[k][1] != 32:
                    raise ValueError("the FiLM Linears take the 32-wide sigma embedding")
                self.film_cols[base] = (col, col + Cc)
                for kind in ("gamma_emb", "beta_emb"):
                    woff.append(self.offset[f"{base}.{kind}.weight"] + 32 * torch.arange(Cc, dtype=torch.int64))
                    boff.append(self.offset[f"{base}.{kind}.bias"] + torch.arange(Cc, dtype=torch.int64))
                col += 2 * Cc
        self.film_total = col
        woff_all = torch.cat(woff)
        if bool((woff_all % 4 != 0).any()):   # film_table_fwd_kernel reads each 32-float weight row as f32x4 from flat + offset
            raise ValueError("the FiLM weight rows are not 16-byte aligned in the flat parameter buffer (unexpected state_dict sizes / order)")
        self.film_woff, self.film_boff = woff_all.to(self.dev), torch.cat(boff).to(self.dev)
        self.c1 = self.p["input_dense.weight"].d.shape[0]
        self.style_drop = self.STYLE_DROP   # (0.0 = the TextStyleEncoder's Dropout in eval mode: dhg_amd.DiffusionModel's differentiable forward)
        self._pe = {}
        self.tape = None

    # ---- parameters ------------------------------------------------------------------------------------------------------
    def parameters(self):
        """The flat parameter buffer (every named parameter is a view into it, in state_dict order; Conv1d weights as
        [tap][Cout][Cin], their named views permuted back to [Cout, Cin, 3])."""
        return [self.flat]

    def grads(self):
        return [self.flat_grad]

    def zero_grad(self):
        self.flat_grad.zero_()   # one memset

    def state_dict(self):
        return {k: self.p[k].d for k in self.names}

    def pe(self, L, dim, factor):
        key = (L, dim, factor)
        if key not in self._pe:
            self._pe[key] = positional_encoding(L, dim, factor).to(self.dev)
        return self._pe[key]

    def prepare_tables(self, L: int, Lt: int):
        """Upload the positional-encoding tables a (L, Lt) batch needs (model.py:40-44) ahead of a graph capture."""
        c2, c3 = self.p["enc3.text_dense.weight"].d.shape[0], self.p["enc5.text_dense.weight"].d.shape[0]
        for d, Lx, f in ((c2, L // 2, 4), (c3, L // 4, 2), (2 * c2, L // 8, 1)):
            self.pe(Lx, d, f)
            self.pe(Lt, d, 1.0)

    # ---- modules ---------------------------------------------------------------------------------------------------------
    def _lin(self, t, x, name, addend=None, silu_out=False):
        return t.linear(x, self.p[name + ".weight"], self.p.get(name + ".bias"), addend, silu_out)

    def _ffn(self, t, x, name, addend=None, x_act=None):
        """ff_network (utils/nn.py:145-175): SiLU -> Linear -> SiLU -> Linear (+ the residual that follows it, in the last GEMM).
        ``x_act``: SiLU(x) where the pass that produced x wrote it already; the inner SiLU comes out of the first GEMM's pass."""
        _, ha = self._lin(t, x_act if x_act is not None else t.silu(x), name + ".1", silu_out=True)
        return self._lin(t, ha, name + ".3", addend)

    def _film_table(self, t, sigma, B):
        """gamma / beta of all AffineTransformLayers for this sigma: one launch forward, two backward (instead of 76 small
        GEMMs forward and 228 GEMM / bias-sum launches backward)."""
        table = Var(t.new(B, self.film_total))
        args = (self.flat.data_ptr(), self.film_woff.data_ptr(), self.film_boff.data_ptr(), B, self.film_total)
        t.call("dhw_op_film_table", sigma.d.data_ptr(), *args, table.d.data_ptr())
        t.record(table, lambda: t.call("dhw_op_film_table_bwd", table.g.data_ptr(), sigma.d.data_ptr(), *args, self.flat_grad.data_ptr(),
                                       sigma.grad().data_ptr()))
        self._film = table

    def _affine(self, t, x, sigma, name, B, act=False, addend=None):
        return t.film_cols(x, self._film, *self.film_cols[name], B, act, addend)

    def _ln_affine(self, t, x, sigma, name, B, addend=None, silu_out=False, pe=None):
        """affine(layernorm(x)) (+ addend) as one fused pass (LN statistics span at most 512 channels here)."""
        return t.ln_film_cols(x, self._film, *self.film_cols[name], B, addend, silu_out, pe)

    def _mha(self, t, q, k, v, name, B, H, mask=None, addend=None):
        qp, kp, vp = t.linear_group([(x, self.p[f"{name}.{w}.weight"], self.p[f"{name}.{w}.bias"]) for x, w in ((q, "wq"), (k, "wk"), (v, "wv"))])
        return self._lin(t, t.attention(qp, kp, vp, B, H, mask), name + ".dense", addend)

    def _convblock(self, t, x, sigma, name, B, L, x_act=None):
        """cnn.py:64-87.  ``x_act``: SiLU(x) where the pass that produced x wrote it already."""
        if not FILM_RIDER:
            return self._convblock_passes(t, x, sigma, name, B, L, x_act)
        # every affine (+ SiLU) (+ conv_skip) rides on the output pass of the GEMM in front of it (dhw_gemm_desc.film_out)
        film = lambda k, act, addend=None: (self._film, *self.film_cols[f"{name}.affine{k}"], B, act, addend)   # noqa: E731
        conv = lambda v, n, defer=None, fl=None: t.conv3(v, self.p[f"{name}.{n}.weight"], self.p[f"{name}.{n}.bias"], L, defer=defer, film=fl)   # noqa: E731
        xa = x_act if x_act is not None else t.silu(x)
        both = []                      # conv_skip(x) and conv1(SiLU(x)) are independent: one launch
        skip, h = conv(x, "conv_skip", both), conv(xa, "conv1", both, film(1, True))      # h = SiLU(affine1(conv1(.)))
        if both:
            t.gemm_group(both)
        h = conv(h, "conv2", None, film(2, True))                                           # SiLU(affine2(conv2(.)))
        return t.linear(h, self.p[name + ".fc.weight"], self.p[name + ".fc.bias"], film=film(3, False, skip))   # affine3(fc(h)) + conv_skip(x)

    def _convblock_passes(self, t, x, sigma, name, B, L, x_act=None):
        """The same with every affine as its own pass (DHW_TRAIN_FILM_RIDER=0)."""
        conv = lambda v, n, defer=None: t.conv3(v, self.p[f"{name}.{n}.weight"], self.p[f"{name}.{n}.bias"], L, defer=defer)   # noqa: E731
        xa = x_act if x_act is not None else t.silu(x)
        both = []
        skip, h1 = conv(x, "conv_skip", both), conv(xa, "conv1", both)
        if both:
            t.gemm_group(both)
        h = self._affine(t, h1, sigma, name + ".affine1", B, act=True)
        h = self._affine(t, conv(h, "conv2"), sigma, name + ".affine2", B, act=True)
        return self._affine(t, self._lin(t, h, name + ".fc"), sigma, name + ".affine3", B, addend=skip)

    def _drop(self, t, v, B):
        """EncoderLayer.drop (model.py:23): identity at rate 0; otherwise the next caller-supplied keep-mask (parity tests) or a
        mask drawn on the device for this site of this update."""
        if self.drop_rate == 0.0:
            return v
        if self._masks is not None:
            keep = next(self._masks).to(self.dev, torch.float32).contiguous().view_as(v.d)
        else:
            keep = torch.empty_like(v.d)
            t.call("dhw_op_keep_mask", self.rng.data_ptr(), self._site, v.d.numel(), v.d.numel() // B, self.drop_rate, keep.data_ptr())
        self._site += 1
        return t.dropout(v, keep, self.drop_rate)

    def _encoder(self, t, x, text, sigma, mask, name, B, H, pos_factor, td=None):
        """EncoderLayer.forward (model.py:36-58); ``text`` is SiLU(text features) already; ``td``: text_dense(text) where the
        caller has evaluated it (all layers' text_dense in one launch: they share their input)."""
        d = x.d.shape[1]
        Lx, Lt = x.d.shape[0] // B, text.d.shape[0] // B
        tx, text_pe = self._ln_affine(t, td if td is not None else self._lin(t, text, name + ".text_dense"), sigma, name + ".affine0", B, pe=self.pe(Lt, d, 1.0))   # text: SiLU(text features), shared
        x_pe = t.add_rows(x, self.pe(Lx, d, pos_factor), B)
        x2 = self._mha(t, x_pe, text_pe, tx, name + ".mha", B, H, mask)
        # the residual adds and the positional-encoding adds ride on the passes / GEMMs around them
        x2, x2_pe = self._ln_affine(t, self._drop(t, x2, B), sigma, name + ".affine1", B, addend=x, pe=self.pe(Lx, d, pos_factor))
        if self.drop_rate == 0.0:
            x3, x3a = self._ln_affine(t, self._mha(t, x2_pe, x2_pe, x2, name + ".mha2", B, H, addend=x2), sigma, name + ".affine2", B, silu_out=True)
            x4 = self._ffn(t, x3, name + ".ffn", addend=x3, x_act=x3a)
        else:       # (the dropout sits between the GEMM and the add)
            x3 = self._mha(t, x2_pe, x2_pe, x2, name + ".mha2", B, H)
            x3, x3a = self._ln_affine(t, t.add(x2, self._drop(t, x3, B)), sigma, name + ".affine2", B, silu_out=True)
            x4 = t.add(self._drop(t, self._ffn(t, x3, name + ".ffn", x_act=x3a), B), x3)
        return self._ln_affine(t, x4, sigma, name + ".affine3", B)

    def _text_style(self, t, ids, style, sigma, keep, B):
        """TextStyleEncoder.forward (text_style.py:96-110)."""
        n = "text_style_model"
        S = style.d.shape[1]
        st = t.dropout(style, keep, self.style_drop)                          # [B, S, 1280]
        up = _ViewVar(st, (B * S * 5, st.d.shape[2] // 5))                      # reshape_up(., 5): a pure view of the same rows
        stf = self._ffn(t, up, n + ".style_ffn")
        stf = self._ln_affine(t, stf, sigma, n + ".affine1", B)
        tx = t.embedding(ids, self.p[n + ".emb.weight"])
        tx = self._ln_affine(t, tx, sigma, n + ".affine2", B)
        tx, txa = self._ln_affine(t, self._mha(t, tx, stf, stf, n + ".mha", B, 8, addend=tx), sigma, n + ".affine3", B, silu_out=True)
        return self._ln_affine(t, self._ffn(t, tx, n + ".text_ffn", x_act=txa), sigma, n + ".affine4", B)

    # ---- forward / backward ----------------------------------------------------------------------------------------------
    def check_tokens(self, text: torch.Tensor):
        if text.is_cuda:
            raise ValueError("token ids are validated on the host: pass a CPU tensor")
        if int(text.min()) < 0 or int(text.max()) >= self.p["text_style_model.emb.weight"].d.shape[0]:
            raise ValueError("token id out of the embedding's range")

    def forward(self, strokes: torch.Tensor, text: torch.Tensor, sigma: torch.Tensor, style: torch.Tensor, style_keep: torch.Tensor | None = None,
                drop_masks=None):
        """strokes [B, L, 2], text int64 [B, Lt] (host), sigma [B, 1] (= sqrt(abar), train.py:49), style [B, S, 1280];
        ``style_keep``: the Dropout(0.3) keep-mask [B, S, 1280] (drawn here when omitted); ``drop_masks``: with drop_rate > 0,
        the EncoderLayers' keep-masks in call order (3 per layer: enc3, enc5, att_layers...), else drawn on the device.
        -> (score [B, L, 2], pen [B, L])."""
        dev = self.dev
        self._masks = iter(drop_masks) if drop_masks is not None else None
        f = lambda a: a.to(dev, torch.float32).contiguous()   # noqa: E731
        self.check_tokens(text)
        mask = (text == 0).to(torch.float32).to(dev).contiguous()          # create_padding_mask (utils/nn.py:189), host side
        if style_keep is None:
            style_keep = (torch.rand(style.shape) >= self.style_drop).float()
        return self.forward_device(f(strokes), text.to(dev, torch.int64).contiguous(), mask, f(sigma), f(style), f(style_keep))

    def forward_device(self, strokes, ids, mask, sigma, style, keep):
        """``forward`` on device-resident fp32 inputs (ids int64, mask = 1 at padding): kernel launches only."""
        dev = self.dev
        B, L, _ = strokes.shape
        if L % 8:
            raise ValueError("the stroke length must be a multiple of 8 (three AvgPool1d(2) stages)")
        t = self.tape = Tape(dev, bf16=self.precision == "bf16")
        self._site = 3
        x_in, sig_in, sty = Var(strokes.view(B * L, 2), leaf=True), Var(sigma.view(B, 1), leaf=True), Var(style, leaf=True)

        sigma_v = self._ffn(t, sig_in, "sigma_ffn")                                     # [B, 32]
        t.mark(lambda: self._bucket_done(3))       # the sweep is back here: text side, text_dense and the FiLM table are done
        self._film_table(t, sigma_v, B)
        txt = t.silu(self._text_style(t, ids, sty, sigma_v, keep, B))                  # SiLU([B*Lt, 2 c2]): every EncoderLayer's text_dense starts
                                                                                       # with it (model.py:38) — once, not once per layer
        # every EncoderLayer's text_dense (model.py:38) reads the same SiLU(text features): one launch for all of them
        enc_names = ["enc3", "enc5"] + [f"att_layers.{i}" for i in range(self.num_layers)]
        td = dict(zip(enc_names, t.linear_group([(txt, self.p[n + ".text_dense.weight"], self.p[n + ".text_dense.bias"]) for n in enc_names])))
        t.mark(lambda: self._bucket_done(2))       # ... encoder + input_dense done
        x, xa = self._lin(t, x_in, "input_dense", silu_out=True)
        h1 = self._convblock(t, x, sigma_v, "enc1", B, L, x_act=xa)
        h2 = self._convblock(t, t.resample(0, h1), sigma_v, "enc2", B, L // 2)
        h2 = self._encoder(t, h2, txt, sigma_v, mask, "enc3", B, 3, 4, td=td["enc3"])
        h3 = self._convblock(t, t.resample(0, h2), sigma_v, "enc4", B, L // 4)
        h3 = self._encoder(t, h3, txt, sigma_v, mask, "enc5", B, 4, 2, td=td["enc5"])
        t.mark(lambda: self._bucket_done(1))       # ... att_dense + bottleneck layers done
        x = self._lin(t, t.resample(0, h3), "att_dense")
        for i in range(self.num_layers):
            x = self._encoder(t, x, txt, sigma_v, mask, f"att_layers.{i}", B, 6, 1, td=td[f"att_layers.{i}"])
        t.mark(lambda: self._bucket_done(0))       # ... decoder + skip convolutions done
        # upsample(x) + skip_conv(h): the add rides on the skip convolution's output pass
        # (and SiLU of the sum, which the decoder block's conv1 starts with, is its second output)
        skip = lambda v, n, Lr, up: t.conv3(v, self.p[n + ".weight"], self.p[n + ".bias"], Lr, addend=up, silu_out=True)   # noqa: E731
        x, xa = skip(h3, "skip_conv3", L // 4, t.resample(2, x))
        x = self._convblock(t, x, sigma_v, "dec3", B, L // 4, x_act=xa)
        x, xa = skip(h2, "skip_conv2", L // 2, t.resample(2, x))
        x = self._convblock(t, x, sigma_v, "dec2", B, L // 2, x_act=xa)
        x, xa = skip(h1, "skip_conv1", L, t.resample(2, x))
        x = self._convblock(t, x, sigma_v, "dec1", B, L, x_act=xa)
        self._score = self._lin(t, x, "output_dense")
        self._pen = t.sigmoid(self._lin(t, x, "pen_lifts_dense.0"))
        return self._score.d.view(B, L, 2), self._pen.d.view(B, L)

    def _bucket_done(self, i: int):
        if self.bucket_hook is not None:
            self.bucket_hook(i)

    def backward(self, d_score: torch.Tensor, d_pen: torch.Tensor):
        """Accumulate every parameter gradient for the upstream gradients of the two outputs.  ``self.bucket_hook(i)`` (if set) is
        called from inside the sweep as soon as gradient bucket i = 0 .. N_BUCKETS - 2 is final; the last bucket is final on return."""
        self._score.g = d_score.contiguous().view_as(self._score.d)
        self._pen.g = d_pen.contiguous().view_as(self._pen.d)
        self.tape.backward()
        self.last_launches, self.last_gemm_flops = self.tape.launches, self.tape.flops
        self.tape = None


class _ViewVar(Var):
    """A reshaped alias of another Var: shares the data AND the gradient storage (reshape_up is a pure view)."""
    __slots__ = ("base", "shape")

    def __init__(self, base: Var, shape):
        self.base, self.shape = base, shape
        self.d = base.d.view(*shape)
        self.g = None
        self.leaf = base.leaf
        self.pre = None

    def grad(self):
        if self.g is None:
            self.g = self.base.grad().view(*self.shape)
        return self.g


def train_step(model: TrainModel, optimizer: Adam, batch: dict, alpha_set: torch.Tensor, step: int, *, eps: torch.Tensor | None = None,
               alphas: torch.Tensor | None = None, style_keep: torch.Tensor | None = None, drop_masks=None, d_model: int = 256,
               warmup: int = 10000, lr_mul: float = 1.0):
    """One update, in the reference's order (train.py:26-67): abar draw, eps draw, perturbation, forward, loss, backward,
    (data-parallel gradient mean when torch.distributed is initialised), clip + Adam at the Noam rate of update number ``step`` >= 1.
    batch: {"strokes" [B,L,3], "text" [B,Lt], "style" [B,S,1280]}.  Returns the device tensor (loss, score_loss, pen_loss)."""
    strokes3 = batch["strokes"]
    x, pen = strokes3[:, :, :2].float(), strokes3[:, :, 2].float()
    B = x.shape[0]
    if alphas is None:
        alphas = get_alphas(B, alpha_set)                    # [B, 1], torch's CPU generator like the reference
    if eps is None:
        eps = torch.randn(x.shape)
    x_pert = perturb(x, eps, alphas)
    model.zero_grad()
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
    model.rng.copy_(torch.tensor([model.seed, step * world + rank], dtype=torch.int64))
    score, pen_pred = model.forward(x_pert, batch["text"], torch.sqrt(alphas), batch["style"], style_keep, drop_masks)
    out, d_score, d_pen = loss_fn(eps, score, pen, pen_pred, alphas)
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()   # (also a one-rank group: the same code path)
    reducer = GradBucketReducer(model.flat_grad, model.bucket_ranges) if dist_on and GRAD_BUCKETS else None
    model.bucket_hook = reducer.launch if reducer else None
    try:
        model.backward(d_score, d_pen)
    finally:
        model.bucket_hook = None
    grads = model.grads()
    if reducer:
        reducer.launch(N_BUCKETS - 1)
        reducer.wait(average=True)
    elif dist_on:
        allreduce_grads(grads)
    model.last_grad_norm = float(optimizer.step(grads, noam_lr(step, d_model, warmup, lr_mul)))   # (one host sync per update)
    return out


class GraphedTrainStep:
    """``train_step`` with its gradient computation captured once into a hipGraph and replayed: the ~1100 kernel launches of
    an update (draws, perturbation, forward, loss, backward) become one graph launch, which takes the host off the critical
    path; the gradient all-reduce (multi-rank) and clip + Adam follow as three eager launches.  The batch
    lives in static device buffers that each call overwrites; the step's learning rate and Adam bias corrections are 8 floats
    in device memory rewritten before every replay.  One instance serves one batch shape (B, L, Lt, S)."""

    def __init__(self, model: TrainModel, optimizer: Adam, B: int, L: int, Lt: int, S: int = 14, d_model: int = 256, warmup: int = 10000,
                 lr_mul: float = 1.0, device_rng: bool = True, seed: int = 0):
        """``device_rng``: eps and the style Dropout mask are drawn inside the step by the library's Philox generator (as the
        reference draws them on its device, train.py:39, text_style.py:97); False: the caller passes them (tests)."""
        dev = model.dev
        self.device_rng, self.seed = device_rng, seed
        self.rng = model.rng           # one generator state for the eps / style draws and the model's dropout sites
        self.model, self.opt = model, optimizer
        self.shape = (B, L, Lt, S)
        self.sched = (d_model, warmup, lr_mul)
        z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
        # Every per-update input lives in ONE device block that a single asynchronous copy from pinned host memory fills (the
        # separate pageable copies of round 3 — strokes, pen, alphas, sigma, ids, mask, style, hyper, rng: nine host-blocking
        # transfers — left the GPU idle for ~0.4 ms between two updates, rocprofv3 trace of tools/bench_train.py).  Two pinned
        # blocks alternate, so the host fills update k + 1's inputs while update k runs.
        fields = [("ids", (B, Lt), torch.int64), ("rng", (2,), torch.int64), ("x", (B, L, 2), torch.float32), ("pen", (B, L), torch.float32),
                  ("alphas", (B,), torch.float32), ("sigma", (B, 1), torch.float32), ("mask", (B, Lt), torch.float32),
                  ("hyper", (9,), torch.float32), ("style", (B, S, 1280), torch.float32)]
        if not device_rng:
            fields += [("eps", (B, L, 2), torch.float32), ("keep", (B, S, 1280), torch.float32)]
        offs, off = {}, 0
        for name, shape, dt in fields:
            n = int(np.prod(shape)) * (8 if dt == torch.int64 else 4)
            offs[name] = (off, n, shape, dt)
            off += (n + 15) // 16 * 16
        self._stage_dev = torch.zeros(off, dtype=torch.uint8, device=dev)
        self._stage_host = [torch.zeros(off, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._stage_ev = [None, None]
        self._stage_turn = 0
        carve = lambda buf: {k: buf[o:o + n].view(dt).view(shape) for k, (o, n, shape, dt) in offs.items()}   # noqa: E731
        self._dv, self._hv = carve(self._stage_dev), [carve(h) for h in self._stage_host]
        d = self._dv
        self.x, self.pen, self.alphas, self.sigma, self.ids, self.mask, self.style, self.hyper = (d[k] for k in ("x", "pen", "alphas", "sigma", "ids", "mask", "style", "hyper"))
        self.eps = d["eps"] if not device_rng else z(B, L, 2)
        self.keep = d["keep"] if not device_rng else z(B, S, 1280)
        d["rng"].copy_(model.rng)
        self.rng = model.rng = d["rng"]      # (the model's dropout sites read the same generator state: one pointer, inside the block)
        self.sqnorm, self.out = z(1), z(3)
        model.prepare_tables(L, Lt)
        self.graph = None
        self.segments, self._reducer = None, None     # with a process group: N_BUCKETS graph segments + the bucketed all-reduce
        self._opt_in_graph = False

    def _body(self):
        """Everything of one update that runs on the device (what the graph holds)."""
        m = self.model
        B, L = self.shape[:2]
        x_pert = torch.empty_like(self.x)
        lib = _lib.lib()
        st = _stream(m.dev)
        if self.device_rng:
            _tcheck(lib.dhw_train_draw(self.rng.data_ptr(), B, L, self.eps.data_ptr(), self.keep.numel(), self.keep.numel() // B,
                                       TrainModel.STYLE_DROP, self.keep.data_ptr(), st))
        _tcheck(lib.dhw_train_perturb(self.x.data_ptr(), self.eps.data_ptr(), self.alphas.data_ptr(), B, L, x_pert.data_ptr(), st))
        m.zero_grad()
        score, pen_pred = m.forward_device(x_pert, self.ids, self.mask, self.sigma, self.style, self.keep)
        d_score, d_pen = torch.empty_like(score), torch.empty_like(pen_pred)
        _tcheck(lib.dhw_train_loss(self.eps.data_ptr(), score.data_ptr(), self.pen.data_ptr(), pen_pred.data_ptr(), self.alphas.data_ptr(), B, L,
                                   self.out.data_ptr(), d_score.data_ptr(), d_pen.data_ptr(), st))
        m.backward(d_score, d_pen)

    def _apply(self, reducer=None):
        """The optimizer behind the gradient all-reduce.  ``reducer``: the update's bucketed all-reduces are already in flight
        (GradBucketReducer: launched bucket by bucket while the backward was still running) — wait for them; without one
        (DHW_TRAIN_BUCKETS=0) ONE SUM all-reduce of the flat 40 MB buffer, issued here, behind the whole backward.  Adam applies
        1 / world size (hyper[8]) in its own pass either way; the collectives stay outside the captured graphs."""
        m = self.model
        if reducer is not None:
            reducer.launch(N_BUCKETS - 1)
            reducer.wait()
        elif torch.distributed.is_available() and torch.distributed.is_initialized():   # (also a one-rank group: the same code path)
            allreduce_grads(m.grads(), average=False)
        self.opt.step_dev(m.grads(), self.hyper, self.sqnorm)

    def _capture_segments(self):
        """With a process group the update is captured as N_BUCKETS graph SEGMENTS cut at the tape's bucket markers (one shared
        memory pool, replayed in capture order): after segment i has been enqueued, gradient bucket i is final and its all-reduce is
        issued beside segment i + 1."""
        import gc
        m = self.model
        gc.collect()
        torch.cuda.synchronize(m.dev)
        segs = []
        side = torch.cuda.Stream(device=m.dev)
        side.wait_stream(torch.cuda.current_stream(m.dev))
        with torch.cuda.stream(side):
            cur = torch.cuda.CUDAGraph()
            cur.capture_begin()
            state = {"cur": cur}

            def cut(i):
                state["cur"].capture_end()
                segs.append(state["cur"])
                if i != len(segs) - 1:
                    raise RuntimeError(f"gradient bucket markers fired out of order: {i} after {len(segs) - 1} segments")
                nxt = torch.cuda.CUDAGraph()
                nxt.capture_begin(pool=segs[0].pool())
                state["cur"] = nxt
            m.bucket_hook = cut
            try:
                self._body()
            finally:
                m.bucket_hook = None
                state["cur"].capture_end()
            segs.append(state["cur"])
        torch.cuda.current_stream(m.dev).wait_stream(side)
        if len(segs) != N_BUCKETS:
            raise RuntimeError(f"captured {len(segs)} graph segments for {N_BUCKETS} gradient buckets")
        return segs

    def __call__(self, batch: dict, alpha_set, step: int, *, eps=None, alphas=None, style_keep=None, graph: bool = True):
        B, L, Lt, S = self.shape
        strokes3 = batch["strokes"]
        if tuple(strokes3.shape) != (B, L, 3) or tuple(batch["text"].shape) != (B, Lt) or tuple(batch["style"].shape) != (B, S, 1280):
            raise ValueError(f"this step was built for strokes {(B, L, 3)}, text {(B, Lt)}, style {(B, S, 1280)}")
        self.model.check_tokens(batch["text"])
        if alphas is None:
            alphas = get_alphas(B, alpha_set)
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        turn = self._stage_turn
        self._stage_turn ^= 1
        if self._stage_ev[turn] is not None:
            self._stage_ev[turn].synchronize()       # the copy that last read this pinned block (two updates ago) has finished
        hv = self._hv[turn]
        hv["rng"][0], hv["rng"][1] = self.seed, step * world + rank
        if self.device_rng:
            if eps is not None or style_keep is not None:
                raise ValueError("this step draws eps and the dropout mask on the device (device_rng=True)")
        else:
            hv["eps"].copy_(eps if eps is not None else torch.randn(B, L, 2))
            hv["keep"].copy_(style_keep if style_keep is not None else (torch.rand(B, S, 1280) >= TrainModel.STYLE_DROP).float())
        hv["x"].copy_(strokes3[:, :, :2])
        hv["pen"].copy_(strokes3[:, :, 2])
        hv["alphas"].copy_(alphas.reshape(B))
        hv["sigma"].copy_(torch.sqrt(alphas).reshape(B, 1))
        hv["ids"].copy_(batch["text"])
        hv["mask"].copy_(batch["text"] == 0)
        hv["style"].copy_(batch["style"])
        hv["hyper"].copy_(torch.tensor(self.opt.hyper(noam_lr(step, *self.sched), 1.0 / world), dtype=torch.float32))
        self._gscale = 1.0 / world
        self._stage_dev.copy_(self._stage_host[turn], non_blocking=True)
        ev = self._stage_ev[turn] = self._stage_ev[turn] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.model.dev))
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        bucketed = dist_on and GRAD_BUCKETS
        if bucketed and self._reducer is None:
            self._reducer = GradBucketReducer(self.model.flat_grad, self.model.bucket_ranges)
        if not graph:
            self.model.bucket_hook = self._reducer.launch if bucketed else None
            try:
                self._body()
            finally:
                self.model.bucket_hook = None
            self._apply(self._reducer if bucketed else None)
            return self.out
        if bucketed:
            if self.graph is not None:
                raise RuntimeError("this step was captured without a process group; build a new GraphedTrainStep after init_process_group")
            if self.segments is None:
                self.segments = self._capture_segments()
            for i, g in enumerate(self.segments):
                g.replay()
                if i < N_BUCKETS - 1:
                    self._reducer.launch(i)       # bucket i is final behind segment i: its all-reduce runs beside segment i + 1
            self._apply(self._reducer)
            return self.out
        if self.graph is None:
            # loss_kernel accumulates into out[1], out[2] — dhw_train_loss zeroes them itself; capture on torch's capture stream.
            # Without a process group there is nothing between the backward and the optimizer: clip + Adam are the graph's last
            # nodes and an update is ONE launch; with one, the collective stays outside the capture and the optimizer follows it.
            self._opt_in_graph = not dist_on
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
                if self._opt_in_graph:
                    self.opt.step_dev(self.model.grads(), self.hyper, self.sqnorm)
        elif self._opt_in_graph and dist_on:
            raise RuntimeError("this step was captured without a process group; build a new GraphedTrainStep after init_process_group")
        self.graph.replay()
        if not self._opt_in_graph:
            self._apply()
        return self.out

    def grad_norm(self) -> float:
        """||g|| of the last update before clipping (one host synchronisation)."""
        return float(self.sqnorm.sqrt()) * getattr(self, "_gscale", 1.0)


def read_train_config(config_path) -> dict:
    """The keys of a reference experiment config the training loop uses (configs/best.yml; train.py:136-151).  Missing keys
    raise, as attribute access on the reference's config object does."""
    import yaml
    with open(config_path) as f:
        cfg = yaml.safe_load(f) or {}
    ta, ds, opt = cfg.get("training_args"), cfg.get("dataset_args"), (cfg.get("optimizer") or {}).get("params")
    for name, sec in (("training_args", ta), ("dataset_args", ds), ("optimizer.params", opt)):
        if not isinstance(sec, dict):
            raise KeyError(f"{config_path}: no `{name}` section")
    need = ("steps", "batch_size", "warmup_steps", "clip_grad", "dropout", "att_layers_num", "channels", "log_freq", "save_freq")
    missing = [k for k in need if k not in ta] + [f"dataset_args.{k}" for k in ("max_seq_len", "max_text_len") if k not in ds]
    if missing:
        raise KeyError(f"{config_path}: missing {', '.join(missing)}")
    ch = int(ta["channels"])
    return {"steps": int(ta["steps"]), "batch_size": int(ta["batch_size"]), "warmup": int(ta["warmup_steps"]),
            "clip_grad": None if ta["clip_grad"] is None else float(ta["clip_grad"]), "dropout": float(ta["dropout"]),
            "num_layers": int(ta["att_layers_num"]), "c1": ch, "c2": ch * 3 // 2, "c3": ch * 2, "log_freq": int(ta["log_freq"]),
            "save_freq": int(ta["save_freq"]), "L": int(ds["max_seq_len"]), "Lt": int(ds["max_text_len"]),
            "betas": tuple(float(b) for b in opt.get("betas", (0.9, 0.999))), "weight_decay": float(opt.get("weight_decay", 0.0)),
            "seed": int((cfg.get("experiment") or {}).get("seed", 0))}


def fit(config_path, data_path=None, out_dir="runs/exp", steps=None, init=None, seed=0, log=print, precision="fp32"):
    """The reference's ``TrainingLoop.train`` (train.py:84-134): updates until ``training_args.steps``, a log line every
    ``log_freq`` updates (mean losses since the last line), ``checkpoint_<n>.pth`` every ``save_freq``, ``model_final.pth`` at
    the end.  Cadence and numbering are the reference's own (train.py:111-126): after update number ``count`` it tests
    ``(count + 1) % freq`` and labels the line / file ``count + 1`` — so ``checkpoint_1000.pth`` holds 999 updates, and runs line
    up with the reference step for step.  One process per GPU under ``torch.distributed.run`` (LOCAL_RANK picks the device,
    gradients averaged by RCCL); rank 0 logs and saves.  ``training_args.batch_size`` is the PER-RANK batch (the reference is
    single-process, so its 96 is one GPU's batch): N ranks train on a global batch of N x batch_size — divide it in the config to
    keep the reference's global batch.  ``precision``: "fp32" (the reference's) or "bf16" (bf16-rounded GEMM operands, fp32
    accumulation and fp32 master weights / optimizer state).  Returns the trained ``TrainModel``."""
    import os
    import time
    from . import spec
    from .checkpoint import read_state_dict
    cfg = read_train_config(config_path)
    n_steps = steps if steps is not None else cfg["steps"]
    # the host side draws abar and assembles batches with small torch CPU ops: with one thread per VISIBLE core on a box whose
    # cgroup grants fewer, each of them takes milliseconds (31 vs 7 ms per update on the GPU box) — cap at the granted share
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        cores = os.cpu_count() if quota == "max" else max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        cores = os.cpu_count() or 1
    if torch.get_num_threads() > max(1, min(cores, 16)):
        torch.set_num_threads(max(1, min(cores, 16)))
    dist = torch.distributed
    world, rank = 1, 0
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device())
    B, L, Lt = cfg["batch_size"], cfg["L"], cfg["Lt"]
    if L % 8:
        raise ValueError("dataset_args.max_seq_len must be a multiple of 8 (configs/best.yml:12)")
    sd = read_state_dict(init) if init else spec.synthetic_state_dict(cfg["num_layers"], cfg["c1"], cfg["c2"], cfg["c3"], seed=seed)
    model = TrainModel(sd, num_layers=cfg["num_layers"], device=dev, drop_rate=cfg["dropout"], seed=seed, precision=precision)
    opt = Adam(model.parameters(), betas=cfg["betas"], weight_decay=cfg["weight_decay"], max_norm=cfg["clip_grad"] or 0.0)
    step_fn = GraphedTrainStep(model, opt, B, L, Lt, d_model=2 * cfg["c1"], warmup=cfg["warmup"], seed=seed)
    alpha_set = torch.cumprod(1 - (0.02 + torch.exp(torch.linspace(math.log(1e-5), math.log(0.4), 60))), dim=0)   # utils/nn.py:19-39
    gen = torch.Generator().manual_seed(cfg["seed"] * 1000 + rank)
    data = None
    if data_path:
        data = torch.load(data_path, map_location="cpu", weights_only=True)
        for k, shape in (("strokes", (L, 3)), ("text", (Lt,)), ("style", (14, 1280))):
            if k not in data or tuple(data[k].shape[1:]) != shape:
                raise ValueError(f"{data_path}: `{k}` must be [N, {', '.join(map(str, shape))}]")

    def next_batch():
        if data is not None:                      # a random batch, as `next(iter(shuffled loader))` gives (train.py:99)
            idx = torch.randint(0, data["strokes"].shape[0], (B,), generator=gen)
            return {k: data[k][idx] for k in ("strokes", "text", "style")}
        pen = (torch.rand(B, L, 1, generator=gen) < 0.1).float()
        return {"strokes": torch.cat([torch.randn(B, L, 2, generator=gen), pen], dim=-1),
                "text": torch.randint(1, 73, (B, Lt), generator=gen), "style": torch.randn(B, 14, 1280, generator=gen).abs()}

    os.makedirs(out_dir, exist_ok=True)
    acc, t0 = [], time.time()
    for count in range(1, n_steps + 1):
        acc.append(step_fn(next_batch(), alpha_set, count).clone())
        if (count + 1) % cfg["log_freq"] == 0:      # (train.py:111: the reference's own off-by-one, reproduced)
            m = torch.stack(acc).mean(0).tolist()   # (one host sync per log line)
            acc = []
            if rank == 0:
                log(f"Step {count + 1} | Loss: {m[0]:.3f} | Score: {m[1]:.3f} | Pen: {m[2]:.3f} | Time: {time.time() - t0:.3f} sec")
        if rank == 0 and (count + 1) % cfg["save_freq"] == 0:
            # save_checkpoint's form (checkpoint.py:244): {"meta", "state_dict"}; model_final.pth is the bare state_dict (train.py:131)
            torch.save({"meta": None, "state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}},
                       os.path.join(out_dir, f"checkpoint_{count + 1}.pth"))
    if rank == 0:
        torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, "model_final.pth"))
    if world > 1:
        dist.barrier()
    return model
