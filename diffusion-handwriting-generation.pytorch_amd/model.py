"""``DiffusionModel`` — host-side mirror of the reference's denoiser interface
(reference model.py:61-182) over the C-ABI HIP library.

Same constructor arguments, same ``forward(strokes, text, sigma, style_vector)
-> (eps, pen_lifts, None)`` contract, same 323-key ``state_dict`` (torch-native
layouts), so a reference checkpoint loads with ``load_state_dict`` and callers
(`inference.py:89`, `tests/test_model.py:21` in the reference) work unchanged.
All arithmetic runs in libdhw_hip.so on the GPU; PyTorch only owns the device
buffers and the stream.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _lib
from .spec import param_spec


class _Node(nn.Module):
    """Name-space container so parameters keep the reference's dotted keys."""


VOCAB = 73   # nn.Embedding(73, d_model), reference text_style.py:71


def check_token_ids(text: torch.Tensor, seen: list | None = None) -> None:
    """The reference's ``nn.Embedding`` raises IndexError for an id outside [0, 73); the kernels would clamp it.  One
    reduction + one host read — a blocking device sync when ``text`` lives on the GPU, so a caller that samples the same
    prompt tensor repeatedly passes ``seen`` (a small list it owns): a tensor object that was validated before and has not
    been written since (same object, same storage, same ``_version``) is not read again, and the next sampling call's launch
    does not wait for the previous one to drain.

    Caveat of the cache (exact semantics need ``seen=None``, which ``forward`` uses): ``_version`` counts in-place torch
    operations only — a write through ``.data``, through numpy memory shared with ``torch.from_numpy``, or by another library is
    invisible to it, and an out-of-range id written that way is clamped by the kernels instead of raising.  The entries are
    WEAK references (a validated prompt tensor is not kept alive, and a dead entry can never match a new tensor that reuses its
    address: the comparison is on the live object)."""
    import weakref
    if seen is not None:
        for ref, ptr, ver in seen:
            t = ref()
            if t is text and ptr == text.data_ptr() and ver == text._version:
                return
    if text.numel() and bool(((text < 0) | (text >= VOCAB)).any()):
        raise IndexError(f"index out of range in self: token ids must lie in [0, {VOCAB})")
    if seen is not None:
        seen[:] = [e for e in seen if e[0]() is not None]
        seen.append((weakref.ref(text), text.data_ptr(), text._version))
        del seen[:-4]


class _DifferentiableForward(torch.autograd.Function):
    """``DiffusionModel.forward`` under autograd (reference train.py:46-60: ``model(x, text, sigma, style)`` ... ``loss.backward()``):
    the forward and backward passes of ``train_model.TrainModel`` (hand-written fp32 HIP ops, include/dhw_train.h) as ONE autograd
    node whose inputs are the model's parameters.  The tape lives in the TrainModel, so a backward belongs to the model's LATEST
    differentiable forward (what a training loop does); an older graph raises instead of returning another batch's gradients."""

    @staticmethod
    def forward(ctx, model, strokes, text, sigma, style, *params):
        tm = model._trainer
        model._train_generation += 1
        if tm.drop_rate > 0.0:
            # A fresh Philox draw index per differentiable forward (and per data-parallel rank), as train_step / GraphedTrainStep set
            # one per update: the EncoderLayers' keep-masks are keyed by (seed, draw index, sample, site), and the reference draws a
            # fresh mask in every call (model.py:23 nn.Dropout).  Without this every train-mode call replayed draw 0's masks.
            dist = torch.distributed
            on = dist.is_available() and dist.is_initialized()
            rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
            tm.rng.copy_(torch.tensor([tm.seed, model._train_generation * world + rank], dtype=torch.int64))
        score, pen = tm.forward(strokes, text.to("cpu"), sigma.reshape(strokes.shape[0], 1), style)
        ctx.model, ctx.generation = model, model._train_generation
        return score.clone(), pen.clone()

    @staticmethod
    def backward(ctx, d_score, d_pen):
        model = ctx.model
        tm = model._trainer
        if tm is None or tm.tape is None or ctx.generation != model._train_generation:
            raise RuntimeError("DiffusionModel: backward through a forward that is not the model's latest one (one graph at a time)")
        tm.zero_grad()
        tm.backward(d_score.to(tm.dev, torch.float32), d_pen.to(tm.dev, torch.float32))
        return (None, None, None, None, None) + tuple(tm.p[k].g.clone() for k in tm.names)


def _torch_dtype_code(t: torch.Tensor) -> int:
    return {torch.float32: _lib.DHW_F32, torch.bfloat16: _lib.DHW_BF16, torch.float16: _lib.DHW_F16,
            torch.float64: _lib.DHW_F64}[t.dtype]


class DiffusionModel(nn.Module):
    """Diffusion denoiser conditioned on text and style (inference path, MI355X)."""

    def __init__(self, num_layers: int = 4, c1: int = 128, c2: int = 192, c3: int = 256, drop_rate: float = 0.1,
                 *, precision: str = "bf16", max_B: int = 64, max_L: int = 1000, max_Lt: int = 64, style_rows: int = 14):
        super().__init__()
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.num_layers, self.c1, self.c2, self.c3 = num_layers, c1, c2, c3
        self.drop_rate = drop_rate  # kept for signature parity; dropout is identity on the sampling path (eval)
        self.precision = precision
        self._cap = dict(max_B=max_B, max_L=max_L, max_Lt=max_Lt, S=style_rows)
        self._handle = None
        self._handle_dev = None
        self._wtoken = None
        self._trainer = None          # train_model.TrainModel behind the differentiable forward (created on first use)
        self._train_generation = 0
        for name, shape, kind in param_spec(num_layers, c1, c2, c3):
            node = self
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(self._init(shape, kind), requires_grad=False))

    # torch-default-like init (nn.Linear / nn.Conv1d: U(+-1/sqrt(fan_in)); nn.Embedding: N(0,1);
    # gamma_emb.bias = 1, conditioning.py:13)
    @staticmethod
    def _init(shape, kind):
        if kind == "ones":
            return torch.ones(shape)
        if kind == "normal":
            return torch.randn(shape)
        fan_in = math.prod(shape[1:]) if kind in ("linear_w", "conv_w") else int(kind.split(":")[1])
        bound = 1.0 / math.sqrt(fan_in)
        return torch.empty(shape).uniform_(-bound, bound)

    # ------------------------------------------------------------------ handle management
    def _destroy(self):
        if self._handle is not None:
            _lib.lib().dhw_destroy(self._handle)
            self._handle = None
            self._wtoken = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _device(self, *tensors) -> torch.device:
        for t in tensors:
            if t.is_cuda:
                return t.device
        p = next(self.parameters())
        if p.is_cuda:
            return p.device
        if not torch.cuda.is_available():
            raise RuntimeError("DiffusionModel needs an MI355X (HIP device): there is no CPU path in this package")
        return torch.device("cuda", torch.cuda.current_device())

    def _ensure_handle(self, dev: torch.device, B: int, L: int, Lt: int, S: int):
        cap = self._cap
        grow = B > cap["max_B"] or L > cap["max_L"] or Lt > cap["max_Lt"] or S != cap["S"]
        if self._handle is not None and (grow or self._handle_dev != dev):
            self._destroy()
        if self._handle is None:
            cap.update(max_B=max(B, cap["max_B"]), max_L=max(L, cap["max_L"]), max_Lt=max(Lt, cap["max_Lt"]), S=S)
            dims = _lib.DhwDims(self.num_layers, self.c1, self.c2, self.c3, cap["max_B"], cap["max_L"], cap["max_Lt"],
                                cap["S"], _lib.PREC_F32 if self.precision == "fp32" else _lib.PREC_BF16)
            h = C.c_void_p()
            _lib.check(_lib.lib().dhw_create(C.byref(h), C.byref(dims), dev.index or 0))
            self._handle, self._handle_dev = h, dev
        token = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if token != self._wtoken:
            self._push_weights()
            self._wtoken = token
        return self._handle

    def _push_weights(self):
        l = _lib.lib()
        for k, v in self.state_dict().items():
            t = v.detach().to("cpu").contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(l.dhw_load(self._handle, k.encode(), C.c_void_p(t.data_ptr()), _torch_dtype_code(t), shape, t.dim()),
                       self._handle)
        _lib.check(l.dhw_finalize(self._handle), self._handle)

    # ------------------------------------------------------------------ training: the same object under autograd
    def train(self, mode: bool = True):
        """``model.train()`` also marks the parameters as requiring gradients, the state a freshly constructed reference model is
        in (train.py:26-45 goes straight from the constructor to the optimizer); the constructor itself leaves them frozen so that
        plain inference calls never record a graph."""
        super().train(mode)
        if mode:
            self.requires_grad_(True)
        return self

    def _link_trainer(self, dev: torch.device):
        """Make every parameter a VIEW of the TrainModel's flat fp32 buffer (Conv1d weights: the permuted view of its
        [tap][Cout][Cin] storage), so an optimizer's in-place updates are the training kernels' weights without a copy, and
        re-link after anything that re-allocated the parameters (``.to()``, ``.cuda()``)."""
        from . import train_model as tm_mod
        tm = self._trainer
        if tm is None or tm.dev != dev:
            tm = self._trainer = tm_mod.TrainModel({k: v.detach() for k, v in self.state_dict().items()}, self.num_layers, device=dev,
                                                   drop_rate=0.0, precision="fp32")
        for k, p in self.named_parameters():
            view = tm.p[k].d
            if p.data_ptr() != view.data_ptr() or p.device != view.device:
                view.copy_(p.detach().to(dev, torch.float32))
                p.data = view
        tm.drop_rate = float(self.drop_rate) if self.training else 0.0      # EncoderLayer dropout (model.py:23)
        tm.style_drop = tm.STYLE_DROP if self.training else 0.0             # TextStyleEncoder's Dropout(0.3) (text_style.py:88)
        return tm

    # ------------------------------------------------------------------ forward == reference model.py:121-182
    def forward(self, strokes: torch.Tensor, text: torch.Tensor, sigma: torch.Tensor, style_vector: torch.Tensor):
        """strokes [B,T,2], text int [B,Lt] (0 = pad), sigma [B,1] or [B,1,1], style_vector [B,S,1280]
        -> (eps [B,T,2] fp32, pen_lifts [B,T] fp32 in (0,1), None).

        With autograd recording and parameters that require gradients (after ``model.train()`` / ``requires_grad_(True)``) the
        outputs carry a graph back to the parameters: the call runs the fp32 training kernels (``train_model.TrainModel``) with
        dropout as ``self.training`` says, and ``loss.backward()`` fills ``p.grad`` as the reference's would (train.py:46-60).
        Gradients with respect to the INPUTS are not produced (the reference's training loop does not use them)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if strokes.dim() != 3 or strokes.shape[-1] != 2 or strokes.shape[1] % 8:
                raise ValueError("strokes must be [B, T, 2] with T a multiple of 8")
            check_token_ids(text)
            dev = self._device(strokes, text, sigma, style_vector)
            with torch.cuda.device(dev):
                tm = self._link_trainer(dev)
                named = dict(self.named_parameters())
                eps, pen = _DifferentiableForward.apply(self, strokes, text, sigma, style_vector, *(named[k] for k in tm.names))
            return eps.to(strokes.device), pen.to(strokes.device), None
        if strokes.dim() != 3 or strokes.shape[-1] != 2:
            raise ValueError("strokes must be [B, T, 2]")
        B, L, _ = strokes.shape
        if L % 8:
            raise ValueError("T must be a multiple of 8 (three 2x down/up-sampling levels, reference model.py:169-175)")
        if style_vector.dim() != 3 or (style_vector.shape[1] * style_vector.shape[2]) % 256 or style_vector.shape[2] != 1280:
            raise ValueError("style_vector must be [B, S, 1280]")
        if sigma.numel() != B:
            raise ValueError("sigma must hold one value per sample ([B,1] or [B,1,1])")
        check_token_ids(text)
        ret_dev = strokes.device
        dev = self._device(strokes, text, sigma, style_vector)
        h = self._ensure_handle(dev, B, L, text.shape[1], style_vector.shape[1])
        with torch.cuda.device(dev):
            s = strokes.to(dev, torch.float32).contiguous()
            t = text.to(dev, torch.int64).contiguous()
            sg = sigma.to(dev, torch.float32).reshape(B).contiguous()
            sv = style_vector.to(dev, torch.float32).contiguous()
            eps = torch.empty((B, L, 2), device=dev, dtype=torch.float32)
            pen = torch.empty((B, L), device=dev, dtype=torch.float32)
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.lib().dhw_forward(h, s.data_ptr(), t.data_ptr(), sg.data_ptr(), sv.data_ptr(), B, L,
                                              t.shape[1], eps.data_ptr(), pen.data_ptr(), C.c_void_p(st)), h)
            # keep the inputs alive until the stream has consumed them
            for x in (s, t, sg, sv):
                x.record_stream(torch.cuda.current_stream(dev))
        return eps.to(ret_dev), pen.to(ret_dev), None

    # ------------------------------------------------------------------ test / measurement hooks (include/dhw_debug.h)
    def debug_read(self, name: str) -> torch.Tensor:
        import numpy as np
        shape = (C.c_int64 * 3)()
        cap = self._cap
        buf = np.empty(cap["max_B"] * max(cap["max_L"], cap["max_Lt"], cap["S"] * 5) * 768, np.float32)
        n = _lib.lib().dhw_debug_read(self._handle, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, shape)
        _lib.check(int(n), self._handle)
        return torch.from_numpy(buf[:n].reshape(shape[0], shape[1], shape[2]).copy())

    def debug_randn(self, seed: int, first_sample: int, B: int, L: int, it: int = -1) -> torch.Tensor:
        """The device generator's N(0,1) draws [B,L,2] for sampler iteration `it` (-1 = x_T): include/dhw_debug.h."""
        import numpy as np
        dev = self._device()
        h = self._ensure_handle(dev, B, L, 1, self._cap["S"])
        buf = np.empty((B, L, 2), np.float32)
        _lib.check(_lib.lib().dhw_debug_randn(h, seed, first_sample, B, L, it, buf.ctypes.data_as(C.POINTER(C.c_float))), h)
        return torch.from_numpy(buf)

    def set_teacher(self, reset: "torch.Tensor | None", every: int = 0) -> "torch.Tensor | None":
        """Teacher forcing of the next sample() calls (include/dhw_debug.h, tests only): `reset` [K,B,L,2] on the model's
        device; returns the capture tensor [K,B,L,2] the library fills (x after k*every steps), or None when switched off.
        Takes effect when the next sample() call has its handle (handles are created lazily, per problem size)."""
        if reset is None or every <= 0:
            self._teacher = (None, None, 0)
            return None
        reset = reset.to(torch.float32).contiguous()
        cap = torch.empty_like(reset)
        self._teacher = (reset, cap, every)   # (keeps the buffers alive)
        return cap

    def _apply_teacher(self):
        t = getattr(self, "_teacher", None)
        if t is not None:
            reset, cap, every = t
            _lib.check(_lib.lib().dhw_debug_set_teacher(self._handle, reset.data_ptr() if every else None, cap.data_ptr() if every else None, every),
                       self._handle)

    def persistent_plans(self) -> int:
        """Number of `sample` shapes that ran every denoiser call as ONE persistent launch (csrc/persist.h; 0 = kernel by kernel)."""
        return int(_lib.lib().dhw_debug_persist_plans(self._handle)) if self._handle else 0

    def profile(self, on: bool):
        _lib.check(_lib.lib().dhw_profile_enable(self._handle, int(on)), self._handle)
        if on:
            _lib.check(_lib.lib().dhw_profile_reset(self._handle), self._handle)

    def profile_results(self):
        l = _lib.lib()
        out = []
        for i in range(_lib.check(l.dhw_profile_count(self._handle), self._handle)):
            lab, ms, n, fl, by = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            l.dhw_profile_get(self._handle, i, C.byref(lab), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
            out.append(dict(label=lab.value.decode(), total_ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value))
        return out

    def attention_time(self, layer: int, iters: int = 20):
        """(us with, us without, FLOPs) of EncoderLayer ``layer``'s self-attention stage on the last forward's activations
        (include/dhw_debug.h dhw_debug_attention_time: the product's enc_bc_kernel with and without the stage)."""
        a, b, fl = C.c_double(), C.c_double(), C.c_double()
        st = torch.cuda.current_stream(self._handle_dev).cuda_stream
        _lib.check(_lib.lib().dhw_debug_attention_time(self._handle, layer, iters, C.byref(a), C.byref(b), C.byref(fl), C.c_void_p(st)), self._handle)
        return a.value, b.value, fl.value

    def work(self, L: int, Lt: int):
        """(FLOPs, block-boundary bytes) of one denoiser call per sample."""
        fl, by = C.c_double(), C.c_double()
        _lib.check(_lib.lib().dhw_work(self._handle, L, Lt, C.byref(fl), C.byref(by)), self._handle)
        return fl.value, by.value


# BASELINE.json's north_star calls the class DiffusionWriter; the reference's symbol is DiffusionModel.
DiffusionWriter = DiffusionModel
