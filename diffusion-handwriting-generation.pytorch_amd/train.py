"""Pieces of the training step (BASELINE configs[4]; reference train.py:26-67).

HIP (include/dhw_train.h): the forward-diffusion perturbation, ``loss_fn`` with its gradient, global gradient-norm clipping +
Adam on flat buffers, ConvBlock forward + backward as one call.  Host logic mirrored here: ``get_alphas`` (the reference's
torch RNG calls, in its order), the Noam learning-rate schedule, and the data-parallel gradient all-reduce
(``torch.distributed``: RCCL over xGMI on the GPU ranks).  The whole model's forward / backward and the complete update
are in ``train_model.py``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _tcheck(code: int):
    if code < 0:
        raise _lib.DhwError(code, (_lib.lib().dhw_train_last_error() or b"?").decode())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.to(dev, torch.float32).contiguous()


def get_alphas(batch_size: int, alpha_set: torch.Tensor) -> torch.Tensor:
    """abar ~ U(abar_i, abar_{i+1}) for a random schedule interval per sample (reference utils/nn.py:42-61: one
    ``torch.randint`` then one ``torch.rand`` on torch's global CPU generator — the same calls in the same order, so a seeded
    run draws what the reference draws)."""
    idx = torch.randint(low=0, high=len(alpha_set) - 1, size=(batch_size, 1), dtype=torch.int64)
    lower, upper = alpha_set[idx], alpha_set[idx + 1]
    return torch.rand(lower.shape) * (upper - lower) + lower


def noam_lr(step: int, d_model: int = 256, n_warmup_steps: int = 10000, lr_mul: float = 1.0) -> float:
    """Learning rate of update number ``step`` >= 1 (reference scheduler.py:16-29, d_model = 2 * channels, train.py:150-155)."""
    return lr_mul * (d_model ** -0.5) * min(step ** (-0.5), step * n_warmup_steps ** (-1.5))


def perturb(x: torch.Tensor, eps: torch.Tensor, alphas: torch.Tensor) -> torch.Tensor:
    """x_perturbed = sqrt(abar) x + sqrt(1 - abar) eps (train.py:41-43); x, eps [B,L,2], alphas [B,1] -> device tensor."""
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    B, L, _ = x.shape
    xd, ed, ad = _f32(x, dev), _f32(eps, dev), _f32(alphas.reshape(B), dev)
    out = torch.empty_like(xd)
    _tcheck(_lib.lib().dhw_train_perturb(xd.data_ptr(), ed.data_ptr(), ad.data_ptr(), B, L, out.data_ptr(), _stream(dev)))
    return out


def loss_fn(eps, score_pred, pen_lifts, pen_lifts_pred, alphas, with_grad: bool = True):
    """``loss_fn`` of the reference (loss.py:5-37) -> (loss, score_loss, pen_lifts_loss) as a device tensor [3] and, with
    ``with_grad``, (d loss / d score_pred [B,L,2], d loss / d pen_lifts_pred [B,L]) — what ``loss.backward()`` hands the model."""
    dev = score_pred.device if score_pred.is_cuda else torch.device("cuda", torch.cuda.current_device())
    B, L, _ = score_pred.shape
    e, s, p, pp, a = (_f32(t, dev) for t in (eps, score_pred, pen_lifts, pen_lifts_pred, alphas.reshape(B)))
    out = torch.empty(3, device=dev)
    ds = torch.empty_like(s) if with_grad else None
    dp = torch.empty_like(pp) if with_grad else None
    _tcheck(_lib.lib().dhw_train_loss(e.data_ptr(), s.data_ptr(), p.data_ptr(), pp.data_ptr(), a.data_ptr(), B, L, out.data_ptr(),
                                      ds.data_ptr() if with_grad else None, dp.data_ptr() if with_grad else None, _stream(dev)))
    return (out, ds, dp) if with_grad else out


class Adam:
    """torch.optim.Adam(lr, betas, eps, weight_decay) + clip_grad_norm_(max_norm) over a list of flat fp32 device
    buffers (configs/best.yml:33-38, clip_grad 100, utils/clip_grad.py:42-43), one fused kernel per buffer."""

    def __init__(self, params, betas=(0.9, 0.98), eps: float = 1e-8, weight_decay: float = 1e-5, max_norm: float = 100.0):
        self.params = list(params)
        self.betas, self.eps, self.weight_decay, self.max_norm = betas, eps, weight_decay, max_norm
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.step_count = 0

    def step(self, grads, lr: float):
        self.step_count += 1
        n = len(self.params)
        dev = self.params[0].device
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        sizes = (C.c_int64 * n)(*[p.numel() for p in self.params])
        gn = torch.zeros(1, device=dev)
        _tcheck(_lib.lib().dhw_train_adam(n, arr(self.params), arr(grads), arr(self.m), arr(self.v), sizes, lr, self.betas[0], self.betas[1],
                                          self.eps, self.weight_decay, self.step_count, self.max_norm, gn.data_ptr(), _stream(dev)))
        return gn.sqrt()


    def hyper(self, lr: float, grad_scale: float = 1.0) -> list:
        """Advance the update counter and return the 9 scalars ``dhw_train_adam_dev`` reads from device memory.  ``grad_scale``:
        1 / world size when the gradient buffer holds the SUM over ranks (``allreduce_grads(..., average=False)``)."""
        self.step_count += 1
        b1, b2 = self.betas
        return [lr, b1, b2, self.eps, self.weight_decay, 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count, self.max_norm, grad_scale]

    def step_dev(self, grads, hyper_dev: torch.Tensor, sqnorm_dev: torch.Tensor):
        """The update with every scalar on the device (``hyper_dev`` = the 9 floats of ``hyper``): no allocation and no
        host synchronisation, so it can sit inside a captured hipGraph.  ``sqnorm_dev`` receives ||g||^2."""
        n = len(self.params)
        dev = self.params[0].device
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        sizes = (C.c_int64 * n)(*[p.numel() for p in self.params])
        _tcheck(_lib.lib().dhw_train_adam_dev(n, arr(self.params), arr(grads), arr(self.m), arr(self.v), sizes, hyper_dev.data_ptr(),
                                              sqnorm_dev.data_ptr(), _stream(dev)))


def allreduce_grads(flat_grads, world_size: int | None = None, average: bool = True):
    """Data-parallel gradient averaging (BASELINE configs[4]: DDP over 8 GPUs): ONE all-reduce of each flat gradient buffer
    (the whole model is 40.1 MB of fp32 gradients — a single bucket per buffer keeps the ring collective bandwidth-bound on
    the per-link xGMI rate instead of latency-bound), then 1 / world_size.  backend "nccl" is RCCL on ROCm; gloo on CPU."""
    import torch.distributed as dist
    ws = world_size or dist.get_world_size()
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in flat_grads]
    for w in works:
        w.wait()
    if average:          # (``average=False``: the caller folds 1 / world size into its optimizer — Adam.hyper(grad_scale=...))
        for g in flat_grads:
            g.div_(ws)
    return flat_grads


class GradBucketReducer:
    """The data-parallel gradient all-reduce in BUCKETS, overlapped with the tail of the backward pass (reference: single
    process, train.py:53-60 — nothing to mirror; BASELINE configs[4] / north_star: "RCCL grad all-reduce over xGMI").

    ``flat_grad`` is the trainer's one flat gradient buffer, laid out in the ORDER IN WHICH THE BACKWARD SWEEP COMPLETES the
    gradients (train_model.TrainModel orders it so: decoder + skip convolutions, bottleneck layers, encoder, text side + FiLM
    Linears, sigma MLP), and ``ranges[i]`` = [start, end) of bucket i in it.  ``launch(i)`` is called the moment bucket i's last
    weight-gradient kernel has been enqueued (eagerly: from the tape's bucket markers; under graph replay: after the graph segment
    that ends at the marker): one asynchronous SUM all-reduce of that contiguous range.  With backend "nccl" (= RCCL) the
    collective runs on the process group's own stream behind an event on the launching stream, i.e. beside the kernels the main
    stream enqueues next — the rest of the backward; ``wait()`` makes the launching stream wait for all of them (and, for host-side
    backends such as gloo, blocks until they are done).  1 / world size is left to the caller (Adam's grad_scale, or ``average``).

    Bucket sizes (fp32, num_layers = 2): 6.4 / 12.2 / 6.9 / 11.3 / 0.3 MB — each a bandwidth-bound ring message on the per-link xGMI
    rate; the last two cannot overlap with anything (the text side is the first thing in the forward, so its gradients are the last
    to complete).  The 8-rank overlap itself is unmeasured here (one-GPU boxes): what is tested is that the bucketed result equals the
    single flat all-reduce (2 ranks, gloo) and that every bucket's gradients are final when its marker fires (GPU test)."""

    def __init__(self, flat_grad: torch.Tensor, ranges, reduce_one_rank: bool | None = None):
        """``reduce_one_rank``: issue the collectives even in a one-rank group, where every all-reduce is the identity (default: env
        DHW_TRAIN_REDUCE_ONE_RANK=1, else skip them).  RCCL executes a one-rank all-reduce as a real copy kernel — 0.28 ms for the
        40 MB buffer on MI355X (profiles/r05_train_dist_overhead.log) — which a product run has no reason to pay; the RCCL rehearsal
        on a one-GPU box (tests/test_gpu_train.py, bench.py --force-dist) forces it."""
        import os
        self.reduce_one_rank = (os.environ.get("DHW_TRAIN_REDUCE_ONE_RANK", "0") == "1") if reduce_one_rank is None else bool(reduce_one_rank)
        self.flat_grad, self.ranges = flat_grad, [tuple(r) for r in ranges]
        if any(a >= b for a, b in self.ranges) or any(self.ranges[i][1] != self.ranges[i + 1][0] for i in range(len(self.ranges) - 1)) \
                or self.ranges[0][0] != 0 or self.ranges[-1][1] != flat_grad.numel():
            raise ValueError("bucket ranges must tile the flat gradient buffer in order")
        self.works, self.launched = [], []

    def launch(self, i: int):
        import torch.distributed as dist
        if i in self.launched:
            raise RuntimeError(f"gradient bucket {i} reduced twice in one update")
        a, b = self.ranges[i]
        self.launched.append(i)
        if dist.get_world_size() == 1 and not self.reduce_one_rank:
            return                      # the SUM over one rank is the buffer itself
        self.works.append(dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def wait(self, average: bool = False, world_size: int | None = None):
        import torch.distributed as dist
        if sorted(self.launched) != list(range(len(self.ranges))):
            raise RuntimeError(f"gradient buckets reduced this update: {sorted(self.launched)} of {len(self.ranges)}")
        for w in self.works:
            w.wait()
        self.works, self.launched = [], []
        if average:
            self.flat_grad.div_(world_size or dist.get_world_size())
        return self.flat_grad


_CB_FIELDS = ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "fc_w", "fc_b", "skip_w", "skip_b", "film_w", "film_b")


def convblock_forward_backward(sd: dict, x: torch.Tensor, sigma: torch.Tensor, dout: torch.Tensor):
    """ConvBlock (cnn.py:64-87) forward + autograd backward in HIP.  ``sd``: the block's state_dict (reference key names:
    conv1.weight, affine1.gamma_emb.weight, ...); x [B,Cin,L], dout [B,C,L] C-first as the reference module takes them;
    sigma [B,32].  Returns (out [B,C,L], dx [B,Cin,L], dsigma [B,32], {reference parameter name: gradient})."""
    dev = torch.device("cuda", torch.cuda.current_device())
    B, cin, L = x.shape
    cout = dout.shape[1]
    c1 = cout // 2
    host = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in sd.items()}
    film_w = torch.cat([host[f"affine{i}.gamma_emb.weight"] for i in (1, 2, 3)] + [host[f"affine{i}.beta_emb.weight"] for i in (1, 2, 3)]).contiguous()
    film_b = torch.cat([host[f"affine{i}.gamma_emb.bias"] for i in (1, 2, 3)] + [host[f"affine{i}.beta_emb.bias"] for i in (1, 2, 3)]).contiguous()
    hw = {"conv1_w": host["conv1.weight"], "conv1_b": host["conv1.bias"], "conv2_w": host["conv2.weight"], "conv2_b": host["conv2.bias"],
          "fc_w": host["fc.weight"], "fc_b": host["fc.bias"], "skip_w": host["conv_skip.weight"], "skip_b": host["conv_skip.bias"],
          "film_w": film_w, "film_b": film_b}
    w = _lib.ConvBlockWeights(**{k: hw[k].data_ptr() for k in _CB_FIELDS})
    gd = {k: torch.zeros(hw[k].shape, device=dev) for k in _CB_FIELDS}
    g = _lib.ConvBlockWeights(**{k: gd[k].data_ptr() for k in _CB_FIELDS})
    xd = _f32(x.permute(0, 2, 1).reshape(B * L, cin), dev)
    dd = _f32(dout.permute(0, 2, 1).reshape(B * L, cout), dev)
    sg = _f32(sigma.reshape(B, 32), dev)
    out = torch.empty(B * L, cout, device=dev)
    dx = torch.empty(B * L, cin, device=dev)
    dsig = torch.empty(B, 32, device=dev)
    _tcheck(_lib.lib().dhw_train_convblock(dev.index or 0, B, L, cin, cout, xd.data_ptr(), sg.data_ptr(), dd.data_ptr(), C.byref(w), out.data_ptr(),
                                           dx.data_ptr(), dsig.data_ptr(), C.byref(g), _stream(dev)))
    tot = c1 + 2 * cout
    offs = {"affine1": (0, c1), "affine2": (c1, cout), "affine3": (c1 + cout, cout)}
    grads = {"conv1.weight": gd["conv1_w"], "conv1.bias": gd["conv1_b"], "conv2.weight": gd["conv2_w"], "conv2.bias": gd["conv2_b"],
             "fc.weight": gd["fc_w"], "fc.bias": gd["fc_b"], "conv_skip.weight": gd["skip_w"], "conv_skip.bias": gd["skip_b"]}
    for name, (o, n) in offs.items():
        grads[f"{name}.gamma_emb.weight"] = gd["film_w"][o:o + n]
        grads[f"{name}.gamma_emb.bias"] = gd["film_b"][o:o + n]
        grads[f"{name}.beta_emb.weight"] = gd["film_w"][tot + o:tot + o + n]
        grads[f"{name}.beta_emb.bias"] = gd["film_b"][tot + o:tot + o + n]
    return (out.reshape(B, L, cout).permute(0, 2, 1), dx.reshape(B, L, cin).permute(0, 2, 1), dsig, grads)
