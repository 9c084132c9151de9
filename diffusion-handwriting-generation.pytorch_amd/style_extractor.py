"""``StyleExtractor`` — host-side mirror of the reference's style front end (reference text_style.py:11-59) over the C-ABI
HIP library (include/dhw_style.h): grey handwriting image(s) [B,1,H,W] -> writer-style features [B,14,1280], the
``style_vector`` input of ``DiffusionModel.forward`` / ``sample``.

The reference builds torchvision's MobileNetV2 with downloaded ImageNet weights.  There is no network here, so the weights
come from a local copy of that checkpoint (torchvision's ``mobilenet_v2-*.pth`` state_dict, read with
``torch.load(weights_only=True)``) or from a state_dict the caller passes; without either the extractor is random-initialised
and says so.  All arithmetic runs in libdhw_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib


class StyleExtractor:
    """Extracts style features from handwriting images using MobileNetV2 (MI355X).

    precision: "fp32" (default: the front end runs once per prompt and is <1 % of its FLOPs; exact-f32 MFMA) or "bf16"
    (bf16 activations and pointwise weights, fp32 accumulation)."""

    def __init__(self, weights=None, *, precision: str = "fp32", device: int | None = None):
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        if not torch.cuda.is_available():
            raise RuntimeError("StyleExtractor needs an MI355X (HIP device): there is no CPU path in this package")
        self.precision = precision
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._handle = C.c_void_p()
        l = _lib.lib()
        _lib.check(l.dhw_style_create(C.byref(self._handle), _lib.PREC_F32 if precision == "fp32" else _lib.PREC_BF16,
                                      self.device.index or 0), None, style=True)
        if weights is None:
            warnings.warn("StyleExtractor: no MobileNetV2 weights given (the reference downloads torchvision's ImageNet "
                          "checkpoint, unavailable offline): using a random initialisation")
            weights = self.random_state_dict()
        elif isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
            weights = torch.load(str(weights), map_location="cpu", weights_only=True)
        self.load_state_dict(weights)

    # torchvision's MobileNetV2 state_dict inventory (features.* only), as the library expects it
    def state_dict_keys(self):
        l = _lib.lib()
        out = []
        for i in range(_lib.check(l.dhw_style_num_keys(self._handle), self._handle, style=True)):
            key, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            l.dhw_style_key_info(self._handle, i, C.byref(key), shape, C.byref(nd))
            out.append((key.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def random_state_dict(self, seed: int = 0):
        """torch-default-like random init of every tensor (kaiming convolutions, BN weight 1 / bias 0 / mean 0 / var 1)."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shp in self.state_dict_keys():
            if k.endswith("running_var") or (k.endswith(".weight") and len(shp) == 1):
                sd[k] = torch.ones(shp)
            elif len(shp) == 1:
                sd[k] = torch.zeros(shp)
            else:
                fan_out = shp[0] * shp[2] * shp[3]
                sd[k] = torch.randn(shp, generator=g) * (2.0 / fan_out) ** 0.5   # torchvision: kaiming_normal_(mode="fan_out")
        return sd

    def load_state_dict(self, sd, strict: bool = True):
        l = _lib.lib()
        for k, v in sd.items():
            t = v.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            rc = l.dhw_style_load(self._handle, k.encode(), C.c_void_p(t.data_ptr()), _lib.DHW_F32, shape, t.dim())
            if rc < 0 and strict:
                _lib.check(rc, self._handle, style=True)
        _lib.check(l.dhw_style_finalize(self._handle), self._handle, style=True)   # names the first missing key

    def forward(self, img_batch) -> torch.Tensor:
        """img_batch: array / tensor [B,1,H,W] of grey levels 0..255 (text_style.py:50-51) -> [B,14,1280] fp32 on the device."""
        x = torch.as_tensor(np.asarray(img_batch) if not isinstance(img_batch, torch.Tensor) else img_batch, dtype=torch.float32)
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("img_batch must be [B, 1, H, W]")
        B, _, H, W = x.shape
        with torch.cuda.device(self.device):
            x = x.to(self.device).contiguous()
            out = torch.empty((B, 14, 1280), device=self.device, dtype=torch.float32)
            st = torch.cuda.current_stream(self.device)
            _lib.check(_lib.lib().dhw_style_forward(self._handle, x.data_ptr(), B, H, W, out.data_ptr(), C.c_void_p(st.cuda_stream)),
                       self._handle, style=True)
            x.record_stream(st)
        return out

    __call__ = forward

    def debug_features(self) -> torch.Tensor:
        """Feature map of the last forward, NHWC [B, H/32, W/32, 1280] fp32 (include/dhw_style.h test hook)."""
        shape = (C.c_int64 * 4)()
        buf = np.empty(64 * 1024 * 1024, np.float32)
        n = _lib.lib().dhw_style_debug_features(self._handle, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, shape)
        _lib.check(int(n), self._handle, style=True)
        return torch.from_numpy(buf[:n].reshape(shape[0], shape[1], shape[2], shape[3]).copy())

    def __del__(self):
        try:
            if self._handle:
                _lib.lib().dhw_style_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
