"""The reverse-diffusion sampler: ``sample`` is the body of the reference's ``infer``
loop (reference inference.py:80-96) for a whole prompt batch; ``infer`` wraps it with
the tokenizer and the stroke-length heuristic (inference.py:65-78)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .model import DiffusionModel, check_token_ids
from .tokenizer import Tokenizer, stroke_length


def get_beta_set(T: int = 60) -> torch.Tensor:
    """beta_i = 0.02 + exp(linspace(ln 1e-5, ln 0.4, T)) (reference utils/nn.py:19-39); host-only."""
    return torch.from_numpy(_lib.schedule(T)[0])


def get_alpha_set(T: int = 60) -> torch.Tensor:
    """abar = cumprod(1 - beta) (reference inference.py:81)."""
    return torch.from_numpy(_lib.schedule(T)[1])


def sample(model: DiffusionModel, text: torch.Tensor, style_vector: torch.Tensor, L: int | None = None, T: int = 60,
           diffusion_mode: str = "new", noise: torch.Tensor | None = None, seed: int = 0,
           first_sample: int = 0) -> torch.Tensor:
    """Reverse-sample a batch.  text int [B,Lt] (0 = pad), style_vector [B,S,1280] -> [B,L,3] = (dx, dy, pen).

    ``noise`` (optional, f32 [T+1,B,L,2]): noise[0] = x_T, noise[1+k] = the N(0,1) draw of the k-th loop
    iteration (the reference draws them from torch's global RNG, inference.py:82 / utils/nn.py:86,111).
    Without it the library draws from a counter-based generator keyed by (seed, first_sample + b,
    iteration, position), so any sharding of a prompt batch over GPUs yields the same samples.
    """
    if diffusion_mode not in ("new", "standard"):
        raise ValueError("diffusion_mode must be 'new' or 'standard'")
    B, Lt = text.shape
    if L is None:
        L = stroke_length(Lt)
    if L % 8:
        raise ValueError("L must be a multiple of 8")
    if not hasattr(model, "_validated_text"):
        model._validated_text = []
    check_token_ids(text, model._validated_text)   # (once per prompt tensor: the check is a host read)
    dev = model._device(text, style_vector)
    h = model._ensure_handle(dev, B, L, Lt, style_vector.shape[1])
    model._apply_teacher()
    ret_dev = text.device
    with torch.cuda.device(dev):
        t = text.to(dev, torch.int64).contiguous()
        sv = style_vector.to(dev, torch.float32).contiguous()
        nz = None
        if noise is not None:
            if tuple(noise.shape) != (T + 1, B, L, 2):
                raise ValueError(f"noise must be [T+1,B,L,2] = {(T + 1, B, L, 2)}")
            nz = noise.to(dev, torch.float32).contiguous()
        out = torch.empty((B, L, 3), device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev)
        _lib.check(_lib.lib().dhw_sample(h, t.data_ptr(), sv.data_ptr(), B, L, Lt, T,
                                         0 if diffusion_mode == "new" else 1,
                                         nz.data_ptr() if nz is not None else None, seed, first_sample,
                                         out.data_ptr(), C.c_void_p(stream.cuda_stream)), h)
        # the library's graph holds raw pointers: pin the inputs to the model so they outlive the launch
        model._last_sample_inputs = (t, sv, nz, out)
    return out.to(ret_dev)


def infer(prompt: str, style_vector: torch.Tensor, model: DiffusionModel, diffusion_mode: str = "new", T: int = 60,
          seed: int = 0) -> np.ndarray:
    """Single-prompt convenience wrapper with the reference's front end: tokenise, L = 16 per token rounded up
    to a multiple of 8, sample, return the [L,3] stroke array that the reference hands to ``show_strokes``."""
    ids = Tokenizer().encode(prompt)
    text = torch.tensor([ids], dtype=torch.int64)
    out = sample(model, text, style_vector, L=stroke_length(len(ids)), T=T, diffusion_mode=diffusion_mode, seed=seed)
    return out[0].detach().cpu().numpy()


def remove_whitespace(img: np.ndarray, thresh: float) -> np.ndarray:
    """Crop to the rows / columns that hold a pixel darker than `thresh` (reference utils/preprocessing.py:47-62, the
    `remove_middle=False` branch, including its exclusive upper bounds: the last inked row and column are dropped)."""
    rows = np.nonzero(np.amin(img, axis=1) < thresh)[0]
    cols = np.nonzero(np.amin(img, axis=0) < thresh)[0]
    return img[rows[0]:rows[-1], cols[0]:cols[-1]]


def _resize_cubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """uint8 bicubic resize with OpenCV's INTER_CUBIC conventions (Keys kernel a = -0.75, pixel centres aligned by
    (dst + 0.5) * scale - 0.5, replicated border, no antialiasing), separable, float accumulation, round + saturate.
    PARITY UNPINNED: cv2 is not importable here, its fixed-point rounding may differ in the last grey level."""
    def taps(n_in, n_out):
        x = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        x0 = np.floor(x).astype(np.int64)
        t = x - x0
        a = -0.75
        w = np.stack([((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a,
                      ((a + 2) * t - (a + 3)) * t * t + 1,
                      ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1,
                      ((a * (2 - t) - 5 * a) * (2 - t) + 8 * a) * (2 - t) - 4 * a], axis=1)
        idx = np.clip(x0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
        return idx, w
    f = img.astype(np.float64)
    idx, w = taps(f.shape[1], out_w)
    f = (f[:, idx] * w[None]).sum(-1)
    idx, w = taps(f.shape[0], out_h)
    f = (f[idx] * w[:, :, None]).sum(1)
    return np.clip(np.rint(f), 0, 255).astype(np.uint8)


def read_img(path, height: int = 96) -> np.ndarray:
    """Load a handwriting image as grey levels, crop the white margins and resize to `height` rows keeping the aspect
    ratio (reference utils/io.py:98-115: cv2.imread(GRAYSCALE) -> remove_whitespace(thresh=127) -> cv2.resize(INTER_CUBIC))."""
    from PIL import Image
    img = np.asarray(Image.open(str(path)).convert("L"))
    img = remove_whitespace(img, thresh=127)
    h, w = img.shape
    return _resize_cubic(img, height * w // h, height)


_IMAGE_SUFFIXES = (".png", ".tif", ".tiff", ".jpg", ".jpeg", ".bmp", ".gif")
_extractors = {}


def load_style(source, style_weights=None) -> torch.Tensor:
    """The writer-style features of a prompt, [1,S,1280].  `source` is what the reference's `infer` takes — the path of a
    handwriting image, run through `read_img(source, 96)` and the StyleExtractor (inference.py:66-70, text_style.py:43-59;
    `style_weights` = a local torchvision MobileNetV2 state_dict file, without it the extractor is random-initialised) —
    or features computed elsewhere: a tensor / array, or a file holding one (.npy, or .pt read with weights_only=True)."""
    if isinstance(source, (str, bytes)) or hasattr(source, "__fspath__"):
        path = str(source)
        if path.lower().endswith(_IMAGE_SUFFIXES):
            from .style_extractor import StyleExtractor
            key = str(style_weights)
            if key not in _extractors:
                _extractors[key] = StyleExtractor(style_weights, precision="fp32")
            return _extractors[key](read_img(path, 96)[None, None, :])
        if path.endswith(".npy"):
            source = np.load(path, allow_pickle=False)
        elif path.endswith((".pt", ".pth")):
            source = torch.load(path, map_location="cpu", weights_only=True)
        else:
            raise ValueError(f"{path}: pass a handwriting image ({', '.join(_IMAGE_SUFFIXES)}) or style features ([S,1280], .npy / .pt)")
    sv = torch.as_tensor(source, dtype=torch.float32)
    if sv.dim() == 2:
        sv = sv[None]
    if sv.dim() != 3 or sv.shape[0] != 1 or sv.shape[2] != 1280:
        raise ValueError(f"style features must be [S,1280] or [1,S,1280], got {tuple(sv.shape)}")
    return sv


def infer_file(prompt: str, source, config_path: str | None = None, checkpoint_path: str | None = None,
               experiment_path: str | None = None, output: str = "result", diffusion_mode: str = "new", *, precision: str = "bf16",
               seed: int = 0, render: bool = True, style_weights: str | None = None) -> np.ndarray:
    """The reference's command-line entry (inference.py:19-27) around this build's sampler: resolve config / checkpoint
    (directly or inside ``experiment_path``), load the model, sample one prompt, write ``./<output>.png``.
    Returns the [L,3] strokes."""
    from .checkpoint import find_checkpoint, load_model
    from .vis import show_strokes

    if experiment_path:
        from pathlib import Path
        if not config_path:
            config_path = str(Path(experiment_path) / "config.yml")
        if not checkpoint_path:
            ckpt = find_checkpoint(experiment_path)
            checkpoint_path = str(ckpt) if ckpt else None
    if not config_path or not checkpoint_path:
        raise ValueError("Both config_path and checkpoint_path must be provided, either directly or via experiment_path.")
    style = load_style(source, style_weights)
    model = load_model(config_path, checkpoint_path, precision=precision, max_B=1, style_rows=style.shape[1])
    strokes = infer(prompt, style, model, diffusion_mode=diffusion_mode, seed=seed)
    if render:
        show_strokes(strokes, scale=1, name=output, show_output=False)
    return strokes
