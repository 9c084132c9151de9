"""StyleExtractor front end (reference text_style.py:11-59) through the C-ABI (include/dhw_style.h).

PARITY UNPINNED: torchvision (whose MobileNetV2 the reference calls) is not importable here and its pretrained weights
cannot be fetched, and no reference test runs this component.  These tests check the HIP path against the build's own
PyTorch restatement of the published architecture (oracle/mobilenet_ref.py) on random-init weights — self-consistency of
the kernels (BN folding, channel padding, stride / padding arithmetic, the two pools), not parity with the reference."""
import ctypes as C

import numpy as np
import pytest
import torch

import dhg_amd
from dhg_amd import _lib
from oracle import mobilenet_ref

pytestmark = pytest.mark.gpu


def _imgs(B, H, W, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(B, 1, H, W)).astype(np.float32)


@pytest.fixture(scope="module")
def sd():
    return mobilenet_ref.synthetic_state_dict(0)


def test_state_dict_inventory_is_torchvisions_mobilenet_v2_features(sd):
    ex = dhg_amd.StyleExtractor(sd, precision="fp32")
    assert ex.state_dict_keys() == [(k, tuple(s)) for k, s in mobilenet_ref.key_shapes()]
    assert len(ex.state_dict_keys()) == 260 and sum(int(np.prod(s)) for _, s in ex.state_dict_keys()) == 2257984
    # strict load, as load_state_dict: the classifier head and BN counters of the full checkpoint are accepted and ignored
    full = dict(sd)
    full["classifier.1.weight"] = torch.zeros(1000, 1280)
    full["features.0.1.num_batches_tracked"] = torch.tensor(0)
    ex.load_state_dict(full)
    with pytest.raises(_lib.DhwError, match="unexpected key"):
        ex.load_state_dict({**sd, "features.19.0.weight": torch.zeros(1)})
    with pytest.raises(_lib.DhwError, match="size mismatch"):
        ex.load_state_dict({**sd, "features.0.0.weight": torch.zeros(32, 1, 3, 3)})
    h = C.c_void_p()
    l = _lib.lib()
    assert l.dhw_style_create(C.byref(h), _lib.PREC_F32, 0) == 0
    try:
        x = torch.zeros(1, 1, 96, 96, device="cuda")
        o = torch.zeros(1, 14, 1280, device="cuda")
        assert l.dhw_style_forward(h, x.data_ptr(), 1, 96, 96, o.data_ptr(), None) == -2
        assert b"missing key" in l.dhw_style_last_error(h)
    finally:
        l.dhw_style_destroy(h)


@pytest.mark.parametrize("B,H,W", [(2, 96, 200), (1, 96, 96), (3, 96, 459), (1, 128, 1405)])
def test_fp32_matches_the_pytorch_restatement(sd, B, H, W):
    """odd widths exercise ceil(W/2) at all five stride-2 stages; W=96 gives a 3x3 map (one pooled column spread over 14
    bins); H=128 a 4-row map (floor pooling drops a row); W=1405 is a real line width (44 columns -> 14 bins of 1-2)."""
    ex = dhg_amd.StyleExtractor(sd, precision="fp32")
    img = _imgs(B, H, W, seed=W)
    ref, feat = mobilenet_ref.forward(sd, img, return_features=True)
    out = ex(img).cpu()
    f = ex.debug_features().permute(0, 3, 1, 2)
    assert out.shape == (B, 14, 1280) and f.shape == feat.shape
    assert (f - feat).abs().max().item() < 2e-4 * max(1.0, feat.abs().max().item())
    assert (out - ref).abs().max().item() < 5e-4     # fp32 summation order over 53 layers; pooled features reach ~6 (measured 1.7e-4)


def test_bf16_tracks_fp32_and_inputs_are_flexible(sd):
    img = _imgs(2, 96, 333, seed=5)
    ref = mobilenet_ref.forward(sd, img)
    out = dhg_amd.StyleExtractor(sd, precision="bf16")(torch.from_numpy(img)).cpu()     # tensor input, bf16 activations
    # 53 layers of bf16 activations / pointwise weights (fp32 accumulation) through a random-init ReLU6 network, whose
    # pre-activations sit at random offsets from the clipping points: rounding differences grow layer by layer.
    # Relative L2 error of the pooled features (measured r2: see the printed value); the fp32 mode is the parity mode.
    rel = ((out - ref).norm() / ref.norm()).item()
    print(f"StyleExtractor bf16 vs fp32 restatement: relative L2 error {rel:.4f}, max abs {(out - ref).abs().max().item():.3f}")
    assert rel < 0.15
    with pytest.raises(ValueError):
        dhg_amd.StyleExtractor(sd)(np.zeros((2, 96, 96), np.float32))
    with pytest.raises(_lib.DhwError, match="at least 96"):
        dhg_amd.StyleExtractor(sd)(np.zeros((1, 1, 64, 200), np.float32))


def test_features_feed_the_sampler():
    """image -> StyleExtractor -> style_vector -> sample: the full front-to-back path the reference's `infer` runs."""
    from dhg_amd import spec
    with pytest.warns(UserWarning, match="random"):
        ex = dhg_amd.StyleExtractor()
    sv = ex(_imgs(2, 96, 640, seed=1))
    assert sv.shape == (2, 14, 1280) and torch.isfinite(sv).all()
    m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=2, max_L=64, max_Lt=4).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()})
    out = dhg_amd.sample(m, torch.tensor([[3, 4, 5, 0], [9, 8, 7, 6]]).cuda(), sv, L=64, T=4, seed=1)
    assert out.shape == (2, 64, 3) and torch.isfinite(out).all()


def test_infer_file_takes_a_handwriting_image_like_the_reference(tmp_path, monkeypatch):
    """reference inference.py:19-96 end to end with `source` = an image file: read_img -> StyleExtractor -> sampler."""
    from PIL import Image
    from dhg_amd import spec
    rng = np.random.Generator(np.random.PCG64(3))
    img = np.full((140, 900), 255, np.uint8)
    for _ in range(400):                                   # dark strokes inside a white margin
        y, x = rng.integers(20, 110), rng.integers(30, 850)
        img[y:y + rng.integers(2, 9), x:x + rng.integers(2, 20)] = rng.integers(0, 90)
    Image.fromarray(img).save(tmp_path / "writer.png")
    (tmp_path / "config.yml").write_text("training_args:\n  att_layers_num: 2\n  channels: 128\n  dropout: 0.0\n")
    torch.save({"state_dict": {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}}, tmp_path / "model_final.pth")
    torch.save(mobilenet_ref.synthetic_state_dict(0), tmp_path / "mobilenet_v2.pth")
    monkeypatch.chdir(tmp_path)
    strokes = dhg_amd.infer_file("Rabbit", str(tmp_path / "writer.png"), experiment_path=str(tmp_path), output="res", seed=2,
                                 render=False, style_weights=str(tmp_path / "mobilenet_v2.pth"))
    assert strokes.shape == (7 * 16 - (7 * 16) % 8 + 8, 3) and np.isfinite(strokes).all()
    # the same features, computed step by step
    x = dhg_amd.read_img(tmp_path / "writer.png", 96)
    assert x.shape[0] == 96 and x.dtype == np.uint8
    sv = dhg_amd.load_style(str(tmp_path / "writer.png"), str(tmp_path / "mobilenet_v2.pth"))
    ref = mobilenet_ref.forward(mobilenet_ref.synthetic_state_dict(0), x[None, None].astype(np.float32))
    assert (sv.cpu() - ref).abs().max().item() < 5e-4
