"""Front end around the sampler: checkpoint + config resolution and the stroke -> polyline output
side.  CPU only: nothing here needs the HIP library beyond constructing the (lazy) model object."""
import os

import numpy as np
import pytest
import torch

import dhg_amd
from dhg_amd import spec


def _sd(nl=2):
    return {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(nl).items()}


@pytest.mark.parametrize("wrap", ["bare", "state_dict", "module_prefix"])
def test_checkpoint_formats_of_the_reference_are_read(tmp_path, wrap):
    sd = _sd()
    obj = {"bare": sd, "state_dict": {"state_dict": sd, "meta": {"iter": 7}},
           "module_prefix": {"state_dict": {"module." + k: v for k, v in sd.items()}}}[wrap]
    f = tmp_path / "model_final.pth"
    torch.save(obj, f)
    got = dhg_amd.read_state_dict(f)
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_checkpoint_without_a_dict_is_rejected(tmp_path):
    f = tmp_path / "x.pth"
    torch.save(torch.zeros(3), f)
    with pytest.raises(RuntimeError, match="No state_dict"):
        dhg_amd.read_state_dict(f)


def test_checkpoint_discovery_order(tmp_path):
    assert dhg_amd.find_checkpoint(tmp_path) is None
    for name in ("checkpoint_900.pth", "checkpoint_10000.pth", "checkpoint_last.pth"):
        (tmp_path / name).write_bytes(b"")
    assert dhg_amd.find_checkpoint(tmp_path).name == "checkpoint_10000.pth"      # numeric, not lexicographic
    (tmp_path / "model_last.pth").write_bytes(b"")
    assert dhg_amd.find_checkpoint(tmp_path).name == "model_last.pth"
    (tmp_path / "model_final.pth").write_bytes(b"")
    assert dhg_amd.find_checkpoint(tmp_path).name == "model_final.pth"


def test_config_maps_to_model_dims_and_strict_load(tmp_path):
    cfg = tmp_path / "config.yml"
    cfg.write_text("training_args:\n  att_layers_num: 2\n  channels: 128\n  dropout: 0.0\n")
    assert dhg_amd.read_config(cfg) == {"num_layers": 2, "c1": 128, "c2": 192, "c3": 256, "drop_rate": 0.0}
    # an incomplete config fails here, as `cfg.training_args.<key>` does in the reference (checkpoint.py:280-286): no defaults
    bad = tmp_path / "bad.yml"
    for text in ("training_args:\n  att_layers_num: 2\n  channels: 128\n", "other: 1\n", ""):
        bad.write_text(text)
        with pytest.raises(KeyError):
            dhg_amd.read_config(bad)
    ck = tmp_path / "model_final.pth"
    torch.save({"state_dict": _sd(2)}, ck)
    m = dhg_amd.load_model(cfg, ck)
    assert m.num_layers == 2 and not m.training
    assert torch.equal(m.state_dict()["enc1.conv1.weight"], _sd(2)["enc1.conv1.weight"])
    m2 = dhg_amd.load_model(None, ck)                      # layer count read off the checkpoint's keys
    assert m2.num_layers == 2
    torch.save({"state_dict": _sd(4)}, ck)                 # config says 2 attention layers, checkpoint has 4
    with pytest.raises(RuntimeError, match="att_layers"):
        dhg_amd.load_model(cfg, ck)


def test_infer_file_argument_errors_mirror_the_reference(tmp_path):
    with pytest.raises(ValueError, match="config_path and checkpoint_path"):
        dhg_amd.infer_file("abc", np.zeros((14, 1280), np.float32))
    with pytest.raises(ValueError, match="config_path and checkpoint_path"):
        dhg_amd.infer_file("abc", np.zeros((14, 1280), np.float32), experiment_path=str(tmp_path))   # empty directory
    with pytest.raises(ValueError, match="handwriting image"):
        dhg_amd.load_style("writer.xyz")
    with pytest.raises(ValueError, match="style features"):
        dhg_amd.load_style(np.zeros((14, 100), np.float32))
    f = tmp_path / "style.npy"
    np.save(f, np.ones((14, 1280), np.float32))
    assert tuple(dhg_amd.load_style(str(f)).shape) == (1, 14, 1280)


def test_strokes_to_polylines_follows_the_reference_plot_loop():
    s = np.array([[1, 0, 0], [1, 0, 0], [0, 1, 0.9], [1, 0, 0.2], [1, 1, 0], [0, 0, 1], [2, 0, 0]], np.float32)
    lines = dhg_amd.strokes_to_polylines(s)
    # pen-up at rows 2 and 5: the stretch before row 2, then rows 2..4; the tail after the last lift is not drawn (vis.py:19-31)
    assert [l.tolist() for l in lines] == [[[1, 0], [2, 0]], [[2, 1], [3, 1], [4, 2]]]
    assert dhg_amd.strokes_to_polylines(np.zeros((4, 3), np.float32)) == []


def test_polylines_match_the_reference_show_strokes(golden_dir):
    """tests/golden/vis.npz: the (x, y) arrays of every `plt.plot` call the reference's show_strokes (utils/vis.py:5-36)
    makes for a seeded stroke array, recorded by oracle/make_golden_r2.py — incl. a pen-up on row 0 (empty first
    polyline) and a pen value of exactly 0.5 (numpy rounds half to even: no split)."""
    g = np.load(os.path.join(golden_dir, "vis.npz"))
    lines = dhg_amd.strokes_to_polylines(g["strokes"])
    assert len(lines) == int(g["n_lines"]) and [len(l) for l in lines] == g["lens"].tolist()
    xs = np.concatenate([l[:, 0] for l in lines])
    ys = np.concatenate([l[:, 1] for l in lines])
    assert np.array_equal(xs, g["xs"]) and np.array_equal(ys, g["ys"])
    pos = np.cumsum(g["strokes"][:, :2], axis=0).T
    w, h = np.max(pos, axis=-1) - np.min(pos, axis=-1)
    assert np.allclose([w / h, 1.0], g["figsize"], rtol=1e-6)       # the figure size show_strokes asks for (scale = 1)


def test_read_img_crops_the_margins_and_resizes_to_96_rows(tmp_path):
    """reference utils/io.py:98-115 + utils/preprocessing.py:47-62: grey load, crop to the inked rows / columns (exclusive
    upper bounds, as the reference slices), height 96 with width = 96 * w // h.  (cv2 is absent here: the cubic kernel
    follows OpenCV's INTER_CUBIC conventions but is not pinned against it.)"""
    from PIL import Image
    img = np.full((60, 200), 255, np.uint8)
    img[10:40, 20:150] = 255
    img[10:40:3, 20:150:2] = 0            # ink inside rows 10..37, columns 20..148
    Image.fromarray(img).save(tmp_path / "w.png")
    cropped = dhg_amd.remove_whitespace(img, 127)
    rows = np.nonzero(img.min(1) < 127)[0]
    cols = np.nonzero(img.min(0) < 127)[0]
    assert cropped.shape == (rows[-1] - rows[0], cols[-1] - cols[0])          # last inked row / column dropped
    out = dhg_amd.read_img(tmp_path / "w.png", 96)
    h, w = cropped.shape
    assert out.dtype == np.uint8 and out.shape == (96, 96 * w // h)
    # cubic interpolation of a constant image is that constant; of a step it stays within the overshoot bounds
    from dhg_amd.inference import _resize_cubic
    assert np.array_equal(_resize_cubic(np.full((7, 9), 93, np.uint8), 31, 20), np.full((20, 31), 93, np.uint8))
    up = _resize_cubic(np.repeat(np.array([[0, 255]], np.uint8), 4, 0), 8, 4)
    assert up[0, 0] == 0 and up[0, -1] == 255 and (np.diff(up[0].astype(int)) >= 0).all()
