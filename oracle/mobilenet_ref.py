"""CPU restatement of the reference's StyleExtractor (text_style.py:11-59) — TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py.

PARITY UNPINNED.  The arithmetic of this component lives in third-party torchvision 0.17.2 (reference uv.lock:1389,
`models.mobilenet_v2(weights=MobileNet_V2_Weights.DEFAULT).features`, text_style.py:19-22,54), which is not importable in
the build container, and its pretrained weights cannot be fetched; the reference's own tests never run it (they feed
random style vectors, tests/test_model.py:19).  So this file restates torchvision's published MobileNetV2 architecture
(Sandler et al. 2018; torchvision/models/mobilenetv2.py: inverted_residual_setting [[1,16,1,1],[6,24,2,2],[6,32,3,2],
[6,64,4,2],[6,96,3,1],[6,160,3,2],[6,320,1,1]], stem Conv 3x3/2 -> 32, last Conv 1x1 -> 1280, Conv-BN(eps 1e-5)-ReLU6,
residual iff stride 1 and equal widths) with torchvision's state_dict key names, and the checks against it are
self-consistency checks of the HIP path, not parity with the reference.

`forward(sd, img)` follows text_style.py:43-59 line by line around that network.
"""
import numpy as np
import torch
import torch.nn.functional as F

SETTINGS = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
BN_EPS = 1e-5


def blocks():
    """[(features index, t, cin, hidden, cout, stride, residual)]"""
    out, cin, idx = [], 32, 1
    for t, c, n, s in SETTINGS:
        for i in range(n):
            stride = s if i == 0 else 1
            out.append((idx, t, cin, cin * t, c, stride, stride == 1 and cin == c))
            cin, idx = c, idx + 1
    return out


def key_shapes():
    """torchvision MobileNetV2 `features.*` state_dict: [(key, shape)] in module order (without num_batches_tracked)."""
    ks = []

    def bn(n, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            ks.append((f"{n}.{s}", (c,)))

    ks.append(("features.0.0.weight", (32, 3, 3, 3)))
    bn("features.0.1", 32)
    for idx, t, cin, hid, cout, stride, res in blocks():
        p, j = f"features.{idx}.conv.", 0
        if t != 1:
            ks.append((p + "0.0.weight", (hid, cin, 1, 1)))
            bn(p + "0.1", hid)
            j = 1
        ks.append((f"{p}{j}.0.weight", (hid, 1, 3, 3)))
        bn(f"{p}{j}.1", hid)
        ks.append((f"{p}{j + 1}.weight", (cout, hid, 1, 1)))
        bn(f"{p}{j + 2}", cout)
    ks.append(("features.18.0.weight", (1280, 320, 1, 1)))
    bn("features.18.1", 1280)
    return ks


def synthetic_state_dict(seed: int = 0):
    """Random weights with realistic scales: kaiming-like convolutions, BN gamma ~ U(0.5, 1.5), beta ~ N(0, 0.1),
    running_mean ~ N(0, 0.1), running_var ~ U(0.5, 1.5) — every BN term matters, activations stay O(1) through 53 layers."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for k, shp in key_shapes():
        if k.endswith("running_var") or (k.endswith(".weight") and len(shp) == 1):
            v = rng.uniform(0.5, 1.5, shp)
        elif len(shp) == 1:
            v = rng.standard_normal(shp) * 0.1
        else:
            fan_in = int(np.prod(shp[1:]))
            v = rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)
        sd[k] = torch.from_numpy(v.astype(np.float32))
    return sd


def _cbr(x, sd, conv, bn, stride=1, groups=1, act=True):
    w = sd[conv + ".weight"]
    x = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2, groups=groups)
    x = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"], False, 0.0, BN_EPS)
    return F.relu6(x) if act else x


def features(sd, x):
    """mobilenet_v2.features in eval mode: [B,3,H,W] -> [B,1280,H/32,W/32]"""
    x = _cbr(x, sd, "features.0.0", "features.0.1", stride=2)
    for idx, t, cin, hid, cout, stride, res in blocks():
        p, j, h = f"features.{idx}.conv.", 0, x
        if t != 1:
            h = _cbr(h, sd, p + "0.0", p + "0.1")
            j = 1
        h = _cbr(h, sd, f"{p}{j}.0", f"{p}{j}.1", stride=stride, groups=hid)
        h = _cbr(h, sd, f"{p}{j + 1}", f"{p}{j + 2}", act=False)
        x = x + h if res else h
    return _cbr(x, sd, "features.18.0", "features.18.1")


def forward(sd, img_batch, return_features=False):
    """StyleExtractor.forward (text_style.py:43-59): img_batch [B,1,H,W] grey levels -> [B,14,1280]."""
    with torch.no_grad():
        x = torch.as_tensor(np.asarray(img_batch), dtype=torch.float32)
        x = (x / 127.5) - 1
        x = x.repeat(1, 3, 1, 1)
        f = features(sd, x)
        x = F.avg_pool2d(f, kernel_size=3, stride=3)
        x = F.adaptive_avg_pool2d(x, (1, 14))
        x = x.squeeze(2).permute(0, 2, 1)
    return (x, f) if return_features else x
