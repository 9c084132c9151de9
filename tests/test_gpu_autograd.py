"""`dhg_amd.DiffusionModel.forward` under autograd: the reference's ONE class serves inference.py:89 and train.py:46-60; here the
same object routes a grad-recording call through the training kernels (train_model.TrainModel) as one autograd node.  Checked
against the CPU oracle under torch autograd (the oracle is pinned to the reference's gradients by tests/golden/model_grad.npz in
test_gpu_train.py)."""
import numpy as np
import pytest
import torch

import dhg_amd
from dhg_amd import spec
from oracle import ref_cpu

pytestmark = pytest.mark.gpu


def _setup(B=2, L=64, Lt=8, pad=2, seed=5):
    sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}
    m = dhg_amd.DiffusionModel(2, precision="fp32", max_B=B, max_L=L, max_Lt=Lt)
    m.load_state_dict(sd, strict=True)
    inp = spec.synthetic_inputs(B, L, Lt, seed=seed, pad=pad)
    x, text, style = (torch.from_numpy(inp[k]) for k in ("strokes", "text", "style"))
    sigma = torch.linspace(0.3, 0.9, B).reshape(B, 1)
    rng = np.random.Generator(np.random.PCG64(seed))
    w1 = torch.from_numpy(rng.standard_normal((B, L, 2)).astype(np.float32))
    w2 = torch.from_numpy(rng.standard_normal((B, L)).astype(np.float32))
    return sd, m, x, text, sigma, style, w1, w2


def test_forward_is_differentiable_and_matches_oracle_autograd():
    sd, m, x, text, sigma, style, w1, w2 = _setup()
    m.eval()                       # no dropout: deterministic
    assert not any(p.requires_grad for p in m.parameters())
    with torch.no_grad():
        e_inf, p_inf, _ = m(x.cuda(), text.cuda(), sigma.cuda(), style.cuda())       # inference kernels (fp32 mode)
    m.requires_grad_(True)
    eps, pen, none = m(x.cuda(), text.cuda(), sigma.cuda(), style.cuda())
    assert none is None and eps.requires_grad and pen.requires_grad
    assert (eps - e_inf).abs().max() < 5e-5 and (pen - p_inf).abs().max() < 5e-5     # two independent HIP implementations agree
    loss = (eps * w1.cuda()).sum() + (pen * w2.cuda()).sum()
    loss.backward()
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    e_o, p_o = ref_cpu.forward(sdo, x, text, sigma, style)
    ((e_o * w1).sum() + (p_o * w2).sum()).backward()
    assert (eps.detach().cpu() - e_o.detach()).abs().max() < 2e-5
    worst = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == sdo[k].grad.shape, k
        go = sdo[k].grad
        err = float((p.grad.cpu() - go).abs().max())
        if float(go.norm()) > 1e-3:
            worst = max(worst, err / float(go.norm()))
        assert err <= 2e-4 * float(go.norm()) + 1e-6, (k, err, float(go.norm()))
    print("worst |g - g_oracle| / ||g_oracle|| =", worst)
    # an optimizer step on the SAME parameters is what the inference path then samples from
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    before = m.state_dict()["enc1.fc.weight"].detach().clone()
    opt.step()
    after = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    assert not torch.equal(after["enc1.fc.weight"], before.cpu())
    with torch.no_grad():
        e2, p2, _ = m(x.cuda(), text.cuda(), sigma.cuda(), style.cuda())
        e2o, p2o = ref_cpu.forward(after, x, text, sigma, style)
    assert (e2.cpu() - e2o).abs().max() < 5e-5 and (p2.cpu() - p2o).abs().max() < 5e-5
    # gradients accumulate like autograd's: a second backward adds to .grad
    g0 = m.get_parameter("enc1.fc.weight").grad.clone()
    eps, pen, _ = m(x.cuda(), text.cuda(), sigma.cuda(), style.cuda())
    stale = eps.sum()
    eps_b, pen_b, _ = m(x.cuda(), text.cuda(), sigma.cuda(), style.cuda())
    with pytest.raises(RuntimeError, match="latest"):
        stale.backward()           # the tape belongs to the latest forward
    (eps_b.sum() + pen_b.sum()).backward()
    assert not torch.equal(m.get_parameter("enc1.fc.weight").grad, g0)


def test_train_mode_runs_with_dropout_and_survives_a_move():
    sd, m, x, text, sigma, style, w1, w2 = _setup(seed=6)
    m.train()                      # the reference's state after construction: parameters require grad, dropout on
    assert all(p.requires_grad for p in m.parameters())
    eps, pen, _ = m(x, text, sigma, style)                  # host inputs, as train.py hands them over after .to(device)
    (eps.square().mean() + pen.mean()).backward()
    g = [p.grad for p in m.parameters()]
    assert all(t is not None and torch.isfinite(t).all() for t in g)
    assert sum(float(t.abs().sum()) for t in g) > 0
    m.cpu()                        # re-allocates the parameters: the next call re-links them to the trainer's buffer
    m.zero_grad()
    eps, pen, _ = m(x.cuda(), text.cuda(), sigma.cuda(), style.cuda())
    (eps.square().mean() + pen.mean()).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_train_mode_draws_fresh_dropout_masks_per_call_and_eval_mode_is_deterministic():
    """reference model.py:23 / text_style.py:88: nn.Dropout draws a new mask in every train-mode call; eval mode is a function
    of the inputs.  The EncoderLayer masks come from the device generator keyed by a per-call draw index, which the
    differentiable forward has to advance itself (it is not a train_step)."""
    sd, m, x, text, sigma, style, w1, w2 = _setup(seed=8)
    m.train()
    assert m.drop_rate > 0
    # the style Dropout(0.3) mask is drawn on the host per call as well: pin torch's generator so that ONLY the EncoderLayer
    # dropout (device generator) can differ between the two calls
    torch.manual_seed(123)
    a, pa, _ = m(x, text, sigma, style)
    idx_a = int(m._trainer.rng[1])
    torch.manual_seed(123)
    b, pb, _ = m(x, text, sigma, style)
    idx_b = int(m._trainer.rng[1])
    assert idx_b == idx_a + 1
    assert not torch.equal(a.detach(), b.detach()), "two train-mode forwards replayed the same EncoderLayer dropout masks"
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    m.eval()
    m.requires_grad_(True)             # still the differentiable path (fp32 training kernels), dropout off
    c, pc, _ = m(x, text, sigma, style)
    d, pd, _ = m(x, text, sigma, style)
    # (the training kernels' split-K GEMMs accumulate with fp32 atomics: two runs agree to rounding level, not bit for bit)
    assert torch.allclose(c.detach(), d.detach(), rtol=1e-5, atol=1e-6) and torch.allclose(pc.detach(), pd.detach(), rtol=1e-5, atol=1e-6)
    assert (a.detach() - b.detach()).abs().max() > 100 * (c.detach() - d.detach()).abs().max()     # dropout changes values, not just their last bit
