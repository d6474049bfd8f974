"""A failed step is loud (SpairStep.status / status_host, spair_adam_guarded; include/spair_hip.h).  The reference raises on any NaN in its
forward (spair/debug_tools.py:245-271, called at models.py:65,108,245); the HIP step has no host synchronisation to raise at, so
  * the loss kernel flags a non-finite loss term in two device ints the model owns (sticky bits, this step's bits) and in one host word,
  * ``FusedAdam`` leaves a flagged step out whole and a non-finite gradient element on its own (no ``lr * NaN`` in a parameter),
  * the next ``forward()`` raises from the host word (a plain load), ``step_status()`` / ``check_step_status()`` report where the host waits."""
import numpy as np
import pytest
import torch

from helpers import load_case
import golden_inputs as gi

pytestmark = pytest.mark.gpu


def _setup(dtype):
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.optim import FusedAdam
    z, case = load_case("c1_b8_step1001")
    cfg.set_grid(case["I"], case["strides"])
    m = SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gi.make_weights(case["wseed"], case["wscale"]).items()})
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    return m, FusedAdam(m, lr=1e-3), x, noise


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_nonfinite_step_is_flagged_skipped_and_raised(dtype):
    from spair_pytorch_amd._lib import SpairHipError
    m, opt, x, noise = _setup(dtype)

    def step(gs):
        opt.zero_grad()
        loss = m(x, gs, noise=noise)[0]
        loss.backward()
        opt.step()
        return loss

    assert np.isfinite(step(1001).item()) and m.step_status() == 0 and opt.skipped() == (0, False)
    good = m.flat_parameters().clone()
    mom = opt.exp_avg.clone()
    # poison one bias of the attribute encoder: its Gaussian KL term is NaN from here on
    bias = dict(m.named_parameters())["object_encoder.out.bias"]
    keep = bias.detach().clone()
    with torch.no_grad():
        bias[3] = float("nan")
    poisoned = m.flat_parameters().clone()
    loss = step(1002)
    assert not np.isfinite(loss.item())
    assert m.step_status() & 2
    # the optimizer left the step out whole: parameters (but for the poison itself) and moments are what they were
    assert opt.skipped()[0] == 1
    assert torch.equal(torch.nan_to_num(m.flat_parameters(), nan=7.0), torch.nan_to_num(poisoned, nan=7.0))
    assert torch.equal(opt.exp_avg, mom)
    # the next forward raises by itself (host word, no synchronisation needed by then) ...
    with pytest.raises(SpairHipError, match="non-finite"):
        m(x, 1003, noise=noise)
    with pytest.raises(SpairHipError, match="non-finite"):
        m.check_step_status()
    # ... until the caller has dealt with it
    with torch.no_grad():
        bias.copy_(keep)
    m.clear_step_status()
    assert torch.equal(m.flat_parameters(), good)
    assert np.isfinite(step(1003).item()) and m.step_status() == 0
    assert opt.skipped()[0] == 1 and not torch.equal(m.flat_parameters(), good)


def test_guarded_adam_leaves_out_nonfinite_gradient_elements():
    m, opt, x, noise = _setup("bf16")
    opt.zero_grad()
    loss = m(x, 1001, noise=noise)[0]
    loss.backward()
    g = m.flat_gradients()
    idx = torch.tensor([5, 1000, g.numel() - 1], device=g.device)
    nz = (g != 0).nonzero().flatten()
    before = m.flat_parameters().clone()
    g[idx] = torch.tensor([float("inf"), float("nan"), float("-inf")], device=g.device)
    opt.step()
    after = m.flat_parameters()
    assert opt.skipped() == (0, True)
    assert torch.isfinite(after).all() and torch.equal(after[idx], before[idx])
    moved = nz[~torch.isin(nz, idx)]
    assert (after[moved] != before[moved]).float().mean().item() > 0.99      # everything else took its Adam step


def test_stock_adam_sees_nan_loss_not_silent():
    """train.py:64-67 verbatim with torch.optim.Adam: nothing guards the parameters there, but the failure is not silent either -- the loss the
    loop logs is NaN, the status word says why, and the following forward raises."""
    from spair_pytorch_amd._lib import SpairHipError
    m, _, x, noise = _setup("bf16")
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    with torch.no_grad():
        dict(m.named_parameters())["object_encoder.out.bias"][0] = float("inf")
    opt.zero_grad()
    loss = m(x, 1001, noise=noise)[0]
    loss.backward()
    assert not np.isfinite(loss.item()) and m.step_status() == 2
    with pytest.raises(SpairHipError):
        m(x, 1002, noise=noise)
    # an in-place op on the returned loss leaves the logged terms alone (it is a view of the kernel's second copy of the total)
    m.clear_step_status()
    with torch.no_grad():
        dict(m.named_parameters())["object_encoder.out.bias"][0] = 0.0
    loss = m(x, 1002, noise=noise)[0]
    t0 = m.loss_terms().clone()
    with torch.no_grad():
        loss /= 4
    assert torch.equal(m.loss_terms(), t0) and abs(loss.item() * 4 - t0[0].item()) <= 1e-6 * abs(t0[0].item())
