"""Checkpoint / resume: a resumed run continues where it stopped (same loss to the bit, same update to rounding), reference-format state_dict files load,
and the fused optimizer's state is interchangeable with torch.optim.Adam."""
import os

import pytest
import torch

from helpers import assert_adam_updates_close

pytestmark = pytest.mark.gpu


def _model(seed=3):
    from spair_pytorch_amd import config as cfg, models
    cfg.set_grid(48, (2, 2, 2, 1, 1, 1))
    torch.manual_seed(seed)
    return models.SPAIR([1, 48, 48], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")


def _batch():
    from spair_pytorch_amd.data import scattered_digits
    return torch.from_numpy(scattered_digits(7, 4, 48, 3, obj_px=(10, 20))[0]).cuda()


def _step(m, opt, x, it, seed):
    torch.manual_seed(seed)                 # the per-step noise
    opt.zero_grad()
    loss = m(x, it)[0]
    loss.backward()
    opt.step()
    return loss.item()


def test_resume_continues_the_run(tmp_path):
    from spair_pytorch_amd import checkpoint as ck
    from spair_pytorch_amd.optim import FusedAdam
    x = _batch()
    m1 = _model()
    o1 = FusedAdam(m1, lr=1e-3)
    for it in range(2):
        _step(m1, o1, x, 2000 + it, 100 + it)
    path = os.path.join(tmp_path, "ck.pt")
    ck.save_checkpoint(path, m1, o1, iteration=2002)
    l1 = _step(m1, o1, x, 2002, 102)
    m2 = _model(seed=99)                      # different init: everything must come from the file
    o2 = FusedAdam(m2, lr=1e-3)
    it = ck.load_checkpoint(path, m2, o2)
    assert it == 2002 and o2.step_count == 2
    l2 = _step(m2, o2, x, it, 102)
    assert l1 == l2                                                        # forward: bit-exact
    # backward: bias / edge gradients are summed with fp32 atomics, so two runs agree to rounding, not to the bit
    assert_adam_updates_close(m1.flat_parameters().cpu().numpy(), m2.flat_parameters().cpu().numpy(), 1e-3, tight=1e-6)
    scale = o1.exp_avg.abs().max().item()
    assert (o1.exp_avg - o2.exp_avg).abs().max().item() <= 1e-5 * scale
    assert (o1.exp_avg_sq - o2.exp_avg_sq).abs().max().item() <= 1e-5 * o1.exp_avg_sq.abs().max().item()


def test_reference_format_state_dict_file_loads(tmp_path):
    from spair_pytorch_amd import checkpoint as ck
    m1, m2 = _model(3), _model(4)
    path = os.path.join(tmp_path, "spair_reference_style.pt")
    torch.save({k: v.cpu() for k, v in m1.state_dict().items()}, path)     # train.py:85-90
    assert ck.load_checkpoint(path, m2) == 0
    assert torch.equal(m1.flat_parameters(), m2.flat_parameters())


def test_optimizer_state_exchanges_with_torch_adam():
    from spair_pytorch_amd import checkpoint as ck
    from spair_pytorch_amd.optim import FusedAdam
    x = _batch()
    m = _model()
    opt = FusedAdam(m, lr=1e-3)
    for it in range(2):
        _step(m, opt, x, 2000 + it, 5 + it)
    # continue one step with torch.optim.Adam fed from the fused state, and with the fused optimizer, on the same gradients
    tsd = ck.adam_state_to_torch(m, opt)
    torch.manual_seed(9)
    opt.zero_grad()
    m(x, 2002)[0].backward()
    grads = m.flat_gradients().clone()
    before = m.flat_parameters().clone()
    opt.step()
    fused_after = m.flat_parameters().clone()
    m.flat_parameters().copy_(before)
    m.flat_gradients().copy_(grads)
    m._bind_grads()
    tadam = torch.optim.Adam(m.parameters(), lr=1e-3)
    tadam.load_state_dict(tsd)
    tadam.step()
    assert (m.flat_parameters() - fused_after).abs().max().item() <= 2e-7
    # and back: the torch state after that step loads into a fresh fused optimizer
    opt2 = FusedAdam(m, lr=1e-3)
    ck.adam_state_from_torch(m, opt2, tadam.state_dict())
    assert opt2.step_count == 3
    assert (opt2.exp_avg - opt.exp_avg).abs().max().item() <= 1e-6 * max(1.0, opt.exp_avg.abs().max().item())
