"""The sequential count-prior KL (models.py:186-257; csrc/loss.hip k_count_kl, relative-bin form) cell by cell against the oracle, on presence
patterns the fixtures do not reach: every cell present, no cell present, random halves -- at the sharp end of the prior schedule, where the
normaliser's 1e-6 clamp can engage -- on 16 x 16, the reference's 11 x 11 and 32 x 32 grids (5, 2 and 17 bin registers per lane)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import spair_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("I,strides,B", [(128, (2, 2, 2, 1, 1, 1), 6), (128, (3, 2, 2, 1, 1, 1), 5), (256, (2, 2, 2, 1, 1, 1), 4)])
@pytest.mark.parametrize("step", [1, 6000, 12000])
def test_count_kl_cell_by_cell(I, strides, B, step):
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.data import scattered_digits
    cfg.set_grid(I, strides)
    G = gi.grid_side(I, strides)
    torch.manual_seed(2)
    m = SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype="f32").to("cuda")
    x = torch.from_numpy(scattered_digits(5, B, I, 9)[0]).cuda()
    noise = {k: torch.from_numpy(v) for k, v in gi.make_noise(4, B, G).items()}
    # presence patterns through the logistic noise of the relaxed Bernoulli: u -> 1 switches a cell on, u -> 0 off
    rng = np.random.default_rng(8)
    u = noise["u_pres"].numpy().copy()
    u[0] = 1.0 - 1e-7                                   # every cell present
    u[1] = 1e-7                                         # none
    for b in range(2, B):
        u[b] = np.where(rng.uniform(size=u[b].shape) < (0.5 if b % 2 else 0.05), 1.0 - 1e-7, 1e-7)
    noise["u_pres"] = torch.from_numpy(u.astype(np.float32))
    with torch.no_grad():
        _, _, _, z_pres = m(x, step, noise={k: v.cuda() for k, v in noise.items()})
    terms = m.loss_terms().cpu()
    pz = m.export_map(14).cpu().double()                # the kernel's p(z_pres = 1 | counts so far) per cell
    z = z_pres.cpu().double()
    e = 1e-9
    kl_cells = z * (torch.log(z + e) - torch.log(pz + e)) + (1 - z) * (torch.log(1 - z + e) - torch.log(1 - pz + e))
    ref = orc.compute_kl({}, z_pres.cpu(), step, orc.OracleConfig(image_shape=(1, I, I), conv_strides=strides))["pres_dist"].double()
    on = (z > 0.5).flatten(1).sum(1)
    assert on[0].item() == G * G and on[1].item() == 0
    err = (kl_cells - ref).abs()
    assert err.max().item() <= 1e-4 * ref.abs().max().item() + 1e-5, (err.max().item(), ref.abs().max().item())
    want = ref.sum().item() / B
    assert abs(terms[8].item() - want) <= 1e-4 * abs(want) + 1e-4, (terms[8].item(), want)
