"""Device scene generator (csrc/scenes.hip) vs its numpy oracle; batches are a pure function of (seed, sample index)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,I,K,seed,first", [(6, 48, 5, 11, 0), (4, 128, 11, 1234, 1000)])
def test_scenes_match_oracle(B, I, K, seed, first):
    from oracle import scenes_oracle as so
    from spair_pytorch_amd.data import DeviceScatteredDigits
    ds = DeviceScatteredDigits(10 ** 6, B, I, K, seed=seed)
    img, bbox, cnt = ds.batch(first // B)
    ri, rb, rc = so.generate(seed, first, B, I, K)
    assert np.array_equal(cnt.cpu().numpy(), rc)                  # integer outputs: bit-exact
    assert np.array_equal(bbox.cpu().numpy(), rb)
    assert np.abs(img.cpu().numpy() - ri).max() <= 2e-5           # pixels: fp32 rounding (fma contraction on the device)
    assert img.min().item() >= 0.0 and img.max().item() <= 1.0


def test_scenes_stream_is_indexable_and_sharded():
    from spair_pytorch_amd.data import DeviceScatteredDigits
    a = DeviceScatteredDigits(4096, 8, 64, 7, seed=5)
    whole = torch.cat([a.batch(i)[0] for i in range(4)])
    r0 = DeviceScatteredDigits(4096, 8, 64, 7, seed=5, rank=0, world=2)
    r1 = DeviceScatteredDigits(4096, 8, 64, 7, seed=5, rank=1, world=2)
    assert torch.equal(r0.batch(0)[0], whole[0:8]) and torch.equal(r1.batch(0)[0], whole[8:16])
    assert torch.equal(r0.batch(1)[0], whole[16:24]) and torch.equal(r1.batch(1)[0], whole[24:32])
    x, bbox, cnt = a.batch(3)
    assert x.shape == (8, 1, 64, 64) and bbox.shape == (8, 7, 4) and cnt.dtype == torch.int64
    assert (bbox[cnt.unsqueeze(1) <= torch.arange(7, device="cuda").unsqueeze(0)] == 0).all()     # zero padded past the count


def test_scenes_feed_a_training_step():
    from spair_pytorch_amd import config as cfg, models
    from spair_pytorch_amd.data import DeviceScatteredDigits
    from spair_pytorch_amd.optim import FusedAdam
    cfg.set_grid(48, (2, 2, 2, 1, 1, 1))
    torch.manual_seed(3)
    m = models.SPAIR([1, 48, 48], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
    opt = FusedAdam(m, lr=1e-4)
    ds = DeviceScatteredDigits(64, 8, 48, 3, seed=2, obj_px=(10, 20))
    losses = []
    for x, bbox, cnt in ds:
        opt.zero_grad()
        loss = m(x, 2000)[0]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert len(losses) == 8 and all(np.isfinite(losses))
