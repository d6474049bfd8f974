"""The reference's Python surface beyond SPAIR.forward: stn() in both directions with gradients (pinned to the reference's
own vectors in units.npz), Backbone.forward / SequentialMultipleOutput.forward on their own, the verbatim train.py:64-67 loop with the
stock torch.optim.Adam, and the engine-cache / stale-activation guards."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from helpers import assert_adam_updates_close, load_case

pytestmark = pytest.mark.gpu


def _model(case, dtype="f32"):
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    cfg.set_grid(case["I"], case["strides"])
    m = SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gi.make_weights(case["wseed"], case["wscale"]).items()})
    return m


def test_stn_inverse_matches_reference_vectors(golden_dir):
    """modules.py:256-269 through modules.stn(inverse=True): value and both gradients against the reference's own outputs
    (closed-form inverse affine vs its LU inverse: <= 1e-4)."""
    from spair_pytorch_amd import modules
    u = np.load(os.path.join(golden_dir, "units.npz"))
    sp = torch.from_numpy(u["inv_sprite"]).cuda().requires_grad_(True)
    zw = torch.from_numpy(u["stn_zw"]).cuda().requires_grad_(True)
    I = u["inv_out"].shape[-1]
    out = modules.stn(sp, zw, [I, I], torch.device("cuda"), inverse=True)
    assert out.shape == u["inv_out"].shape
    assert np.abs(out.detach().cpu().numpy() - u["inv_out"]).max() < 1e-4
    out.backward(torch.from_numpy(u["inv_gout"]).cuda())
    assert np.abs(sp.grad.cpu().numpy() - u["inv_dsprite"]).max() <= 1e-3 * np.abs(u["inv_dsprite"]).max()
    assert np.abs(zw.grad.cpu().numpy() - u["inv_dzw"]).max() <= 2e-3 * np.abs(u["inv_dzw"]).max()


def test_stn_forward_is_differentiable(golden_dir):
    from spair_pytorch_amd import modules
    u = np.load(os.path.join(golden_dir, "units.npz"))
    img = torch.from_numpy(u["stn_img"]).cuda()
    zw = torch.from_numpy(u["stn_zw"]).cuda().requires_grad_(True)
    g = modules.stn(img, zw, [28, 28], torch.device("cuda"))
    assert np.abs(g.detach().cpu().numpy() - u["stn_glimpse"]).max() < 1e-5
    g.backward(torch.from_numpy(u["stn_gw"]).cuda())
    assert np.abs(zw.grad.cpu().numpy() - u["stn_dzw"]).max() <= 3e-4 * np.abs(u["stn_dzw"]).max()


def test_backbone_and_mlp_forward_standalone():
    """Backbone.forward (modules.py:107-111) and SequentialMultipleOutput.forward (modules.py:282-284) against torch's CPU ops on the
    same modules' weights."""
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.modules import Backbone, build_MLP, hip_mlp_forward
    cfg.set_grid(48, (2, 2, 2, 1, 1, 1))
    torch.manual_seed(0)
    bb = Backbone([1, 48, 48], 100)
    x = torch.rand(3, 1, 48, 48)
    with torch.no_grad():
        ref = bb.net(bb.padding(x))
    out = bb.cuda().forward(x.cuda())
    assert out.shape == ref.shape == (3, 100, 6, 6)
    assert np.abs(out.cpu().numpy() - ref.numpy()).max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    net = build_MLP(478, multiple_output=(2, 100))
    xi = torch.randn(37, 478)
    with torch.no_grad():
        hb = net.body(xi)
        refs = [layer(hb) for layer in net.output_layers]
    outs = list(net.cuda().forward(xi.cuda()))          # a generator, like the reference's
    assert len(outs) == 2
    for o, r in zip(outs, refs):
        assert np.abs(o.cpu().numpy() - r.numpy()).max() <= 2e-5 * max(1.0, float(r.abs().max()))
    enc = build_MLP(784, 100, hidden_layers=[256, 128])
    xe = torch.rand(5, 784)
    with torch.no_grad():
        r = enc(xe)
    assert np.abs(hip_mlp_forward(enc.cuda(), xe.cuda()).cpu().numpy() - r.numpy()).max() <= 2e-5 * max(1.0, float(r.abs().max()))


def test_reference_train_loop_with_stock_adam_equals_fused_adam():
    """train.py:64-67 verbatim -- torch.optim.Adam(model.parameters(), lr=1e-4), zero_grad() (set_to_none), loss.backward(retain_graph=True),
    step() -- for 3 steps equals the same steps with FusedAdam on the flat buffers."""
    from spair_pytorch_amd.optim import FusedAdam
    z, case = load_case("c1_b8_step1001")
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    ma, mb = _model(case), _model(case)
    optimizer = torch.optim.Adam(ma.parameters(), lr=1e-4)
    fused = FusedAdam(mb, lr=1e-4)
    global_step = 1001
    for _ in range(3):
        optimizer.zero_grad()
        loss, out_img, z_where, z_pres = ma(x, global_step, noise=noise)
        loss.backward(retain_graph=True)
        fused.zero_grad()
        loss_b = mb(x, global_step, noise=noise)[0]
        loss_b.backward()
        # two runs of the same backward differ in the last bits (fp32 atomics in the bias column sums), and Adam turns a gradient that
        # is pure rounding noise (|g| ~ eps = 1e-8) into an update of up to +-lr: one parameter in a million then differs by ~1e-4
        # between two independent runs.  What this test compares is the optimizer arithmetic, so both optimizers get the same gradients.
        gdiff = (mb.flat_gradients() - ma.flat_gradients()).abs().max().item()
        assert gdiff <= 1e-5 * ma.flat_gradients().abs().max().item()
        mb.flat_gradients().copy_(ma.flat_gradients())
        optimizer.step()
        fused.step()
        global_step += 1
        assert abs(loss.item() - loss_b.item()) <= 1e-5 * abs(loss_b.item())
    pa, pb = ma.flat_parameters().cpu().numpy(), mb.flat_parameters().cpu().numpy()
    assert np.abs(pa - pb).max() <= 2e-6
    for k, p in ma.named_parameters():                  # the optimizer worked on the flat buffer's views all along
        if k.startswith("attn."):
            assert p.grad is None


def test_stale_activations_raise_and_engines_are_cached():
    """One set of saved activations per batch size: backward after a later forward of that size raises (instead of back-propagating through
    overwritten activations); two batch sizes alternate without re-allocating either workspace."""
    from spair_pytorch_amd._lib import SpairHipError
    z, case = load_case("c1_b8_step1001")
    m = _model(case)
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    half = {k: v[:4] for k, v in noise.items()}
    loss1 = m(x, 1001, noise=noise)[0]
    with torch.no_grad():
        m(x[:4], 1001, noise=half)                      # another batch size (evaluation): its own workspace, loss1 stays valid
    ws8 = m._engines[8]["workspace"].data_ptr()
    loss1.backward()
    g1 = m.flat_gradients().clone()
    m.zero_grad()
    loss2 = m(x, 1001, noise=noise)[0]
    m(x, 1001, noise=noise)                             # same batch size again: loss2's activations are gone
    with pytest.raises(SpairHipError):
        loss2.backward()
    assert m._engines[8]["workspace"].data_ptr() == ws8 and set(m._engines) == {4, 8}
    m.zero_grad()
    loss3 = m(x, 1001, noise=noise)[0]
    loss3.backward()
    assert torch.allclose(m.flat_gradients(), g1, rtol=1e-4, atol=1e-5)


def test_conv_object_encoder_decoder_variant():
    """the opt-in convolutional glimpse encoder / decoder built from CONV_OBJECT_ENCODER_TOPOLOGY (config.py:15-20; the
    reference's own classes, models.py:606-665, cannot run) -- HIP implicit-GEMM forward against torch's fp32 CPU conv / conv_transpose on
    the same weights.  Parity with the reference is unpinned by construction."""
    from spair_pytorch_amd.modules import ObjectConvDecoder, ObjectConvEncoder
    torch.manual_seed(1)
    enc = ObjectConvEncoder([1, 28, 28], 100)
    assert enc.shapes == [(1, 28, 28), (32, 13, 13), (32, 6, 6), (32, 2, 2), (32, 2, 2)]
    x = torch.rand(9, 1, 28, 28)
    with torch.no_grad():
        ref = enc.out(enc.conv(x).flatten(start_dim=1))
    out = enc.cuda().forward(x.cuda())
    assert out.shape == (9, 100)
    assert np.abs(out.cpu().numpy() - ref.numpy()).max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    dec = ObjectConvDecoder(50, 2)
    z = torch.randn(9, 50)
    with torch.no_grad():
        refd = dec.conv(dec.inp(z).view(9, *dec.top))
    assert refd.shape == (9, 2, 28, 28)
    outd = dec.cuda().forward(z.cuda())
    assert outd.shape == refd.shape
    assert np.abs(outd.cpu().numpy() - refd.numpy()).max() <= 2e-5 * max(1.0, float(refd.abs().max()))


def test_whole_step_is_capturable_in_a_hip_graph():
    """the helper stream forks from and joins back into the caller's stream by events only, nothing in the step
    allocates or synchronises -- so forward + backward + fused Adam can be captured by torch.cuda.graph (hipStreamBeginCapture on the
    caller's stream) and replayed.  One replay equals the eager step from the same state to the rounding of the fp32 atomics.  (Scalars
    the C-ABI takes by value -- Adam's step count, the global_step schedules -- are frozen into a captured graph, so a graph stands for
    ONE step index; and on MI355X the replay is slower than eager issue, 4.5 vs 3.9 ms at config 2 (tools/exp/graph_capture.py): the
    step is GPU-bound with 40 us of idle time and the graph serialises part of the two-stream overlap.  Eager is the product path.)"""
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd.optim import FusedAdam
    z, case = load_case("c1_b8_step1001")
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    L.check(L.lib().spair_init(), "spair_init")             # stream / event creation is not capturable: done ahead

    def run(use_graph):
        m = _model(case)
        opt = FusedAdam(m, lr=1e-4)

        def step():
            loss = m(x, 1001, noise=noise)[0]
            loss.backward()
            opt.step()
            return loss.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                          # warm-up outside the capture (kernel attributes, optimizer state)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if use_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = step()
            g.replay()
        else:
            loss = step()
        torch.cuda.synchronize()
        return float(loss), m.flat_parameters().cpu().numpy().copy()

    le, pe = run(False)
    lg, pg = run(True)
    assert abs(le - lg) <= 1e-5 * abs(le)
    assert_adam_updates_close(pe, pg, 1e-4)
