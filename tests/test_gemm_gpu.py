"""GPU parity of the MFMA GEMM core (csrc/gemm.hip) through the C ABI, against torch-CPU fp32."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from spair_pytorch_amd import _lib as L
    return L


def _i(*a):
    return (ctypes.c_int * len(a))(*a)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (300, 100, 324), (2048, 108, 100), (9000, 256, 784), (8192, 128, 128), (70, 1, 100)])
def test_gemm_nt(dtype, M, N, K):
    L = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    Kp = (K + 7) // 8 * 8
    A = torch.zeros(M, Kp)
    A[:, :K] = torch.randn(M, K, generator=g)
    W = torch.zeros(N, Kp)
    W[:, :K] = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    Wb = Wd.to(torch.bfloat16).contiguous() if dtype == 1 else Wd
    ldc = (N + 3) // 4 * 4
    C = torch.full((M, ldc), 7.0, device="cuda")
    rc = L.lib().spair_gemm_nt(L.ptr(Ad), Kp, L.ptr(Wb), Kp, L.ptr(C), ldc, M, N, Kp, L.ptr(bd), None, 0, 1, 0, dtype, L.stream())
    L.check(rc, "gemm_nt")
    if dtype == 1:
        ref = torch.relu(A.to(torch.bfloat16).float() @ W.to(torch.bfloat16).float().t() + bias)
        tol = 2e-3
    else:
        ref = torch.relu(A @ W.t() + bias)
        tol = 2e-5
    out = C.cpu()
    assert (out[:, :N] - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())
    assert (out[:, N:] == 7.0).all()  # pad columns untouched
    # accumulate + relu-backward mask
    Y = torch.randn(M, ldc, generator=g).cuda()
    C2 = torch.ones(M, ldc, device="cuda")
    rc = L.lib().spair_gemm_nt(L.ptr(Ad), Kp, L.ptr(Wb), Kp, L.ptr(C2), ldc, M, N, Kp, None, L.ptr(Y), ldc, 0, 1, dtype, L.stream())
    L.check(rc, "gemm_nt")
    if dtype == 1:
        base = A.to(torch.bfloat16).float() @ W.to(torch.bfloat16).float().t()
    else:
        base = A @ W.t()
    ref2 = (base + 1.0) * (Y.cpu()[:, :N] > 0)
    assert (C2.cpu()[:, :N] - ref2).abs().max() <= tol * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("M,N,R", [(100, 324, 2048), (1568, 256, 4096), (128, 16, 5000), (4, 100, 777), (104, 480, 1000)])
def test_gemm_tn_and_colsum(M, N, R, dtype):
    L = _lib()
    g = torch.Generator().manual_seed(M * 3 + N + R)
    A = torch.randn(R, M, generator=g)
    B = torch.randn(R, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    Cd, Ad, Bd = C0.cuda(), A.cuda(), B.cuda()  # keep device tensors alive across the launches
    rc = L.lib().spair_gemm_tn(L.ptr(Ad), M, L.ptr(Bd), N, L.ptr(Cd), N, M, N, R, dtype, L.stream())
    L.check(rc, "gemm_tn")
    if dtype == 1:
        ref = C0.double() + A.to(torch.bfloat16).double().t() @ B.to(torch.bfloat16).double()
    else:
        ref = C0.double() + A.double().t() @ B.double()
    assert (Cd.cpu().double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    out = torch.zeros(M, device="cuda")
    L.check(L.lib().spair_colsum(L.ptr(Ad), M, R, M, L.ptr(out), L.stream()), "colsum")
    assert (out.cpu().double() - A.double().sum(0)).abs().max() <= 1e-4 * A.double().sum(0).abs().max() + 1e-3


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,Hin,Cin,Cout,k,s", [(3, 34, 128, 128, 4, 2), (2, 16, 128, 100, 1, 1), (5, 22, 8, 36, 4, 2)])
def test_conv_fwd_dgrad_wgrad(dtype, B, Hin, Cin, Cout, k, s):
    """Implicit-GEMM conv on NHWC: forward, stride-2 data-gradient by output-parity class, weight-gradient."""
    L = _lib()
    g = torch.Generator().manual_seed(B + Hin + Cin)
    x = torch.randn(B, Cin, Hin, Hin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    Hout = (Hin - k) // s + 1
    xq, wq = (x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()) if dtype == 1 else (x, w)
    ref = torch.relu(torch.nn.functional.conv2d(xq, wq, bias, stride=s))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    K = k * k * Cin
    w_p = w.permute(0, 2, 3, 1).reshape(Cout, K).contiguous().cuda()      # [Cout][ky][kx][ci]
    w_dev = w_p.to(torch.bfloat16) if dtype == 1 else w_p
    ldc = (Cout + 3) // 4 * 4
    out = torch.zeros(B, Hout, Hout, ldc, device="cuda")
    M = B * Hout * Hout
    conv = _i(Hin, Hin, Cin, Hout, Hout, k, k, s, s, 1, 1, 0, 0)
    bias_d = bias.cuda()
    rc = L.lib().spair_gemm_nt_conv(L.ptr(x_nhwc), conv, L.ptr(w_dev), K, L.ptr(out), ldc, M, Cout, K, L.ptr(bias_d),
                                    None, 0, 1, 0, None, dtype, L.stream())
    L.check(rc, "conv fwd")
    got = out.cpu()[..., :Cout].permute(0, 3, 1, 2)
    tol = 3e-3 if dtype == 1 else 2e-5
    assert (got - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())

    # ---- weight gradient (fp32 TN path): dW[co,(ky,kx,ci)] = sum_m dOut[m,co] * im2col(x)[m,k]
    if Cout % 4 == 0:
        go = torch.randn(B, Cout, Hout, Hout, generator=g)
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        torch.nn.functional.conv2d(xr, wr, None, stride=s).backward(go)
        go_nhwc = go.permute(0, 2, 3, 1).contiguous().cuda()
        dW = torch.zeros(Cout, K, device="cuda")
        rc = L.lib().spair_gemm_tn_conv(L.ptr(go_nhwc), Cout, L.ptr(x_nhwc), conv, L.ptr(dW), K, Cout, K, M, dtype, L.stream())
        L.check(rc, "conv wgrad")
        ref_dw = wr.grad.permute(0, 2, 3, 1).reshape(Cout, K)
        assert (dW.cpu() - ref_dw).abs().max() <= (1e-2 if dtype == 1 else 5e-5) * ref_dw.abs().max()
        # ---- data gradient for k=4,s=2: 4 parity classes, each a 2x2 stride-1 "conv" over dOut
        if k == 4 and s == 2 and dtype == 0:
            dX = torch.zeros(B, Hin, Hin, Cin, device="cuda")
            for py in range(2):
                for px in range(2):
                    # Wc[ci][(ty,tx,co)] = w[co, ci, py+2ty, px+2tx]
                    wc = w[:, :, py::2, px::2].permute(1, 2, 3, 0).reshape(Cin, 4 * Cout).contiguous().cuda()
                    Hc = (Hin - py + 1) // 2  # rows iy' with 2*iy'+py < Hin
                    Wc = (Hin - px + 1) // 2
                    cd = _i(Hout, Hout, Cout, Hc, Wc, 2, 2, 1, 1, -1, -1, 0, 0)
                    cm = _i(Hc, Wc, Hin, Hin, 2, 2, py, px)
                    rc = L.lib().spair_gemm_nt_conv(L.ptr(go_nhwc), cd, L.ptr(wc), 4 * Cout, L.ptr(dX), Cin, B * Hc * Wc, Cin,
                                                    4 * Cout, None, None, 0, 0, 0, cm, 0, L.stream())
                    L.check(rc, "conv dgrad")
            ref_dx = xr.grad.permute(0, 2, 3, 1)
            assert (dX.cpu() - ref_dx).abs().max() <= 5e-5 * ref_dx.abs().max()


# ---- bf16-stored operand kernels (gemm16.hip) -----------------------------------------------------------------------
def _bf(t):
    return t.to(torch.bfloat16).contiguous()


@pytest.mark.parametrize("M,N,K,c16", [(300, 100, 128, 1), (9000, 256, 784 + 8, 0), (65536 // 8, 1568, 256, 0), (4096, 128, 2048, 1)])
def test_gemm_nt16_plain(M, N, K, c16):
    L = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    W = _bf(torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Y = _bf(torch.randn(M, N, generator=g)).cuda()
    ldc = (N + 7) // 8 * 8
    C = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16 if c16 else torch.float32)
    rc = L.lib().spair_gemm_nt16(L.ptr(A), K, L.ptr(W), K, L.ptr(C), ldc, M, N, K, L.ptr(bias), L.ptr(Y), N, 1, 1, c16, None, None, L.stream())
    L.check(rc, "nt16")
    ref = torch.relu(A.float().cpu() @ W.float().cpu().t() + bias.cpu()) * (Y.float().cpu() > 0)
    got = C.float().cpu()[:, :N]
    tol = 1.2e-2 if c16 else 2e-3
    assert (got - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,Hin,Cin,Cout,k,s", [(3, 34, 128, 128, 4, 2), (5, 22, 8, 40, 4, 2)])
def test_conv16_fwd_dgrad_wgrad(B, Hin, Cin, Cout, k, s):
    L = _lib()
    g = torch.Generator().manual_seed(B + Hin + Cin)
    x = _bf(torch.randn(B, Cin, Hin, Hin, generator=g)).float()
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).float()
    bias = torch.randn(Cout, generator=g)
    Hout = (Hin - k) // s + 1
    ref = torch.relu(torch.nn.functional.conv2d(x, w, bias, stride=s))
    x16 = _bf(x.permute(0, 2, 3, 1)).cuda()
    K = k * k * Cin
    w16 = _bf(w.permute(0, 2, 3, 1).reshape(Cout, K)).cuda()
    M = B * Hout * Hout
    out = torch.zeros(M, Cout, device="cuda", dtype=torch.bfloat16)
    conv = _i(Hin, Hin, Cin, Hout, Hout, k, k, s, s, 1, 1, 0, 0)
    bias_d = bias.cuda()
    L.check(L.lib().spair_gemm_nt16(L.ptr(x16), 0, L.ptr(w16), K, L.ptr(out), Cout, M, Cout, K, L.ptr(bias_d), None, 0, 0, 1, 1, conv, None, L.stream()), "conv16 fwd")
    got = out.float().cpu().view(B, Hout, Hout, Cout).permute(0, 3, 1, 2)
    assert (got - ref).abs().max() <= 1.2e-2 * max(1.0, ref.abs().max().item())
    # weight gradient: A = dOut bf16, B = conv gather of x (bf16), fused bias gradient
    go = _bf(torch.randn(B, Cout, Hout, Hout, generator=g)).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, None, stride=s).backward(go)
    go16 = _bf(go.permute(0, 2, 3, 1).reshape(M, Cout)).cuda()
    dW = torch.zeros(Cout, Cin, k, k, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    L.check(L.lib().spair_gemm_tn16(L.ptr(go16), Cout, L.ptr(x16), 0, 1, L.ptr(dW), K, Cout, K, M, conv, Cin, k * k, L.ptr(db), None, 0, L.stream()), "conv16 wgrad")
    assert (dW.cpu() - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max()
    assert (db.cpu() - go.sum((0, 2, 3))).abs().max() <= 2e-3 * go.sum((0, 2, 3)).abs().max() + 1e-3
    # fp32 gather for B (first backbone layer style): same result
    x32 = x.permute(0, 2, 3, 1).contiguous().cuda()
    dW2 = torch.zeros(Cout, Cin, k, k, device="cuda")
    L.check(L.lib().spair_gemm_tn16(L.ptr(go16), Cout, L.ptr(x32), 0, 0, L.ptr(dW2), K, Cout, K, M, conv, Cin, k * k, None, None, 0, L.stream()), "conv16 wgrad f32 B")
    assert (dW2.cpu() - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max()
    # split-K through partial tiles + reduce pass (what the training step uses) instead of atomics; accumulates into dW3
    scratch = torch.empty(1536 * 128 * 128, device="cuda")
    dW3 = torch.ones(Cout, Cin, k, k, device="cuda")
    db3 = torch.zeros(Cout, device="cuda")
    L.check(L.lib().spair_gemm_tn16(L.ptr(go16), Cout, L.ptr(x16), 0, 1, L.ptr(dW3), K, Cout, K, M, conv, Cin, k * k, L.ptr(db3), L.ptr(scratch),
                                    ctypes.c_longlong(scratch.numel()), L.stream()), "conv16 wgrad split-K reduce")
    assert (dW3.cpu() - 1.0 - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max()
    assert (db3.cpu() - go.sum((0, 2, 3))).abs().max() <= 2e-3 * go.sum((0, 2, 3)).abs().max() + 1e-3
    # data gradient by output-parity classes with relu mask and row remap
    if k == 4 and s == 2:
        mask = _bf(torch.randn(B, Hin, Hin, Cin, generator=g)).cuda()
        dX = torch.zeros(B, Hin, Hin, Cin, device="cuda", dtype=torch.bfloat16)
        for py in range(2):
            for px in range(2):
                wc = _bf(w[:, :, py::2, px::2].permute(1, 2, 3, 0).reshape(Cin, 4 * Cout)).cuda()
                Hc, Wc = (Hin - py + 1) // 2, (Hin - px + 1) // 2
                cd = _i(Hout, Hout, Cout, Hc, Wc, 2, 2, 1, 1, -1, -1, 0, 0)
                cm = _i(Hc, Wc, Hin, Hin, 2, 2, py, px)
                L.check(L.lib().spair_gemm_nt16(L.ptr(go16), 0, L.ptr(wc), 4 * Cout, L.ptr(dX), Cin, B * Hc * Wc, Cin, 4 * Cout, None,
                                                L.ptr(mask), Cin, 1, 0, 1, cd, cm, L.stream()), "conv16 dgrad")
        ref_dx = xr.grad.permute(0, 2, 3, 1) * (mask.float().cpu() > 0)
        assert (dX.float().cpu() - ref_dx).abs().max() <= 1.2e-2 * ref_dx.abs().max()


def test_gemm_tn16_plain():
    L = _lib()
    g = torch.Generator().manual_seed(5)
    R, M, N = 5000, 1568, 256
    A, B = _bf(torch.randn(R, M, generator=g)).cuda(), _bf(torch.randn(R, N, generator=g)).cuda()
    C = torch.zeros(M, N, device="cuda")
    cs = torch.zeros(M, device="cuda")
    L.check(L.lib().spair_gemm_tn16(L.ptr(A), M, L.ptr(B), N, 1, L.ptr(C), N, M, N, R, None, 0, 0, L.ptr(cs), None, 0, L.stream()), "tn16")
    ref = A.float().cpu().double().t() @ B.float().cpu().double()
    assert (C.cpu().double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert (cs.cpu().double() - A.float().cpu().double().sum(0)).abs().max() <= 1e-4 * A.float().cpu().double().sum(0).abs().max() + 1e-3


@pytest.mark.parametrize("R,M,N", [(5000, 1568, 256), (4100, 104, 56), (65536, 256, 128), (300, 128, 2048)])
def test_gemm_tn_ring_plain(R, M, N):
    """tn_ring.hip (DMA-staged split-K weight-gradient GEMM, taken whenever scratch is passed and B is bf16): plain rows, R not a multiple of
    the 64-row stage, partial tiles in both directions, bias gradient through the ones-fragment MFMA; two runs agree bit for bit."""
    L = _lib()
    g = torch.Generator().manual_seed(R + M)
    A, B = _bf(torch.randn(R, M, generator=g)).cuda(), _bf(torch.randn(R, N, generator=g)).cuda()
    scratch = torch.empty(1536 * 128 * 128, device="cuda")
    outs = []
    for _ in range(2):
        C = torch.full((M, N), 0.25, device="cuda")
        cs = torch.zeros(M, device="cuda")
        L.check(L.lib().spair_gemm_tn16(L.ptr(A), M, L.ptr(B), N, 1, L.ptr(C), N, M, N, R, None, 0, 0, L.ptr(cs), L.ptr(scratch),
                                        ctypes.c_longlong(scratch.numel()), L.stream()), "tn ring")
        outs.append((C.cpu(), cs.cpu()))
    ref = A.float().cpu().double().t() @ B.float().cpu().double()
    assert (outs[0][0].double() - 0.25 - ref).abs().max() <= 2e-5 * ref.abs().max()
    ref_cs = A.float().cpu().double().sum(0)
    assert (outs[0][1].double() - ref_cs).abs().max() <= 1e-4 * ref_cs.abs().max() + 1e-3
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,Hin", [(3, 142), (2, 54)])
def test_stem_wgrad16_vs_torch(B, Hin):
    """Backbone layer 0 (1 -> 128 channels, 4x4, stride 2): dedicated weight/bias gradient kernel vs torch autograd."""
    L = _lib()
    g = torch.Generator().manual_seed(Hin)
    Cout, k, s = 128, 4, 2
    Hout = (Hin - k) // s + 1
    M = B * Hout * Hout
    x = torch.rand(B, 1, Hin, Hin, generator=g)
    go = _bf(torch.randn(B, Cout, Hout, Hout, generator=g)).float()
    w = torch.zeros(Cout, 1, k, k, requires_grad=True)
    torch.nn.functional.conv2d(_bf(x).float(), w, None, stride=s).backward(go)     # the kernel rounds the patches to bf16
    go16 = _bf(go.permute(0, 2, 3, 1).reshape(M, Cout)).cuda()
    xd = x.reshape(B, Hin, Hin).contiguous().cuda()
    dW = torch.full((Cout, 1, k, k), 0.5, device="cuda")     # accumulated into
    db = torch.zeros(Cout, device="cuda")
    scratch = torch.empty(512 * 128 * 32, device="cuda")
    L.check(L.lib().spair_stem_wgrad16(L.ptr(go16), L.ptr(xd), L.ptr(dW), L.ptr(db), L.ptr(scratch), ctypes.c_longlong(scratch.numel()),
                                       B, Hin, s, Hout, L.stream()), "stem wgrad")
    assert (dW.cpu() - 0.5 - w.grad).abs().max() <= 2e-3 * w.grad.abs().max()
    ref_db = go.sum((0, 2, 3))
    assert (db.cpu() - ref_db).abs().max() <= 2e-3 * ref_db.abs().max() + 1e-3


def test_conv1x1_stack_fwd_bwd_vs_torch():
    """Fused per-pixel MLP (Backbone's trailing 1x1 convs + conv_out): 128 -> 128 -> 128 -> 128 -> 100, bf16 activations."""
    L = _lib()
    g = torch.Generator().manual_seed(11)
    M, Lr, couts = 1000, 4, [128, 128, 128, 100]
    x = _bf(torch.relu(torch.randn(M, 128, generator=g)))
    Ws = [_bf(torch.randn(co, 128, generator=g) / 128 ** 0.5) for co in couts]
    bs = [torch.randn(co, generator=g) * 0.1 for co in couts]
    # torch reference with bf16 rounding of every stored activation
    acts, h = [x.float()], x.float()
    for l in range(Lr):
        h = h @ Ws[l].float().t() + bs[l]
        if l < Lr - 1:
            h = _bf(torch.relu(h)).float()
            acts.append(h)
    ref_out = h

    def parr(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else 0 for t in ts])

    xd = x.cuda()
    Wd = [w.cuda() for w in Ws]
    bd = [b.cuda() for b in bs]
    Y = [torch.zeros(M, 128, device="cuda", dtype=torch.bfloat16) for _ in range(Lr - 1)] + [None]
    out = torch.zeros(M, 104, device="cuda")
    L.check(L.lib().spair_conv1x1_stack_fwd16(L.ptr(xd), parr(Wd), _i(128, 128, 128, 128), _i(*couts), parr(bd), parr(Y), L.ptr(out), 104, M, Lr,
                                              L.stream()), "1x1 stack fwd")
    assert (out[:, :100].cpu() - ref_out).abs().max() <= 2e-2 * ref_out.abs().max()
    for l in range(Lr - 1):
        assert (Y[l].float().cpu() - acts[l + 1]).abs().max() <= 2e-2 * acts[l + 1].abs().max()
    # backward: dX_{l-1} = (dX_l W_l) * [act_{l-1} > 0], top layer first
    dy = torch.zeros(M, 104)
    dy[:, :100] = torch.randn(M, 100, generator=g)
    dy16 = _bf(dy)
    WT = [_bf(Ws[l].t().contiguous()) for l in range(Lr)]          # [128][cout]
    WTp = []
    for l in range(Lr):
        ld = (couts[l] + 7) // 8 * 8
        t = torch.zeros(128, ld, dtype=torch.bfloat16)
        t[:, :couts[l]] = WT[l]
        WTp.append(t.cuda())
    order = list(range(Lr - 1, -1, -1))
    gate = [acts[l].to(torch.bfloat16).cuda() for l in order]
    dX = [torch.zeros(M, 128, device="cuda", dtype=torch.bfloat16) for _ in order]
    L.check(L.lib().spair_conv1x1_stack_bwd16(L.ptr(dy16.cuda()), 104, 100, parr([WTp[l] for l in order]), _i(*[WTp[l].shape[1] for l in order]),
                                              _i(*[couts[l] for l in order]), parr(gate), parr(dX), M, Lr, L.stream()), "1x1 stack bwd")
    gcur = dy16.float()[:, :100]
    for k, l in enumerate(order):
        gcur = (gcur @ Ws[l].float()) * (acts[l] > 0)
        gcur = _bf(gcur).float()
        assert (dX[k].float().cpu() - gcur).abs().max() <= 3e-2 * gcur.abs().max(), (k, l)
